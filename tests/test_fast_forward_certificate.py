"""CPU: the error-band certificate behind the fast-forward of the bisection in k_sweep1 (batotp_amd/csrc/sweep1.hip.h,
DESIGN.md section 4), checked on its own, away from the kernel.

The kernel skips the constraint check of a bisection candidate c when |c*c - x*| > band, band = 2^-40 R x*, and takes the
outcome "violated <=> c*c > x*".  This file restates, in numpy fp64 (IEEE, one rounding per operation, the operations of
BA::verifySecondOrderConstraints in their order: reference batotp/ba.cpp:1495-1534), the check for problems whose constraints are
lines in x = sdot^2 -- joint acceleration limits, torque limits with a3 = 0, the clamp +-sddotMax, standing joints -- and the
kernel's computation of x*, R and the band, with the approximate reciprocals replaced by exact ones perturbed by a few ulps.
Then it probes candidates at relative distances 1e-16 .. 1e-2 on both sides of x*: wherever the certificate says "certain", the
real check must agree; and the disagreements of the bare comparison must all lie deep inside the band (its margin).
"""
import numpy as np
import pytest

EPS = 2.0 ** -53
INF = np.inf


def _real_check(x, c, A, v, D, thr_v, thr_a, sddot_max, trq, a3=None):
    """verify_second_order for line constraints; x = fl(c*c).  Returns violated (bool).  ba.cpp:1495-1534 in order."""
    H, L = sddot_max, -sddot_max
    force = False
    if trq is not None:
        a1, a2, a4, tmax, tmin = trq
        for q in range(len(a1)):
            tmp1 = (0.0 if a3 is None else a3[q]) * c + a4[q]   # a3*sdot + a4 (a3 = 0: the cable robot)
            if not (abs(a1[q]) < thr_v):
                tmp2 = a2[q] * x + tmp1
                s0 = (tmax[q] - tmp2) / a1[q]
                s1 = (tmin[q] - tmp2) / a1[q]
                H = min(H, max(s0, s1))
                L = max(L, min(s0, s1))
    for q in range(len(A)):
        if abs(v[q]) < thr_v:
            if not (abs(D[q]) < thr_a):
                if x > A[q] / abs(D[q]):
                    force = True
        else:
            s = np.sign(v[q])
            vterm = D[q] * x
            H = min(H, (s * A[q] - vterm) / v[q])
            L = max(L, (-s * A[q] - vterm) / v[q])
    if force:
        H = -INF
    return bool(L > H)


def _rcp(rng, d):
    """a reciprocal as good as the kernel's (two Newton steps on v_rcp_f64): exact, then off by up to 4 ulps either way"""
    return (1.0 / d) * (1.0 + rng.integers(-4, 5) * 2.0 * EPS)


def _certificate(rng, x_top, A, v, D, thr_v, thr_a, sddot_max, trq):
    """x*, band, threshold form of the kernel (sweep1.hip.h, CERTIFIED FAST-FORWARD); None when it declines"""
    lines = []                                               # (au, al, m, e)
    x_force = INF
    for q in range(len(A)):
        if abs(v[q]) < thr_v:
            if not (abs(D[q]) < thr_a):
                x_force = min(x_force, A[q] / abs(D[q]))
        else:
            rv = _rcp(rng, v[q])
            au = A[q] * abs(rv)
            m = D[q] * rv
            lines.append((au, -au, m, au + abs(m) * x_top))
    if trq is not None:
        a1, a2, a4, tmax, tmin = trq
        for q in range(len(a1)):
            if not (abs(a1[q]) < thr_v):
                r1 = _rcp(rng, a1[q])
                q0, q1 = (tmax[q] - a4[q]) * r1, (tmin[q] - a4[q]) * r1
                lines.append((max(q0, q1), min(q0, q1), a2[q] * r1,
                              (abs(tmax[q]) + abs(tmin[q]) + 2.0 * abs(a4[q]) + abs(a2[q]) * x_top) * abs(r1)))
    u_min = min([l[0] for l in lines] + [sddot_max])
    l_max = max([l[1] for l in lines] + [-sddot_max])
    e_max = max([l[3] for l in lines] + [0.0])
    s_min2 = 0.5 * (u_min - l_max)
    xs = INF
    for (au, al, m, _) in lines:
        if m > 0.0:
            xs = min(xs, (au + sddot_max) * _rcp(rng, m))
        if m < 0.0:
            xs = min(xs, (sddot_max - al) * _rcp(rng, -m))
        for (_, ali, mi, _) in lines:
            dm = m - mi
            if dm > 0.0:
                xs = min(xs, (au - ali) * _rcp(rng, dm))
    xstar = min(xs, 4.0 * x_top)
    if not s_min2 > 0.0:
        return None
    R = e_max * _rcp(rng, s_min2)
    band = (R * 2.0 ** -40) * xstar
    force_first = x_force < xstar - band
    sane = (R < 2.0 ** 30 and s_min2 > 1e-100 and e_max < 1e100 and 1e-100 < x_top < 1e100 and xstar > 1e-100
            and (force_first or x_force > xstar + band))
    if not sane:
        return None
    return (x_force if force_first else xstar, -1.0 if force_first else band, xstar, R)


def _problem(rng, kind):
    nJ = int(rng.integers(1, 9))
    A = 10.0 ** rng.uniform(-3, 3, nJ)
    v = rng.normal(size=nJ) * 10.0 ** rng.uniform(-2, 2)
    D = rng.normal(size=nJ) * 10.0 ** rng.uniform(-2, 3)
    thr_v, thr_a = 1e-6, 1e-6
    if kind == "slow joint":                  # theta' barely above the threshold: a huge a_q = amax / |theta'|
        v[rng.integers(0, nJ)] = thr_v * (1.0 + 10.0 ** rng.uniform(-6, 2)) * rng.choice([-1, 1])
    elif kind == "standing joint":            # theta' below the threshold: the x <= amax / |theta''| rule
        v[rng.integers(0, nJ)] = thr_v * rng.uniform(0, 0.99)
    elif kind == "parallel lines" and nJ > 1:  # two joints with nearly the same m = theta'' / theta'
        a, b = rng.choice(nJ, 2, replace=False)
        f = 1.0 + 10.0 ** rng.uniform(-13, -3)
        v[b], D[b] = v[a] * 3.0, D[a] * 3.0 * f
    sddot_max = 10.0 ** rng.uniform(2, 9)
    trq = None
    if kind == "torque" or rng.random() < 0.25:
        n = min(nJ, 4)
        a1 = rng.normal(size=n) * 10.0 ** rng.uniform(-2, 1)
        a2 = rng.normal(size=n) * 10.0 ** rng.uniform(-2, 2)
        a4 = rng.normal(size=n)
        tmax = np.abs(a4) + 10.0 ** rng.uniform(-2, 2, n)
        tmin = -np.abs(a4) - 10.0 ** rng.uniform(-2, 2, n) if rng.random() < 0.5 else np.minimum(a4 - 10.0 ** rng.uniform(-3, 1, n), tmax)
        trq = (a1, a2, a4, tmax, tmin)
    return A, v, D, thr_v, thr_a, sddot_max, trq


@pytest.mark.parametrize("kind", ["plain", "slow joint", "standing joint", "parallel lines", "torque"])
def test_certain_candidates_get_the_outcome_of_the_real_check(kind):
    rng = np.random.default_rng({"plain": 1, "slow joint": 2, "standing joint": 3, "parallel lines": 4, "torque": 5}[kind])
    deltas = np.concatenate([10.0 ** np.linspace(-16, -2, 57), [0.05, 0.3]])
    used = probes = skipped = 0
    closest = []          # per problem: the band and the largest |delta| at which the bare comparison with x* was wrong
    for _ in range(400):
        A, v, D, thr_v, thr_a, sddot_max, trq = _problem(rng, kind)
        # a first candidate above the crossing: scan up from a feasible speed until the real check fails
        c0 = None
        for c in 10.0 ** np.linspace(-4, 6, 41):
            if _real_check(c * c, c, A, v, D, thr_v, thr_a, sddot_max, trq):
                c0 = c
                break
        if c0 is None or c0 == 1e-4:
            skipped += 1
            continue
        cert = _certificate(rng, c0 * c0, A, v, D, thr_v, thr_a, sddot_max, trq)
        if cert is None:
            skipped += 1
            continue
        x_thr, band_thr, xstar, R = cert
        used += 1
        last_wrong = 0.0
        for sgn in (-1.0, 1.0):
            for dl in deltas:
                c = np.sqrt(x_thr * (1.0 + sgn * dl))
                if not (0.0 < c <= c0):
                    continue
                x = c * c
                d = x - x_thr
                predicted = d > 0.0
                real = _real_check(x, c, A, v, D, thr_v, thr_a, sddot_max, trq)
                if abs(d) > band_thr:
                    probes += 1
                    assert predicted == real, (kind, dl, sgn, R, band_thr / xstar, cert)
                elif predicted != real:
                    last_wrong = max(last_wrong, abs(d) / xstar)
        if band_thr > 0 and last_wrong > 0.0:
            closest.append((band_thr / xstar, last_wrong))
    assert used > 150 and probes > 10000, (used, probes, skipped)
    # the band is not vacuous -- inside it the bare comparison does go wrong now and then -- and it has room to spare: only far
    # inside it (over 6000 problems of these five kinds the farthest wrong comparison lay at 3.7e-4 of the band's width)
    for rel_band, dl in closest:
        assert dl < 0.1 * rel_band


def _approx_check(rng, c_top, A, v, D, thr_v, thr_a, sddot_max, trq, a3):
    """the division-free check of the general form (sweep1.hip.h, 'general form for serial torque limits'): returns d(c) and the band"""
    x_top = c_top * c_top
    aa = np.full(len(A), INF); ma = np.zeros(len(A)); e = [0.0]
    x_force = INF
    for q in range(len(A)):
        if abs(v[q]) < thr_v:
            if not (abs(D[q]) < thr_a):
                x_force = min(x_force, A[q] / abs(D[q]))
        else:
            ra = _rcp(rng, v[q])
            aa[q] = A[q] * abs(ra); ma[q] = D[q] * ra
            e.append(aa[q] + abs(ma[q]) * x_top)
    a1, a2, a4, tmax, tmin = trq
    n = len(a1)
    tu = np.full(n, INF); tl = np.full(n, -INF); tb = np.zeros(n); tm = np.zeros(n)
    for q in range(n):
        if not (abs(a1[q]) < thr_v):
            r1 = _rcp(rng, a1[q])
            q0, q1 = (tmax[q] - a4[q]) * r1, (tmin[q] - a4[q]) * r1
            tu[q], tl[q], tb[q], tm[q] = max(q0, q1), min(q0, q1), a3[q] * r1, a2[q] * r1
            e.append((abs(tmax[q]) + abs(tmin[q]) + 2.0 * abs(a4[q]) + abs(a3[q]) * c_top + abs(a2[q]) * x_top) * abs(r1))
    band = max(e) * 2.0 ** -44

    def d_of(c):
        x = c * c
        U, Lw = sddot_max, -sddot_max
        for q in range(len(A)):
            U = min(U, aa[q] - ma[q] * x); Lw = max(Lw, -aa[q] - ma[q] * x)
        for q in range(n):
            tt = tb[q] * c + tm[q] * x
            U = min(U, tu[q] - tt); Lw = max(Lw, tl[q] - tt)
        return INF if x > x_force else -(U - Lw)
    return d_of, band


def test_the_division_free_check_of_the_general_form_certifies_the_real_one():
    """serial torque limits with friction (a3 != 0: bounds quadratic in sdot): wherever |d(c)| exceeds the band, the real check of
    ba.cpp:1495-1534 decides as sign(d) says; probed densely around the crossing found by bisection on the real check"""
    rng = np.random.default_rng(77)
    used = probes = 0
    worst = 0.0
    for _ in range(500):
        A, v, D, thr_v, thr_a, sddot_max, _ = _problem(rng, "plain")
        n = len(A)
        a1 = rng.normal(size=n) * 10.0 ** rng.uniform(-2, 1)
        a2 = rng.normal(size=n) * 10.0 ** rng.uniform(-2, 2)
        a3 = rng.normal(size=n) * 10.0 ** rng.uniform(-2, 1)
        a4 = rng.normal(size=n)
        tmax = np.abs(a4) + 10.0 ** rng.uniform(-1, 2, n)
        tmin = -np.abs(a4) - 10.0 ** rng.uniform(-1, 2, n)
        trq = (a1, a2, a4, tmax, tmin)
        chk = lambda c: _real_check(c * c, c, A, v, D, thr_v, thr_a, sddot_max, trq, a3)
        c0 = None
        for c in 10.0 ** np.linspace(-4, 6, 41):
            if chk(c):
                c0 = c
                break
        if c0 is None or c0 == 1e-4:
            continue
        d_of, band = _approx_check(rng, c0, A, v, D, thr_v, thr_a, sddot_max, trq, a3)
        lo, hi = 0.0, c0                          # the crossing below c0 the bisection would approach
        for _ in range(200):
            mid = 0.5 * (lo + hi)
            if chk(mid):
                hi = mid
            else:
                lo = mid
        used += 1
        for sgn in (-1.0, 1.0):
            for dl in 10.0 ** np.linspace(-16, -1, 61):
                c = hi * (1.0 + sgn * dl)
                if not (0.0 < c <= c0):
                    continue
                d = d_of(c)
                real = chk(c)
                if abs(d) > band:
                    probes += 1
                    assert (d > 0.0) == real, (dl, sgn, d, band)
                elif (d > 0.0) != real and band > 0:
                    worst = max(worst, abs(d) / band)
    assert used > 200 and probes > 10000, (used, probes)
    assert worst < 0.1          # wrong signs of the approximate check occur only deep inside the band


def _certain_failure(rng, x_top, A, v, D, thr_v, thr_a, sddot_max, trq):
    """the certificate of a stage whose bisection cannot succeed (sweep1.hip.h, CERTAIN FAILURE; round 6): the sddot interval is empty at
    x = 0 and the two lines that bind there have a negative gap at x_top as well -- both by more than 2^-40 E.  True / False."""
    lines = []                                               # (au, al, m, e)
    for q in range(len(A)):
        if not (abs(v[q]) < thr_v):
            rv = _rcp(rng, v[q])
            au = A[q] * abs(rv)
            m = D[q] * rv
            lines.append((au, -au, m, au + abs(m) * x_top))
    if trq is not None:
        a1, a2, a4, tmax, tmin = trq
        for q in range(len(a1)):
            if not (abs(a1[q]) < thr_v):
                r1 = _rcp(rng, a1[q])
                q0, q1 = (tmax[q] - a4[q]) * r1, (tmin[q] - a4[q]) * r1
                lines.append((max(q0, q1), min(q0, q1), a2[q] * r1,
                              (abs(tmax[q]) + abs(tmin[q]) + 2.0 * abs(a4[q]) + abs(a2[q]) * x_top) * abs(r1)))
    if not lines:
        return False
    u_min = min(l[0] for l in lines)
    l_max = max(l[1] for l in lines)
    e_max = max(l[3] for l in lines)
    uB, lB = min(u_min, sddot_max), max(l_max, -sddot_max)
    if not (0.5 * (uB - lB) < 0.0):
        return False
    mU = max(l[2] for l in lines if l[0] == u_min) if u_min < sddot_max else 0.0
    mL = min(l[2] for l in lines if l[1] == l_max) if l_max > -sddot_max else 0.0
    gap0 = uB - lB
    gap_top = gap0 - (mU - mL) * x_top
    tol = e_max * 2.0 ** -40
    return bool(e_max < 1e100 and x_top < 1e100 and abs(mU) < 1e100 and abs(mL) < 1e100 and gap0 < -tol and gap_top < -tol)


def test_a_certified_failure_fails_every_candidate_of_the_loop():
    """torque lines whose limits cannot be met at rest (the cable robot outside its tension-feasible region): where the certificate says
    "no speed in [0, first candidate] can pass a check", the real check (ba.cpp:1495-1534) must be violated at EVERY candidate of the
    reference's loop (ba.cpp:1276-1320: the bracket shrinks below each violated candidate, then the speed is halved a hundred times) and
    at random speeds in between; and the certificate must decline problems that are feasible somewhere in the interval"""
    rng = np.random.default_rng(606)
    certified = declined_feasible = probes = 0
    for trial in range(3000):
        nJ = int(rng.integers(1, 5))
        A = 10.0 ** rng.uniform(-1, 2, nJ)
        v = rng.normal(size=nJ) * 10.0 ** rng.uniform(-1, 1)
        D = rng.normal(size=nJ) * 10.0 ** rng.uniform(-1, 2)
        thr_v, thr_a = 1e-6, 1e-6
        sddot_max = 10.0 ** rng.uniform(2, 8)
        n = nJ
        a1 = rng.normal(size=n) * 10.0 ** rng.uniform(-1, 1)
        a2 = rng.normal(size=n) * 10.0 ** rng.uniform(-2, 1)
        a4 = rng.normal(size=n) * 10.0
        # tension-like limits [tmin, tmax] that a4 (gravity) may or may not fit into: sometimes infeasible at rest by a wide margin,
        # sometimes barely, sometimes feasible
        tmin = np.full(n, 1.0); tmax = np.full(n, 12.0)
        shift = rng.choice([0.0, 0.0, 5.0, 20.0, 1e-9, 1e-13])
        if rng.random() < 0.7:
            k = rng.integers(0, n)
            a4[k] = (tmax[k] + shift) if rng.random() < 0.5 else (tmin[k] - shift)    # a1 * sddot must then make up for it ...
            j = (k + 1) % n
            if n > 1 and rng.random() < 0.8:
                # ... while another row with the opposite sign of a1 and no slack forbids that direction
                a1[j] = -np.sign(a1[k]) * abs(a1[j]) if rng.random() < 0.5 else np.sign(a1[k]) * abs(a1[j])
                a4[j] = (tmin[j] - shift) if a4[k] > tmax[k] - 1e-12 else (tmax[j] + shift)
        trq = (a1, a2, a4, tmax, tmin)
        c0 = 10.0 ** rng.uniform(-2, 1)
        chk = lambda c: _real_check(c * c, c, A, v, D, thr_v, thr_a, sddot_max, trq)
        if not chk(c0):
            continue                                  # the certificate is only consulted after a violated first check
        cert = _certain_failure(rng, c0 * c0, A, v, D, thr_v, thr_a, sddot_max, trq)
        # the candidates of the reference's loop when every check is violated
        cands, low, c = [c0], 0.01, c0
        for _ in range(100):
            low *= 2.0
            lo = max(0.0, (1.0 - low) * c)
            c = 0.5 * (c + lo)
            cands.append(c)
        extra = list(c0 * rng.uniform(0, 1, 40)) + [0.0, 1e-300, 1e-160]
        feasible_somewhere = any(not chk(x) for x in cands + extra)
        if cert:
            certified += 1
            for x in cands + extra:
                probes += 1
                assert chk(x), ("certified failure, but this speed passes the check", trial, x, c0)
        elif not feasible_somewhere:
            pass                                      # declining is always allowed (the hundred checks run)
        else:
            declined_feasible += 1
    assert certified > 300 and declined_feasible > 100 and probes > 40000, (certified, declined_feasible, probes)
