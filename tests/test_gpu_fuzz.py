"""GPU (-m gpu): randomised parity sweeps -- many small random problems through the HIP library and the oracle, bit for
bit.  Seeds are fixed; the point is breadth (limits, path shapes, lengths, lane layouts) beyond the golden cases."""
import os

import numpy as np
import pytest

from helpers import assert_bit_equal, set_layout
from batotp_amd import capi, pathgen

pytestmark = pytest.mark.gpu

# BATOTP_FUZZ_SCALE=k runs k times as many seeds (the default keeps the GPU suite short)
_SCALE = max(1, int(os.environ.get("BATOTP_FUZZ_SCALE", "1")))


def _random_knots(rng, n_joints, n, scale):
    """smooth random joint paths: a few random Fourier modes per joint, plus a straight segment and a cusp now and then"""
    s = np.linspace(0.0, 1.0, n)
    y = np.zeros((n_joints, n))
    for j in range(n_joints):
        for k in range(1, 5):
            y[j] += rng.normal() / k * np.sin(2 * np.pi * k * s * rng.uniform(0.3, 2.0) + rng.uniform(0, 6.28))
    if rng.random() < 0.3:
        a = rng.integers(n // 4, n // 2)
        y[:, a:a + n // 8] = y[:, a:a + 1] + np.linspace(0, 1, n // 8)[None, :] * rng.normal(size=(n_joints, 1)) * 0.1
    if rng.random() < 0.3:
        y[rng.integers(0, n_joints)] += 0.3 * np.abs(s - rng.uniform(0.2, 0.8))
    return np.ascontiguousarray(scale * y)


def _run(ctx, prob, ys, sres, cap):
    b = capi.Batch(ctx, prob, [y.shape[1] for y in ys], cap)
    for k, y in enumerate(ys):
        b.upload_knots(k, [y], [sres[k]])
    b.precompute(1)
    b.pointwise_mvc()
    b.sweep(-1)
    b.sweep(+1)
    res = b.results()
    out = [(b.curve(k, -1), b.curve(k, +1), np.stack(b.mvc(k))) for k in range(len(ys))]
    b.close()
    return res, out


@pytest.mark.parametrize("seed", range(6 * _SCALE))
@pytest.mark.parametrize("lanes", [0, 8, "flat0", "flat4", 4, 2, "g4flat0", "g4flat8", "g2flat3", 64, "64noff", "64x2"])
def test_random_velocity_acceleration_problems(hip_lib, oracle_ctx, seed, lanes):
    rng = np.random.default_rng(1000 + seed)
    nJ = int(rng.integers(1, 9))
    vmax = list(rng.uniform(0.5, 8.0, nJ))
    amax = list(rng.uniform(1.0, 40.0, nJ))
    flags = capi.F_JNT_ACC_ON if rng.random() < 0.8 else 0
    prob = capi.make_problem(nJ, 0, flags=flags, jnt_vel_max=vmax, jnt_acc_max=amax, integ_res=float(rng.choice([0.004, 0.01, 0.02])),
                             max_integ_time=1e5)
    n_paths = int(rng.integers(3, 20))
    ys = [_random_knots(rng, nJ, int(rng.integers(8, 400)), rng.uniform(0.2, 3.0)) for _ in range(n_paths)]
    sres = [float(rng.uniform(0.01, 0.2)) for _ in range(n_paths)]
    ctx = capi.Context(hip_lib, 0)
    set_layout(ctx, lanes)
    cap = 60000
    ro, oo = _run(oracle_ctx, prob, ys, sres, cap)
    for extra in (0, capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES):
        p2 = capi.Problem.from_buffer_copy(bytes(prob))
        p2.flags |= extra
        rh, ho = _run(ctx, p2, ys, sres, cap)
        for f in rh.dtype.names:
            assert np.array_equal(rh[f], ro[f]), (seed, f)
        for k in range(n_paths):
            for which in (0, 1):
                assert_bit_equal(ho[k][which][0], oo[k][which][0], f"seed {seed} path {k} curve {which} s")
                assert_bit_equal(ho[k][which][1], oo[k][which][1], f"seed {seed} path {k} curve {which} sdot")
            assert_bit_equal(ho[k][2], oo[k][2], f"seed {seed} path {k} pointwise")


def _hard_problem(seed):
    """velocity / acceleration-only problems built to sit near the edges of the fast-forward certificate (see the test below)"""
    rng = np.random.default_rng(7000 + seed)
    nJ = int(rng.integers(2, 9))
    n_paths = int(rng.integers(4, 12))
    decades = rng.uniform(-3, 3, nJ)
    vmax = list(10.0 ** rng.uniform(-1, 1.5, nJ))
    amax = list(10.0 ** decades)
    prob = capi.make_problem(nJ, 0, flags=capi.F_JNT_ACC_ON, jnt_vel_max=vmax, jnt_acc_max=amax,
                             integ_res=float(rng.choice([0.002, 0.005, 0.02])), max_integ_time=1e5)
    ys = []
    for _ in range(n_paths):
        n = int(rng.integers(16, 300))
        y = _random_knots(rng, nJ, n, rng.uniform(0.2, 3.0))
        kind = rng.integers(0, 4)
        if kind == 0:      # a joint that barely moves: theta' around the threshold
            y[rng.integers(0, nJ)] *= 10.0 ** rng.uniform(-9, -4)
        elif kind == 1:    # two joints with the same shape (nearly parallel constraint lines)
            a, b = rng.integers(0, nJ, 2)
            y[b] = y[a] * (1.0 + 10.0 ** rng.uniform(-12, -3))
        elif kind == 2:    # a joint that stands still exactly on part of the path
            j = rng.integers(0, nJ)
            y[j, n // 3: 2 * n // 3] = y[j, n // 3]
        ys.append(np.ascontiguousarray(y))
    sres = [float(rng.uniform(0.01, 0.2)) for _ in range(n_paths)]
    cap = 80000
    return prob, ys, sres, cap, n_paths


@pytest.mark.parametrize("seed", range(16 * _SCALE))
def test_certified_fast_forward_of_the_bisection_on_hard_problems(hip_lib, oracle_ctx, seed):
    """k_sweep1 skips the checks of the bisection iterations whose outcome is certain (sweep1.hip.h): problems built to sit
    near the edges of that certificate -- joints that almost stand still (theta' just above / below jntThresh, huge
    a_q = amax / |theta'|), limits spread over six decades, joints that share one shape (pairs of constraint lines that are
    nearly parallel), all eight joints in use -- against the oracle, bit for bit"""
    prob, ys, sres, cap, n_paths = _hard_problem(seed)
    ro, oo = _run(oracle_ctx, prob, ys, sres, cap)
    # (round 6: the batch kernel k_sweep8 carries the certificate in both directions -- the forward sweep in its check block, the
    #  reverse sweep as a phase of its own that serves the paths gathered in it: flatKcH = hold K of the stage, hold H of that phase)
    for layout in (64, "64noff", "64x2", "flat4", "flat4c1", "flat8c8", "flat0c2", "flat4c0"):
        ctx = capi.Context(hip_lib, 0)
        set_layout(ctx, layout)
        p2 = capi.Problem.from_buffer_copy(bytes(prob))
        p2.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
        rh, ho = _run(ctx, p2, ys, sres, cap)
        for f in rh.dtype.names:
            assert np.array_equal(rh[f], ro[f]), (seed, layout, f)
        for k in range(n_paths):
            for which in (0, 1):
                assert_bit_equal(ho[k][which][0], oo[k][which][0], f"seed {seed} {layout} path {k} curve {which} s")
                assert_bit_equal(ho[k][which][1], oo[k][which][1], f"seed {seed} {layout} path {k} curve {which} sdot")
        ctx.close()


@pytest.mark.parametrize("seed", range(8 * _SCALE))
def test_per_knot_evaluation_kernels_agree_on_hard_problems(hip_lib, oracle_ctx, seed):
    """K3 of velocity / acceleration-only problems has a kernel of its own (pointwise_va.hip.h: shared reciprocals, select-form
    passes, the certified fast-forward on every lane at once).  On the problems built for the edges of the certificate -- and
    with the general kernel, with and without the fast-forward, on coefficient rows and on compact pairs -- the three values of
    every knot (sdot_max, sddot_L, sddot_H; NaN bounds where no speed is admissible) are the oracle's, bit for bit"""
    prob, ys, sres, cap, n_paths = _hard_problem(seed)
    if seed % 4 == 3:
        prob.jnt_acc_max[0] = -1.0     # every bisection of a knot where joint 0 moves fails: NaN bounds
    def per_knot(ctx, p):
        b = capi.Batch(ctx, p, [y.shape[1] for y in ys], 64)
        b.upload_knots(0, ys, sres)
        b.precompute(0)
        b.pointwise_mvc()
        out = [np.stack(b.mvc(k)) for k in range(n_paths)]
        b.close()
        return out

    oo = per_knot(oracle_ctx, prob)
    if seed % 4 == 3:
        assert any(np.isnan(o[1]).any() for o in oo)
    for form, ff, compact in ((1, 1, True), (1, 0, True), (0, 1, True), (1, 1, False), (1, 0, False)):
        ctx = capi.Context(hip_lib, 0)
        ctx.set_k3_form(form)
        ctx.set_fast_forward(ff)
        p2 = capi.Problem.from_buffer_copy(bytes(prob))
        if compact:
            p2.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
        ho = per_knot(ctx, p2)
        for k in range(n_paths):
            assert_bit_equal(ho[k], oo[k], f"seed {seed} form {form} ff {ff} compact {compact} path {k}: per-knot values")
        ctx.close()


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_random_taught_paths_through_the_resampler_and_output_stage(hip_ctx, oracle_ctx, seed):
    rng = np.random.default_rng(5000 + seed)
    nJ = int(rng.integers(2, 8))
    prm = capi.ResampleParams()
    prm.n_joints, prm.n_cart, prm.robot_type, prm.path_type, prm.scale_type = nJ, 3, capi.ROBOT_GENJNT, capi.PATH_JOINT, 1
    prm.s_weights[0], prm.s_weights[1], prm.s_weights[2] = 0.0, 1.0, 0.0
    res = float(rng.uniform(0.02, 0.2))
    prm.theta_norm_res, prm.theta_norm_res2 = res, res * float(rng.choice([1.0, 0.5, 2.0]))
    prm.cart_norm_res = prm.cart_norm_res2 = 0.02
    prm.jnt_thresh = prm.cart_thresh = 1e-6
    prm.input_decim_fact = int(rng.choice([1, 1, 2, 3]))
    prm.smooth_window = int(rng.choice([1, 3]))
    xs = []
    for _ in range(int(rng.integers(2, 9))):
        n = int(rng.integers(30, 1500))
        th = _random_knots(rng, nJ, n, rng.uniform(0.5, 3.0)).astype(np.float32).astype(np.float64)
        if rng.random() < 0.4:   # repeated taught points
            idx = np.sort(np.concatenate([np.arange(n), rng.integers(0, n, n // 10)]))
            th = th[:, idx]
        xs.append(np.vstack([th, np.zeros((3, th.shape[1]))]))
    sr = [0.01] * len(xs)
    h = capi.Resampled(hip_ctx, prm, xs, sr)
    o = capi.Resampled(oracle_ctx, prm, xs, sr)
    assert np.array_equal(h.status, o.status) and np.array_equal(h.n_knots, o.n_knots) and h.sres.tobytes() == o.sres.tobytes()
    knots = [o.knots(k) for k in range(len(xs))]
    for k in range(len(xs)):
        assert_bit_equal(h.knots(k), knots[k], f"seed {seed} path {k} knots")
    good = [k for k in range(len(xs)) if int(o.status[k]) == 0]
    if not good:
        return
    prob = capi.make_problem(nJ, 3, flags=capi.F_JNT_ACC_ON, jnt_vel_max=[float(rng.uniform(1, 6))] * nJ,
                             jnt_acc_max=[float(rng.uniform(2, 30))] * nJ, integ_res=0.01, max_integ_time=1e5)
    outs = []
    for ctx in (hip_ctx, oracle_ctx):
        b = capi.Batch(ctx, prob, [knots[k].shape[1] for k in good], 80000)
        for q, k in enumerate(good):
            b.upload_knots(q, [knots[k]], [float(o.sres[k])])
        b.optimize()
        per = []
        for out_res, smooth in ((0.008, 5.0), (0.01, 1.0), (0.02, 3.0)):
            out = capi.Output(b, capi.OutputParams(nJ, capi.PATH_JOINT, 0.01, out_res, smooth), 0, len(good))
            per.append((out.n_pts.copy(), [out.rows(q) for q in range(len(good))]))
            out.close()
        outs.append(per)
        b.close()
    for (nh, rh), (no, ro) in zip(*outs):
        assert np.array_equal(nh, no)
        for q in range(len(good)):
            assert_bit_equal(rh[q], ro[q], f"seed {seed} output of path {good[q]}")


@pytest.mark.parametrize("par2ser", [0, 1])
@pytest.mark.parametrize("seed", range(5 * _SCALE))
def test_random_cable_robot_problems(hip_lib, oracle_ctx, seed, par2ser):
    """3-cable robot: cable tension limits (parallel-mechanism torque branch or its serial conversion), cable velocity /
    acceleration limits, Cartesian speed limit; random platform paths inside the workspace"""
    import helpers
    rng = np.random.default_rng(9000 + seed)
    base = helpers.Case("synth_cspr_s3").problem
    prob = capi.Problem.from_buffer_copy(bytes(base))
    flags = capi.F_TRQ_ON | capi.F_PARALLEL | capi.F_JNT_ACC_ON
    if par2ser:
        flags |= capi.F_PAR2SER
    if rng.random() < 0.7:
        flags |= capi.F_CART_VEL_ON
    prob.flags = flags
    for j in range(3):
        prob.jnt_vel_max[j] = float(rng.uniform(2, 6)); prob.jnt_acc_max[j] = float(rng.uniform(4, 12))
        prob.jnt_trq_max[j] = float(rng.uniform(10, 16)); prob.jnt_trq_min[j] = float(rng.uniform(0.5, 1.5))
    prob.cart_vel_max = float(rng.uniform(2, 5))
    pm = np.array(list(prob.pmat)).reshape(3, 3)
    ys, sres = [], []
    for _ in range(int(rng.integers(2, 7))):
        n = int(rng.integers(40, 500))
        t = np.linspace(0, 1, n)
        cart = np.stack([1.0 * np.sin(2 * np.pi * t * rng.uniform(0.3, 1.5) + rng.uniform(0, 6)),
                         1.0 * np.cos(2 * np.pi * t * rng.uniform(0.3, 1.5) + rng.uniform(0, 6)) + 0.4,
                         3.0 + 0.8 * np.sin(2 * np.pi * t * rng.uniform(0.2, 1.0) + rng.uniform(0, 6))])
        theta = np.stack([np.sqrt(((cart - pm[:, k:k + 1]) ** 2).sum(axis=0)) for k in range(3)])
        ys.append(np.ascontiguousarray(np.vstack([theta, cart])))
        sres.append(float(rng.uniform(0.005, 0.03)))
    # 64 / "64noff": one path per wavefront with and without the certified fast-forward of the bisection (which covers the
    # serial form's torque lines: a3 = 0)
    def run(c):
        b = capi.Batch(c, prob, [y.shape[1] for y in ys], 4000)
        for k, y in enumerate(ys):
            b.upload_knots(k, [y], [sres[k]])
        b.precompute(0); b.pointwise_mvc(); b.sweep(-1); b.sweep(+1)
        out = (b.results(), [(b.curve(k, -1), b.curve(k, +1), np.stack(b.mvc(k))) for k in range(len(ys))])
        b.close()
        return out

    oracle_out = run(oracle_ctx)      # once: the layouts below are all compared with the same oracle run
    for lanes in (0, 8, 64, "64noff"):
        ctx = capi.Context(hip_lib, 0)
        set_layout(ctx, lanes)
        outs = [run(ctx), oracle_out]
        (rh, ho), (ro, oo) = outs
        for f in rh.dtype.names:
            assert np.array_equal(rh[f], ro[f]), (seed, f, rh[f], ro[f])
        for k in range(len(ys)):
            for which in (0, 1):
                assert_bit_equal(ho[k][which][0], oo[k][which][0], f"seed {seed} path {k} curve {which} s")
                assert_bit_equal(ho[k][which][1], oo[k][which][1], f"seed {seed} path {k} curve {which} sdot")
            assert_bit_equal(ho[k][2], oo[k][2], f"seed {seed} path {k} pointwise")


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_cable_robot_paths_the_tension_limits_do_not_admit(hip_lib, oracle_ctx, seed):
    """the cable robot in serial form on paths that leave the region where the tensions can stay inside [tmin, tmax]: there every
    stage's bisection fails -- the reference halves the speed a hundred times, returns -1 without touching sddot and integrates on
    (ba.cpp:1307-1319, :1091).  k_sweep1 certifies such a stage (no speed in [0, first candidate] can pass a check: sweep1.hip.h) and
    skips the hundred checks.  Rows incl. the failure counts, curves and pointwise values against the oracle: fast-forward on / off,
    coefficient rows and every channel as pairs, one and two paths per wavefront; the paths crawl, so some end by capacity"""
    import helpers
    rng = np.random.default_rng(9500 + seed)
    base = helpers.Case("synth_cspr_s3").problem
    prob = capi.Problem.from_buffer_copy(bytes(base))
    prob.flags = capi.F_TRQ_ON | capi.F_PARALLEL | capi.F_PAR2SER | capi.F_JNT_ACC_ON | (capi.F_CART_VEL_ON if seed % 2 else 0)
    for j in range(3):
        prob.jnt_vel_max[j] = float(rng.uniform(2, 6)); prob.jnt_acc_max[j] = float(rng.uniform(4, 12))
        prob.jnt_trq_max[j] = float(rng.uniform(8, 12)); prob.jnt_trq_min[j] = float(rng.uniform(1.0, 2.5))   # a narrow band
    prob.cart_vel_max = float(rng.uniform(2, 5))
    pm = np.array(list(prob.pmat)).reshape(3, 3)
    ys, sres = [], []
    for k in range(5):
        n = int(rng.integers(60, 400))
        t = np.linspace(0, 1, n)
        # the platform wanders towards the edge of the anchor triangle and low under it, where one cable must go slack
        amp = (1.0, 1.4, 1.8, 2.1, 2.4)[k]
        cart = np.stack([amp * np.sin(2 * np.pi * t * rng.uniform(0.3, 1.2) + rng.uniform(0, 6)),
                         amp * np.cos(2 * np.pi * t * rng.uniform(0.3, 1.2) + rng.uniform(0, 6)) + 0.4,
                         3.0 + 0.8 * np.sin(2 * np.pi * t * rng.uniform(0.2, 1.0) + rng.uniform(0, 6))])
        theta = np.stack([np.sqrt(((cart - pm[:, q:q + 1]) ** 2).sum(axis=0)) for q in range(3)])
        ys.append(np.ascontiguousarray(np.vstack([theta, cart])))
        sres.append(float(rng.uniform(0.005, 0.03)))

    def run(c, pr):
        b = capi.Batch(c, pr, [y.shape[1] for y in ys], 3000)
        for k, y in enumerate(ys):
            b.upload_knots(k, [y], [sres[k]])
        b.precompute(0); b.pointwise_mvc()
        mv = [np.stack(b.mvc(k)) for k in range(len(ys))]
        b.sweep(-1); b.sweep(+1)
        res = b.results()
        cur = [(b.curve(k, -1), b.curve(k, +1)) if not ((res[k]["status_rev"] | res[k]["status_fwd"]) & ~np.uint32(capi.ST_BISECT_FAIL)) else None
               for k in range(len(ys))]
        b.close()
        return res, cur, mv

    ro, co, mo = run(oracle_ctx, prob)
    fails = int(ro["n_bisect_fail_rev"].sum() + ro["n_bisect_fail_fwd"].sum())
    assert fails > 1000, f"seed {seed}: the problem is meant to fail many bisections ({fails})"
    pairs = capi.Problem.from_buffer_copy(bytes(prob))
    pairs.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
    for lanes, pr in ((64, prob), ("64noff", prob), (64, pairs), ("64x2", pairs), ("64noff", pairs), (0, prob)):
        ctx = capi.Context(hip_lib, 0)
        set_layout(ctx, lanes)
        rh, ch, mh = run(ctx, pr)
        for f in rh.dtype.names:
            assert np.array_equal(rh[f], ro[f]), (seed, lanes, f, rh[f], ro[f])
        for k in range(len(ys)):
            assert_bit_equal(mh[k], mo[k], f"seed {seed} {lanes} path {k} pointwise")
            if co[k] is not None:
                for which in (0, 1):
                    assert_bit_equal(ch[k][which][0], co[k][which][0], f"seed {seed} {lanes} path {k} curve {which} s")
                    assert_bit_equal(ch[k][which][1], co[k][which][1], f"seed {seed} {lanes} path {k} curve {which} sdot")
        ctx.close()


@pytest.mark.parametrize("seed", range(4 * _SCALE))
def test_random_cartesian_constraint_problems(hip_lib, oracle_ctx, seed):
    """joint limits plus Cartesian speed and / or acceleration limits (solveQuadratic branch), random Cartesian channels"""
    rng = np.random.default_rng(7000 + seed)
    nJ = int(rng.integers(2, 8))
    flags = capi.F_JNT_ACC_ON | (capi.F_CART_VEL_ON if seed % 2 == 0 else 0) | (capi.F_CART_ACC_ON if seed % 3 != 2 else 0)
    if not (flags & (capi.F_CART_VEL_ON | capi.F_CART_ACC_ON)):
        flags |= capi.F_CART_VEL_ON
    prob = capi.make_problem(nJ, 3, flags=flags, jnt_vel_max=list(rng.uniform(1, 6, nJ)), jnt_acc_max=list(rng.uniform(2, 30, nJ)),
                             cart_vel_max=float(rng.uniform(0.3, 2.0)), cart_acc_max=float(rng.uniform(0.5, 5.0)), integ_res=0.01,
                             max_integ_time=1e5)
    ys, sres = [], []
    for _ in range(int(rng.integers(2, 12))):
        n = int(rng.integers(10, 350))
        th = _random_knots(rng, nJ, n, rng.uniform(0.3, 2.0))
        ca = _random_knots(rng, 3, n, rng.uniform(0.1, 1.0))
        if rng.random() < 0.25:
            ca[:, n // 3: n // 2] = ca[:, n // 3: n // 3 + 1]   # the tool point stands still for a while (quadratic degenerates)
        ys.append(np.ascontiguousarray(np.vstack([th, ca])))
        sres.append(float(rng.uniform(0.01, 0.1)))
    ro, oo = _run(oracle_ctx, prob, ys, sres, 30000)
    for lanes in (0, 1, 2, 4, 8, 16):
        ctx = capi.Context(hip_lib, 0)
        ctx.set_sweep_group(lanes)
        rh, ho = _run(ctx, prob, ys, sres, 30000)
        for f in rh.dtype.names:
            assert np.array_equal(rh[f], ro[f]), (seed, lanes, f)
        for k in range(len(ys)):
            for which in (0, 1):
                assert_bit_equal(ho[k][which][0], oo[k][which][0], f"seed {seed} lanes {lanes} path {k} curve {which} s")
                assert_bit_equal(ho[k][which][1], oo[k][which][1], f"seed {seed} lanes {lanes} path {k} curve {which} sdot")
            assert_bit_equal(ho[k][2], oo[k][2], f"seed {seed} lanes {lanes} path {k} pointwise")


@pytest.mark.parametrize("seed", range(3 * _SCALE))
def test_random_two_link_arm_with_torque_limits(hip_lib, oracle_ctx, seed):
    """RR arm: torque limits through the serial dynamics (a1..a4 splines), with and without joint acceleration limits;
    the trigonometric terms are uploaded (BATOTP_F_HOST_TRIG), the same arrays to both implementations"""
    import helpers
    rng = np.random.default_rng(3000 + seed)
    base = helpers.Case("RR").problem
    prob = capi.Problem.from_buffer_copy(bytes(base))
    prob.flags = capi.F_TRQ_ON | capi.F_HOST_TRIG | (capi.F_JNT_ACC_ON if seed % 2 else 0)
    for j in range(2):
        prob.jnt_vel_max[j] = float(rng.uniform(100, 400)); prob.jnt_acc_max[j] = float(rng.uniform(500, 2000))
        prob.jnt_trq_max[j] = float(rng.uniform(5, 40)); prob.jnt_trq_min[j] = -float(rng.uniform(5, 40))
    ys = [_random_knots(rng, 2, int(rng.integers(20, 400)), rng.uniform(20, 120)) for _ in range(int(rng.integers(2, 9)))]
    ys = [np.ascontiguousarray(np.vstack([y, np.zeros((prob.n_cart, y.shape[1]))])) for y in ys]
    sres = [float(rng.uniform(0.2, 2.0)) for _ in ys]
    outs = []
    # 64 / "64noff": one path per wavefront with and without the certified fast-forward of the bisection (general form: a3 != 0)
    for lanes, c in ((0, None), (8, None), (4, None), (2, None), (64, None), ("64noff", None), (-1, oracle_ctx)):
        ctx = c
        if ctx is None:
            ctx = capi.Context(hip_lib, 0)
            set_layout(ctx, lanes)
        b = capi.Batch(ctx, prob, [y.shape[1] for y in ys], 12000)
        for k, y in enumerate(ys):
            b.upload_knots(k, [y], [sres[k]])
        b.precompute(1)
        for k in range(len(ys)):
            b.upload_rr_trig(k, helpers.rr_trig(b.samples(k, 0)[0], b.samples(k, 1)[0]))
        b.precompute(2)
        b.pointwise_mvc(); b.sweep(-1); b.sweep(+1)
        outs.append((b.results(), [(b.curve(k, -1), b.curve(k, +1), np.stack(b.mvc(k)),
                                    np.stack([np.stack([b.dyn(k, kk, r) for r in range(2)]) for kk in (1, 2, 3, 4)])) for k in range(len(ys))]))
        b.close()
    ro, oo = outs[-1]
    for rh, ho in outs[:-1]:
        for f in rh.dtype.names:
            assert np.array_equal(rh[f], ro[f]), (seed, f)
        for k in range(len(ys)):
            assert_bit_equal(ho[k][3], oo[k][3], f"seed {seed} path {k} dynamics coefficients")
            for which in (0, 1):
                assert_bit_equal(ho[k][which][0], oo[k][which][0], f"seed {seed} path {k} curve {which} s")
                assert_bit_equal(ho[k][which][1], oo[k][which][1], f"seed {seed} path {k} curve {which} sdot")
            assert_bit_equal(ho[k][2], oo[k][2], f"seed {seed} path {k} pointwise")


@pytest.mark.parametrize("ppw,hold", [(1, -1), (8, -1), (8, 4)])
def test_two_link_arm_paths_that_never_finish(hip_lib, oracle_ctx, ppw, hold):
    """torque-limited paths that stall (the sweep runs into its step capacity with hundreds to thousands of failed
    bisections, one path failing at every stage): failure counts, step counts and statuses equal the oracle's, with one
    and with several such paths per wavefront.  (This is the case that exposed the hold-dependent results of the flat
    stage / bisection loop when it was instantiated for the torque branch, tools/experiments/; a problem with torque
    limits runs the nested loops whatever hold is set: the last parameter set checks exactly that.)"""
    import helpers
    rng = np.random.default_rng(3000)
    base = helpers.Case("RR").problem
    prob = capi.Problem.from_buffer_copy(bytes(base))
    prob.flags = capi.F_TRQ_ON | capi.F_HOST_TRIG
    for j in range(2):
        prob.jnt_vel_max[j] = float(rng.uniform(100, 400)); prob.jnt_acc_max[j] = float(rng.uniform(500, 2000))
        prob.jnt_trq_max[j] = float(rng.uniform(5, 40)); prob.jnt_trq_min[j] = -float(rng.uniform(5, 40))
    ys = [_random_knots(rng, 2, int(rng.integers(20, 400)), rng.uniform(20, 120)) for _ in range(int(rng.integers(2, 9)))]
    ys = [np.ascontiguousarray(np.vstack([y, np.zeros((prob.n_cart, y.shape[1]))])) for y in ys]
    sres = [float(rng.uniform(0.2, 2.0)) for _ in ys]
    res = []
    for c in (None, oracle_ctx):
        ctx = c
        if ctx is None:
            ctx = capi.Context(hip_lib, 0)
            ctx.set_sweep_group(8)
            ctx.set_paths_per_wave(ppw)
            ctx.set_sweep_hold(hold, hold)
        b = capi.Batch(ctx, prob, [y.shape[1] for y in ys], 12000)
        for k, y in enumerate(ys):
            b.upload_knots(k, [y], [sres[k]])
        b.precompute(1)
        for k in range(len(ys)):
            b.upload_rr_trig(k, helpers.rr_trig(b.samples(k, 0)[0], b.samples(k, 1)[0]))
        b.precompute(2)
        b.sweep(-1); b.sweep(+1)
        res.append(b.results())
        b.close()
    rh, ro = res
    assert int(ro["n_bisect_fail_rev"].max()) > 1000 and int((ro["status_rev"] & capi.ST_CAPACITY != 0).sum()) >= 3
    for f in rh.dtype.names:
        assert np.array_equal(rh[f], ro[f]), (f, rh[f], ro[f])


@pytest.mark.parametrize("compact", [0, 1])
@pytest.mark.parametrize("seed", range(2 * _SCALE))
def test_flat_sweep_loop_on_stalled_velocity_acceleration_paths(hip_lib, oracle_ctx, seed, compact):
    """joint velocity / acceleration limits only, one joint with a NEGATIVE acceleration limit: wherever that joint moves
    no sddot is admissible, so some paths fail at every stage and run into the step capacity, some fail now and then (the
    joint hovers around the zero-velocity threshold), one never does -- seven such paths in one wavefront, nested loops
    and flat loop with every hold: everything equal to the oracle"""
    rng = np.random.default_rng(500 + seed)
    nJ = 3
    amax = [float(rng.uniform(5, 30)), -1.0, float(rng.uniform(5, 30))]
    prob = capi.make_problem(nJ, 0, flags=capi.F_JNT_ACC_ON, jnt_vel_max=list(rng.uniform(1, 6, nJ)), jnt_acc_max=amax,
                             integ_res=0.01, max_integ_time=1e5)
    ys = [_random_knots(rng, nJ, int(rng.integers(40, 300)), rng.uniform(0.5, 2.0)) for _ in range(7)]
    ys[3][1, :] = 0.25
    for k, f in ((0, 3e-6), (1, 1e-5), (2, 3e-5), (4, 1e-4)):
        ys[k][1] *= f
    sres = [float(rng.uniform(0.02, 0.1)) for _ in ys]
    cap = 3000
    ro, oo = _run(oracle_ctx, prob, ys, sres, cap)
    assert int(ro["n_bisect_fail_rev"].max()) > 10000 and int(ro["n_bisect_fail_rev"].min()) == 0
    p2 = capi.Problem.from_buffer_copy(bytes(prob))
    if compact:
        p2.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
    # (hold of the stage, hold of the reverse sweep's certificate phase: -1 automatic, 0 none)
    for hold, cert in ((-1, -1), (0, -1), (2, -1), (3, -1), (4, -1), (5, -1), (6, -1), (8, -1), (4, 0), (4, 1), (4, 8), (6, 2)):
        ctx = capi.Context(hip_lib, 0)
        ctx.set_sweep_group(8); ctx.set_paths_per_wave(8); ctx.set_sweep_hold(hold, hold); ctx.set_cert_hold(cert)
        rh, ho = _run(ctx, p2, ys, sres, cap)
        for f in rh.dtype.names:
            assert np.array_equal(rh[f], ro[f]), (seed, hold, cert, f, rh[f], ro[f])
        for k in range(len(ys)):
            for which in (0, 1):
                assert_bit_equal(ho[k][which][0], oo[k][which][0], f"seed {seed} hold {hold} path {k} curve {which} s")
                assert_bit_equal(ho[k][which][1], oo[k][which][1], f"seed {seed} hold {hold} path {k} curve {which} sdot")
        ctx.close()
