"""GPU (-m gpu): the HIP kernels against the oracle, through the C-ABI, bit for bit.

fp64 parity contract: every value the hot path publishes (spline coefficients, knot samples,
dynamics coefficients, pointwise sdot_max / sddot interval, both integrated curves, step counts,
traversal time, status) is IDENTICAL to the oracle's -- tolerance 0 -- and, rounded to float32 the
way the reference writes s-sdot.dat, identical to the reference binary's own output.
"""
import os

import numpy as np
import pytest

import helpers
from helpers import Case, run_pipeline, assert_matches_reference, assert_bit_equal
from batotp_amd import capi

pytestmark = pytest.mark.gpu


def test_fp64_div_sqrt_are_correctly_rounded_and_not_contracted(hip_ctx):
    rng = np.random.default_rng(12345)
    n = 1 << 16
    a = np.abs(rng.standard_normal(n)) * 10.0 ** rng.integers(-30, 30, n)
    b = rng.standard_normal(n) * 10.0 ** rng.integers(-30, 30, n)
    b[b == 0] = 1.0
    # a few hard cases
    a[:4] = [1.0, 2.0, 0.1, 1e-310]
    b[:4] = [3.0, 3.0, 0.3, 7.0]
    q, r, p = hip_ctx.fp64_kat(a, b)
    assert_bit_equal(q, a / b, "fp64 division")
    assert_bit_equal(r, np.sqrt(a), "fp64 sqrt")
    assert_bit_equal(p, (a * b) + (a / b), "a*b+q must be a separate multiply and add (no FMA contraction)")


def test_division_by_six_through_the_reciprocal_is_the_ieee_quotient(hip_ctx):
    """the compact spline form divides by 6 with a reciprocal and one exact residual correction"""
    rng = np.random.default_rng(777)
    n = 1 << 20
    a = rng.standard_normal(n) * 10.0 ** rng.integers(-40, 40, n)
    # numerators next to exact multiples of 6 and to rounding boundaries of the quotient
    k = rng.integers(1, 1 << 52, 1 << 16).astype(np.float64)
    near = np.concatenate([6.0 * k, np.nextafter(6.0 * k, np.inf), np.nextafter(6.0 * k, -np.inf), k, k + 0.5, 3.0 * k])
    edge = np.array([0.0, -0.0, 6.0, -6.0, 1e-300, -1e-300, 1e300, 5e-324, 1.7e308, 1e-290, 1e290])
    a = np.concatenate([a, near, edge])
    assert_bit_equal(capi.div6_kat(hip_ctx, a), a / 6.0, "x / 6.0")


def test_shared_reciprocal_division_is_the_ieee_quotient(hip_ctx):
    """k_sweep8 divides the bounds of a constraint check and the velocity limit by theta' through ONE refined reciprocal
    (the last three operations of the hardware's own division sequence) when both operands lie in [2^-350, 2^350]"""
    rng = np.random.default_rng(4242)
    n = 1 << 21
    a = rng.standard_normal(n) * 2.0 ** rng.integers(-360, 360, n)
    b = rng.standard_normal(n) * 2.0 ** rng.integers(-360, 360, n)
    # mantissa patterns that stress the rounding of the quotient: all-ones divisors, exact quotients and their neighbours
    k = rng.integers(1, 1 << 52, 1 << 18).astype(np.float64)
    m = rng.integers(1, 1 << 26, 1 << 18).astype(np.float64)
    ones = np.nextafter(2.0 ** rng.integers(-300, 300, 1 << 18).astype(np.float64), 0.0)
    a2 = np.concatenate([k * m, np.nextafter(k * m, np.inf), np.nextafter(k * m, -np.inf), k, k, np.full(8, 2.0 ** -350), np.full(8, 2.0 ** 350)])
    b2 = np.concatenate([m, m, m, ones, np.nextafter(ones, -np.inf),
                         np.array([2.0 ** -350, 2.0 ** 350, np.nextafter(2.0 ** -350, 0), np.nextafter(2.0 ** 350, np.inf), 1.0, 3.0, -7.0, 1e-300]),
                         np.array([2.0 ** -350, 2.0 ** 350, np.nextafter(2.0 ** -350, 0), np.nextafter(2.0 ** 350, np.inf), 1.0, 3.0, -7.0, 1e300])])
    edge = np.array([0.0, -0.0, 1e-320, 1e308, np.inf, -np.inf, np.nan, 5.0, -5.0])
    a3, b3 = np.repeat(edge, edge.size), np.tile(edge, edge.size)
    a, b = np.concatenate([a, a2, a3]), np.concatenate([b, b2, b3])
    q, used = capi.sdiv_kat(hip_ctx, a, b)
    with np.errstate(all="ignore"):
        want = a / b
    nan = np.isnan(want)
    assert np.array_equal(np.isnan(q), nan)
    assert_bit_equal(q[~nan], want[~nan], "a / b through the shared reciprocal")
    inside = (np.abs(a) >= 2.0 ** -350) & (np.abs(a) <= 2.0 ** 350) & (np.abs(b) >= 2.0 ** -350) & (np.abs(b) <= 2.0 ** 350)
    assert np.array_equal(used, inside) and np.count_nonzero(used) > n // 2


def _compare(case, ho, oo):
    for key in ("coef", "samp", "dyn", "mvc"):
        if key in oo:
            assert_bit_equal(ho[key], oo[key], f"{case.name}:{key}")
    for which in ("rev", "fwd"):
        assert_bit_equal(ho[which][0], oo[which][0], f"{case.name}:{which}.s")
        assert_bit_equal(ho[which][1], oo[which][1], f"{case.name}:{which}.sdot")
    for f in ho["result"].dtype.names:
        assert ho["result"][f] == oo["result"][f], (case.name, f, ho["result"], oo["result"])


@pytest.mark.parametrize("name", helpers.FULL_CASES)
def test_hip_matches_oracle_and_reference(hip_ctx, oracle_ctx, name):
    case = Case(name)
    ho = run_pipeline(hip_ctx, [case])[0]
    oo = run_pipeline(oracle_ctx, [case])[0]
    _compare(case, ho, oo)
    assert_matches_reference(case, ho)


@pytest.mark.parametrize("name", helpers.SELF_CASES)
def test_serial_chain_dynamics_and_torque_limited_sweep(hip_ctx, oracle_ctx, name):
    """BASELINE config 3 (7-DOF arm with torque limits; no reference model exists: parity unpinned for the model):
    k_dyn_serial and everything downstream of it equal the oracle bit for bit -- a1..a4 at every knot, their
    splines, the per-knot bounds, both curves, step counts, T -- and reproduce the self-generated fixtures"""
    case = Case(name)
    ho = run_pipeline(hip_ctx, [case])[0]
    oo = run_pipeline(oracle_ctx, [case])[0]
    assert ho["dyn"].shape[1] == 7 and np.all(np.isfinite(ho["dyn"]))
    _compare(case, ho, oo)
    assert_matches_reference(case, ho)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16])
def test_serial_chain_other_lane_groupings_and_ragged_batch(hip_lib, oracle_ctx, lanes):
    """torque-limited 7-DOF paths of different length in one batch, every lane layout of the sweep"""
    ctx = capi.Context(hip_lib, 0)
    ctx.set_sweep_group(lanes)
    names = ["KUKA_trq_rated", "KUKA_trq_tight", "KUKA_trq_rated"]
    cases = [Case(n) for n in names]
    for c in cases:
        c.problem = cases[1].problem          # one problem per batch: the tight limits, Cartesian speed limit on
    cap = 12000
    many = run_pipeline(ctx, cases * 3, max_steps=cap, mvc=False, details=False)
    ref = [run_pipeline(oracle_ctx, [c], max_steps=cap, mvc=False, details=False)[0] for c in cases[:2]]
    for k, c in enumerate(cases * 3):
        _compare(c, many[k], ref[1 if c.name == "KUKA_trq_tight" else 0])
    ctx.close()


def test_serial_chain_without_host_trig_is_close(hip_ctx, oracle_ctx):
    """without BATOTP_F_HOST_TRIG the device libm supplies cos / sin: a1..a4 then agree with the oracle to rounding
    of the trigonometric functions, not bit for bit (the documented trig policy)"""
    case = Case("KUKA_trq_rated")
    prob = capi.Problem.from_buffer_copy(bytes(case.problem))
    prob.flags &= ~capi.F_HOST_TRIG
    model = hip_ctx.library.builtin_serial_model(capi.ROBOT_KUKA)
    b = capi.Batch(hip_ctx, prob, [case.n], case.max_steps())
    b.upload_knots(0, [case.y], [case.sres])
    b.precompute(1)
    b.set_serial_model(model)
    b.precompute(2)
    dyn = np.stack([np.stack([b.dyn(0, kk, r) for r in range(7)]) for kk in (1, 2, 3, 4)])
    b.close()
    oo = run_pipeline(oracle_ctx, [case], mvc=False)[0]["dyn"]
    scale = np.abs(oo).max(axis=(1, 2), keepdims=True)   # per coefficient family (the gravity torque about a vertical axis is 0)
    assert np.max(np.abs(dyn - oo) / scale) < 1e-13


def test_serial_chain_needs_a_model(hip_ctx):
    """a serial robot without a chain model: the reference prints "No dynamics model provided" (robot.cpp:355-357);
    the device layer refuses the dynamics stage"""
    case = Case("KUKA_trq_rated")
    b = capi.Batch(hip_ctx, case.problem, [case.n], 64)
    b.upload_knots(0, [case.y], [case.sres])
    b.precompute(1)
    with pytest.raises(capi.BatotpError):
        b.precompute(2)
    b.close()


def test_kuka_torque_limits_at_baseline_size(hip_ctx, oracle_ctx):
    """BASELINE config 3 as worded: KUKA-LWR-IV 7-DOF with torque limits, N = 100k, one trajectory: HIP == oracle"""
    import bench
    y, sres, prob, _ = bench.make_knots("kuka7trq", 14, 100000, tool=bench.ORACLE_KNOTS)      # the checker's resampler
    yp, sresp, _, _ = bench.make_knots("kuka7trq", 14, 100000)                                    # BA::interpInputData on the device
    assert_bit_equal(yp, y, "one-path device resampler (KUKA joints + tool point) vs the oracle's knots")
    assert sresp == sres
    assert 95000 < y.shape[1] < 105000 and prob.n_joints == 7 and (prob.flags & capi.F_TRQ_ON)

    class _C:
        name = "synth_kuka_s14_trq_100k"
    c = _C()
    c.y, c.sres, c.problem, c.n = y, sres, prob, y.shape[1]
    c.max_steps = lambda: int(1.5 * y.shape[1])
    ho = run_pipeline(hip_ctx, [c], details=False)[0]
    oo = run_pipeline(oracle_ctx, [c], details=False)[0]
    assert ho["result"]["status_fwd"] == 0 and ho["result"]["n_fwd"] > 1000
    _compare(c, ho, oo)


def _vel_acc_only(name):
    f = Case(name).problem.flags if name in helpers.FULL_CASES else 0
    return name in helpers.FULL_CASES and not (f & (capi.F_TRQ_ON | capi.F_CART_VEL_ON | capi.F_CART_ACC_ON))


@pytest.mark.parametrize("lanes", [0, 1, 2, 4, 8, 16, 32, 64, "64x2"])
def test_compact_splines_give_identical_results(hip_lib, oracle_ctx, lanes):
    """BATOTP_F_COMPACT_SPLINES (value + second derivative per knot instead of four coefficients): every
    published quantity is bit-identical to the oracle's, for every lane grouping of the sweep"""
    names = [n for n in helpers.FULL_CASES if _vel_acc_only(n)]
    assert len(names) >= 3, names
    ctx = capi.Context(hip_lib, 0)
    helpers.set_layout(ctx, lanes)
    for name in names:
        case = Case(name)
        ho = run_pipeline(ctx, [case], extra_flags=capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES)[0]
        oo = run_pipeline(oracle_ctx, [case])[0]
        oo.pop("samp", None)
        _compare(case, ho, oo)
        assert_matches_reference(case, ho)


def _cart_or_torque(name):
    p = Case(name).problem
    par_branch = (p.flags & capi.F_TRQ_ON) and (p.flags & capi.F_PARALLEL) and not (p.flags & capi.F_PAR2SER)
    return bool(p.flags & (capi.F_TRQ_ON | capi.F_CART_VEL_ON | capi.F_CART_ACC_ON)) and not par_branch


def test_pairs_for_all_channels_give_identical_results(hip_lib, oracle_ctx):
    """BATOTP_F_COMPACT_SPLINES on problems WITH Cartesian limits / torque limits (round 4): every channel -- joints, Cartesian,
    a1..a4 of every dynamics row -- is kept as (value, second derivative) pairs, the samples are formed inside K2, the rows inside
    K3 and in the LDS windows of the sweep kernel.  Every published quantity (the coefficient rows of ALL channels included) is
    bit-identical to the oracle's rows, and the curves to the reference binary's"""
    names = [n for n in helpers.FULL_CASES if _cart_or_torque(n)] + list(helpers.SELF_CASES)
    assert len(names) >= 8, names
    ctx = capi.Context(hip_lib, 0)
    for name in names:
        case = Case(name)
        oo = run_pipeline(oracle_ctx, [case])[0]
        ho = run_pipeline(ctx, [case], extra_flags=capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES, samples_from=[oo])[0]
        oo.pop("samp", None)
        dyn = oo.pop("dyn", None)
        _compare(case, ho, oo)
        if dyn is not None:   # the dynamics values are c0 of their channels' rows (C-ABI channel Cin + (k-1) d + r)
            cin, d = case.problem.n_joints + case.problem.n_cart, case.problem.dyn_dim
            for k in range(4):
                for r in range(d):
                    helpers.assert_bit_equal(ho["coef"][cin + k * d + r][0][:-1], dyn[k][r][:-1], f"{name}: a{k + 1} of row {r}")
        assert_matches_reference(case, ho)
    ctx.close()


@pytest.mark.parametrize("name", helpers.DIGEST_CASES)
def test_hip_baseline_size_paths(hip_ctx, oracle_ctx, name):
    case = Case(name)
    ho = run_pipeline(hip_ctx, [case], mvc=True, details=False)[0]
    oo = run_pipeline(oracle_ctx, [case], mvc=True, details=False)[0]
    _compare(case, ho, oo)
    assert_matches_reference(case, ho)


@pytest.mark.parametrize("lanes", [1, 2, 4, 8, 16, 32, 64, "flat4", "g4flat4", "g2flat5"])
def test_other_lane_groupings_agree(hip_lib, oracle_ctx, lanes):
    """the sweep kernel with 1 or 16 lanes per path publishes the same bits as the default 8"""
    ctx = capi.Context(hip_lib, 0)
    helpers.set_layout(ctx, lanes)
    for name in ("GEN7DOF", "CSPR3DOF", "UR5", "CSPR3DOF_par", "RR_acc", "synth_cspr_s5", "synth_ur_s2"):
        case = Case(name)
        ho = run_pipeline(ctx, [case], mvc=False, details=False)[0]
        oo = run_pipeline(oracle_ctx, [case], mvc=False, details=False)[0]
        _compare(case, ho, oo)
    ctx.close()


def test_ragged_batch_equals_single_paths(hip_ctx, oracle_ctx):
    """a batch of paths of different length: every path equals its single-path run"""
    names = ["synth_gen7dof_s0", "GEN7DOF", "synth_gen7dof_s0", "GEN7DOF", "GEN7DOF"]
    cases = [Case(n) for n in names]
    # same problem for the whole batch: take it from the first case (GEN7DOF example uses the same limits)
    for c in cases:
        c.problem = cases[0].problem
    many = run_pipeline(hip_ctx, cases * 5, mvc=False, details=False)   # 25 paths: several waves
    ref = {n: run_pipeline(oracle_ctx, [c], mvc=False, details=False)[0] for n, c in zip(names, cases)}
    for k, c in enumerate(cases * 5):
        _compare(c, many[k], ref[c.name])


def test_ragged_batch_with_pairs_for_all_channels(hip_lib, oracle_ctx):
    """a batch of cable-robot paths of different length with every channel as pairs (several blocks of the one-path-per-wavefront
    kernel, paths at different offsets of the pair array): every path -- per-knot bounds, both curves, the result row -- equals its
    single-path oracle run"""
    names = ["synth_cspr_s3", "CSPR3DOF", "synth_cspr_s5", "synth_cspr_s3", "synth_cspr_s9_dup", "CSPR3DOF", "synth_cspr_s5"]
    cases = [Case(n) for n in names]
    for c in cases:
        c.problem = cases[0].problem
    ctx = capi.Context(hip_lib, 0)
    cap = 2 * max(c.max_steps() for c in cases)
    many = run_pipeline(ctx, cases * 3, max_steps=cap, mvc=True, details=False, extra_flags=capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES)
    ref = {}
    for c in cases:
        if c.name not in ref:
            ref[c.name] = run_pipeline(oracle_ctx, [c], max_steps=cap, mvc=True, details=False)[0]
    for k, c in enumerate(cases * 3):
        _compare(c, many[k], ref[c.name])
    ctx.close()


@pytest.mark.parametrize("hold", [(0, 0), (3, 5), (4, -1), (5, 3), (8, 8), (-1, 6), (-1, -1), (-2, -2), (4, 8, 0), (4, 8, 1), (5, 8, 2), (0, 0, 8), (8, 8, 5)])
@pytest.mark.parametrize("compact", [False, True])
def test_flat_sweep_loop_with_paths_drifting_apart(hip_lib, oracle_ctx, hold, compact):
    """batotp_hip_set_sweep_hold: 8 different paths per wavefront in the flat stage / bisection loop (each at its own
    stage and step), every hold -- and (third number) every hold of the reverse sweep's certificate phase, batotp_hip_set_cert_hold:
    every path equals its single-path oracle run"""
    names = ["synth_gen7dof_s0", "GEN7DOF", "synth_gen7dof_s1_vel", "GEN7DOF", "synth_gen7dof_s0", "GEN7DOF", "GEN7DOF"]
    cases = [Case(n) for n in names]
    for c in cases:
        c.problem = cases[0].problem
    ctx = capi.Context(hip_lib, 0)
    ctx.set_sweep_group(8)
    ctx.set_paths_per_wave(8)
    ctx.set_sweep_hold(*hold[:2])
    if len(hold) > 2:
        ctx.set_cert_hold(hold[2])
    extra = (capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES) if compact else 0
    cap = 4 * max(c.max_steps() for c in cases)
    many = run_pipeline(ctx, cases * 3, max_steps=cap, mvc=False, details=False, extra_flags=extra)   # 21 paths: 3 wavefronts, the last one partly filled
    ref = {}
    for c in cases:
        if c.name not in ref:
            ref[c.name] = run_pipeline(oracle_ctx, [c], max_steps=cap, mvc=False, details=False)[0]
    for k, c in enumerate(cases * 3):
        _compare(c, many[k], ref[c.name])
    ctx.close()


def test_flat_sweep_loop_capacity_and_max_time_status(hip_lib, oracle_ctx):
    import copy
    case = Case("GEN7DOF")
    ctx = capi.Context(hip_lib, 0)
    helpers.set_layout(ctx, "flat5")
    short = copy.copy(case)
    short.problem = capi.Problem.from_buffer_copy(bytes(case.problem))
    short.problem.max_integ_time = 1.0
    for c, kw in ((case, dict(max_steps=100)), (short, {})):
        a = run_pipeline(ctx, [c, c, c], mvc=False, details=False, **kw)
        b = run_pipeline(oracle_ctx, [c], mvc=False, details=False, **kw)[0]
        for q in a:
            for f in b["result"].dtype.names:
                assert q["result"][f] == b["result"][f], f
    ctx.close()


@pytest.mark.parametrize("layout", ["rows", "pairs", "cable"])
def test_spline_build_in_tiles_equals_the_sequential_kernel(hip_lib, oracle_ctx, layout):
    """K1 in tiles of knots (spline_tile.hip.h: warm-up of 64 knots per chunk, every warm-up value compared bit for bit with
    its neighbour's) against the sequential lane-per-series kernel and the oracle: coefficients of every channel identical,
    on lengths around every boundary of the decomposition (the 1024-knot threshold, multiples of the 256-knot tile +- 1) and
    on long paths.  Smooth paths: no series may need the sequential fallback.  A piecewise-linear path (kinks followed by
    exactly straight stretches: the curvature drops by more than the 20 orders of magnitude a warm-up bridges) is where the
    comparisons are EXPECTED to catch warm-ups that did not arrive -- same coefficients, through the fallback."""
    rng = np.random.default_rng(11)
    lengths = [1023, 1024, 1025, 1280, 1281, 1282, 1279, 1536, 1537, 1538, 1600, 40, 7, 3521, 6401, 20011]
    if layout == "cable":
        base = Case("synth_cspr_s3")
        lengths = [n for n in lengths if n >= 64][:9]
    else:
        base = Case("synth_gen7dof_s0")
    prob = capi.Problem.from_buffer_copy(bytes(base.problem))
    if layout == "pairs":
        prob.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
    C_in = prob.n_joints + prob.n_cart

    def smooth(n):
        if n <= base.n:
            return np.ascontiguousarray(base.y[:, :n])
        # longer than the golden path: the golden rows stretched by cubic interpolation (any smooth values do for K1)
        from scipy.interpolate import CubicSpline
        t0, t = np.linspace(0, 1, base.n), np.linspace(0, 1, n)
        return np.ascontiguousarray(np.stack([CubicSpline(t0, base.y[c])(t) for c in range(C_in)]))

    def run(ys, tiles):
        ctx = capi.Context(hip_lib, 0)
        ctx.set_spline_tiles(tiles)
        b = capi.Batch(ctx, prob, [y.shape[1] for y in ys], 64)
        for k, y in enumerate(ys):
            b.upload_knots(k, [y], [base.sres])
        b.precompute(1)
        co = [np.stack([b.coeffs(k, ch) for ch in range(C_in)]) for k in range(len(ys))]
        fb = b.spline_tile_fallbacks() if tiles else 0
        b.close(); ctx.close()
        return co, fb

    ys = [smooth(n) for n in lengths]
    tiled, fallbacks = run(ys, True)
    seq, _ = run(ys, False)
    assert fallbacks == 0, f"{fallbacks} series of smooth paths fell back to the sequential kernel"
    ob = capi.Batch(oracle_ctx, prob, [y.shape[1] for y in ys], 64)
    for k, y in enumerate(ys):
        ob.upload_knots(k, [y], [base.sres])
    ob.precompute(1)
    for k in range(len(ys)):
        assert_bit_equal(tiled[k], seq[k], f"{layout}: path {k} (N = {lengths[k]}): tiles against the sequential kernel")
        oc = np.stack([ob.coeffs(k, ch) for ch in range(C_in)])
        assert_bit_equal(tiled[k], oc, f"{layout}: path {k} (N = {lengths[k]}): tiles against the oracle")
    ob.close()
    if layout == "pairs":
        # the single-pass kernel of large pair batches (spline_stream.hip.h: forward elimination through an LDS ring, back substitution
        # in blocks that start 48 knots ahead and are checked against their neighbours), forced here on a small batch
        streamed, fb2 = run(ys, 2)
        assert fb2 == 0, f"{fb2} series of smooth paths fell back to the sequential kernel (single-pass kernel)"
        for k in range(len(ys)):
            assert_bit_equal(streamed[k], seq[k], f"pairs: path {k} (N = {lengths[k]}): single-pass kernel against the sequential kernel")
    # kinks and exactly straight stretches
    n = 5000
    t = np.linspace(0, 1, n)
    kinky = np.ascontiguousarray(np.stack([np.interp(t, np.linspace(0, 1, 12), rng.integers(-4, 5, 12).astype(float)) for _ in range(C_in)]))
    if layout == "cable":
        kinky[: prob.n_joints] = np.abs(kinky[: prob.n_joints]) + 3.0
    a, fb = run([kinky], True)
    c, _ = run([kinky], False)
    assert_bit_equal(a[0], c[0], f"{layout}: piecewise-linear path: tiles (with {fb} series through the fallback) against the sequential kernel")
    if layout == "pairs":
        a2, fb2 = run([kinky], 2)
        assert_bit_equal(a2[0], c[0], f"pairs: piecewise-linear path: single-pass kernel (with {fb2} series through the fallback) against the sequential kernel")
    # a series that MUST take the fallback: a spike no warm-up can forget (0.268 per knot: 1e30 is still 1e-7 after 64 knots, far
    # above half an ulp of the values around it) -- the chain 'comparison fails -> series marked -> sequential kernel with the mask'
    # runs, over second-derivative slots the parallel kernel has already written, and leaves the sequential result
    spiky = smooth(5000).copy()
    ch = prob.n_joints if layout == "cable" else 1
    spiky[ch, 2500] = 1e30
    a, fb = run([spiky], True)
    c, _ = run([spiky], False)
    assert fb > 0, "the tiled kernel did not notice a spike its warm-ups cannot have forgotten"
    assert_bit_equal(a[0], c[0], f"{layout}: spike: tiles + fallback against the sequential kernel")
    if layout == "pairs":
        a2, fb2 = run([spiky], 2)
        assert fb2 > 0, "the single-pass kernel did not notice a spike its warm-ups cannot have forgotten"
        assert_bit_equal(a2[0], c[0], "pairs: spike: single-pass kernel + fallback against the sequential kernel")


def _series(kind, n, seed):
    rng = np.random.default_rng(seed)
    t = np.linspace(0.0, 1.0, n)
    if kind == "smooth":
        return np.sin(7.0 * t + seed) + 0.3 * np.cos(31.0 * t) + 2.0 * t
    if kind == "rough":
        return np.cumsum(rng.normal(0.0, 1e-3, n)) + rng.normal(0.0, 1e-6, n)
    if kind == "kinks":
        return np.interp(t, np.linspace(0, 1, 9), rng.integers(-4, 5, 9).astype(float))
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["smooth", "rough", "kinks"])
def test_spline_series_on_the_lanes_of_a_wavefront(hip_ctx, oracle_ctx, kind):
    """spline_lanes.hip.h (output stage, resampler): a series in 64 chunks with bit-checked warm-ups must be the sequential Thomas
    solve exactly -- lengths at the limit of the kernel, with every remainder of the chunking, and long ones; against the
    lane-per-series kernel AND the oracle's solve"""
    for n in (4, 5, 63, 1000, 16385, 16386, 16387, 16386 + 63, 16386 + 64, 16386 + 65, 20011, 65536, 100003, 250000):
        y = _series(kind, n, n % 7)
        sol, seq, redone = capi.spline_lanes_kat(hip_ctx, y)
        ref, _, _ = capi.spline_lanes_kat(oracle_ctx, y)
        assert_bit_equal(seq, ref, f"{kind}, n = {n}: lane-per-series kernel against the oracle")
        assert_bit_equal(sol, ref, f"{kind}, n = {n}: wavefront-per-series solve (redone = {redone}) against the oracle")
        if n < 16386:
            assert redone == 1, (n, "shorter than the kernel takes: the sequential kernel must have run")
        elif kind == "smooth":
            assert redone == 0, (n, "a smooth series fell back to the sequential kernel")


def test_spline_series_whose_warm_ups_disagree_take_the_fallback(hip_ctx, oracle_ctx):
    """the chain 'a boundary comparison fails -> the series is flagged -> the lane-per-series kernel solves it again' must run and
    leave the sequential result.  A warm-up starts from the guess 0 and forgets that guess's error at 0.268 per knot (2.6e-37 after
    its 64 knots), so it arrives wrong only where the true value at its START is ~1e20 times what it is at the chunk: a spike two or
    three knots before the start of a forward warm-up (67 knots left of a chunk boundary) or behind the start of a backward one
    (66 knots right of it); boundaries of the first, a middle and the last chunk"""
    n = 40000
    lc = (n - 2) // 64
    for where in [1 + l * lc + off for l in (1, 3, 31, 63) for off in (-67, 66)]:
        for amp in (1e30, -1e25):
            y = _series("smooth", n, 3)
            y[where] = amp
            sol, seq, redone = capi.spline_lanes_kat(hip_ctx, y)
            ref, _, _ = capi.spline_lanes_kat(oracle_ctx, y)
            assert redone == 1, (where, amp, "a spike the warm-ups cannot have forgotten went unnoticed")
            assert_bit_equal(seq, ref, f"spike at {where}: lane-per-series kernel against the oracle")
            assert_bit_equal(sol, ref, f"spike at {where}: after the fallback against the oracle")
    # non-finite values: flagged as well, and the fallback leaves what the sequential kernel leaves (NaN payloads are the device's)
    y = _series("smooth", n, 4)
    y[5 * lc + 3] = np.inf
    sol, seq, redone = capi.spline_lanes_kat(hip_ctx, y)
    assert redone == 1 and sol.tobytes() == seq.tobytes()


def test_flat_sweep_loop_is_gated_by_toolchain_and_canary(hip_lib, oracle_ctx, monkeypatch):
    """the AUTOMATIC loop choice takes the flat reverse loop only if the library was built by the toolchain the loop was
    validated with and the on-device canary (nested against flat loop on ordinary, crawling and always-failing paths) found
    no difference; a library from another toolchain runs the nested loops.  The mismatch is forced through
    BATOTP_ASSUME_TOOLCHAIN, which only the TEST BUILD of the library reads (csrc/libbatotp_hip_testhooks.so, the same source
    with -DBATOTP_TEST_HOOKS); the shipped library ignores the variable.  Results are the oracle's either way."""
    shipped = hip_lib
    hip_lib = capi.Library(os.path.join(helpers.ROOT, "batotp_amd", "csrc", "libbatotp_hip_testhooks.so"))
    built, validated = hip_lib.toolchain()
    assert built and validated
    case = Case("synth_gen7dof_s0")
    flags = capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
    oo = run_pipeline(oracle_ctx, [case], mvc=False, details=False)[0]
    seen = {}
    for label, assume in (("real", None), ("other", "clang 99.0 / HIP 99.0")):
        if assume is None:
            monkeypatch.delenv("BATOTP_ASSUME_TOOLCHAIN", raising=False)
        else:
            monkeypatch.setenv("BATOTP_ASSUME_TOOLCHAIN", assume)
        ctx = capi.Context(hip_lib, 0)
        ctx.set_sweep_group(8)                                 # the layout of large batches; the loop form stays automatic
        ctx.set_paths_per_wave(8)
        prob = capi.Problem.from_buffer_copy(bytes(case.problem))
        prob.flags |= flags
        b = capi.Batch(ctx, prob, [case.n] * 9, case.max_steps())
        b.upload_knots(0, [case.y] * 9, [case.sres] * 9)
        b.optimize()
        seen[label] = (ctx.flat_loop_status(), b.last_sweep_launch(-1), b.last_sweep_launch(+1))
        res = b.results()
        for p in (0, 8):
            for f in res.dtype.names:
                assert res[p][f] == oo["result"][f], (label, p, f)
            s, sd = b.curve(p, +1)
            assert_bit_equal(s, oo["fwd"][0], f"{label} fwd.s"); assert_bit_equal(sd, oo["fwd"][1], f"{label} fwd.sdot")
        b.close(); ctx.close()
    expect_real = 1 if built == validated else -1
    assert seen["real"][0] == expect_real, seen
    # gate open: reverse = flat loop with hold 4, forward = k_sweep8 with hold 8 (the nested loops' schedule in the leaner kernel)
    assert seen["real"][1] == (8, 8, 4 if expect_real == 1 else -1) and seen["real"][2] == (8, 8, 8 if expect_real == 1 else -1), seen
    assert seen["other"][0] == -1 and seen["other"][1] == (8, 8, -1) and seen["other"][2] == (8, 8, -1), seen
    # the shipped library does not read the environment: the same variable leaves its gate where the real toolchain puts it
    monkeypatch.setenv("BATOTP_ASSUME_TOOLCHAIN", "clang 99.0 / HIP 99.0")
    ctx = capi.Context(shipped, 0)
    assert ctx.flat_loop_status() == expect_real
    ctx.close()


@pytest.mark.parametrize("lanes", [0, 1, 8, 16, 32, 64, "flat4", "64x2"])
def test_curves_in_place_give_identical_results(hip_lib, oracle_ctx, lanes):
    """BATOTP_F_CURVES_IN_PLACE: one curve buffer per path, the forward curve written over the reverse points its cursor has left
    behind -- result rows and both curves (the reverse one fetched between the sweeps) equal the oracle's"""
    ctx = capi.Context(hip_lib, 0)
    helpers.set_layout(ctx, lanes)
    for name in ("GEN7DOF", "synth_gen7dof_s0", "UR5", "CSPR3DOF", "RR_acc", "synth_cspr_s5", "synth_ur_s2", "KUKA_trq"):
        case = Case(name)
        variants = [capi.F_CURVES_IN_PLACE]
        if _vel_acc_only(name):
            variants.append(capi.F_CURVES_IN_PLACE | capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES)
        oo = run_pipeline(oracle_ctx, [case], mvc=False, details=False)[0]
        for flags in variants:
            ho = run_pipeline(ctx, [case], mvc=False, details=False, extra_flags=flags)[0]
            _compare(case, ho, oo)
    ctx.close()


def test_curves_in_place_state_and_capacity(hip_ctx, oracle_ctx):
    """after the forward sweep the reverse curve is gone (state errors until the reverse sweep has run again); a curve buffer
    with room for the forward curve + 72 points is enough, one that is too short ends with the capacity status"""
    case = Case("synth_gen7dof_s0")
    ref = run_pipeline(oracle_ctx, [case], mvc=False, details=False)[0]
    n_fwd = int(ref["result"]["n_fwd"])
    prob = capi.Problem.from_buffer_copy(bytes(case.problem))
    prob.flags |= capi.F_CURVES_IN_PLACE
    b = capi.Batch(hip_ctx, prob, [case.n, case.n], n_fwd + 72)
    b.upload_knots(0, [case.y, case.y], [case.sres, case.sres])
    b.precompute(0)
    b.sweep(-1)
    b.sweep(+1)
    for k in range(2):
        r = b.results()[k]
        for f in ref["result"].dtype.names:
            assert r[f] == ref["result"][f], f
        s, sd = b.curve(k, +1)
        helpers.assert_bit_equal(s, ref["fwd"][0], "fwd.s"); helpers.assert_bit_equal(sd, ref["fwd"][1], "fwd.sdot")
    with pytest.raises(capi.BatotpError):
        b.curve(0, -1)
    with pytest.raises(capi.BatotpError):
        b.sweep(+1)
    with pytest.raises(capi.BatotpError):
        b.pack_curves(-1, 0, 2, 0, 0)
    b.sweep(-1)                       # the reverse curve is back
    helpers.assert_bit_equal(b.curve(1, -1)[1], ref["rev"][1], "rev.sdot after re-running the reverse sweep")
    b.sweep(+1)
    assert int(b.results()[1]["n_fwd"]) == n_fwd
    b.close()
    b = capi.Batch(hip_ctx, prob, [case.n], n_fwd + 8)   # fits both curves one after the other, not the margin
    b.upload_knots(0, [case.y], [case.sres])
    b.optimize()
    r = b.results()[0]
    assert int(r["n_rev"]) == int(ref["result"]["n_rev"]) and int(r["n_fwd"]) == 0 and (int(r["status_fwd"]) & capi.ST_CAPACITY)
    b.close()


@pytest.mark.parametrize("lanes", [0, 8, 64])
def test_pointwise_values_in_the_curve_slots(hip_lib, oracle_ctx, lanes):
    """BATOTP_F_MVC_IN_CURVES (+ in-place curves, compact splines without a site array): the pointwise values, fetched before
    the sweeps, and everything the sweeps publish equal the oracle's; afterwards the values are gone (state error)"""
    ctx = capi.Context(hip_lib, 0)
    helpers.set_layout(ctx, lanes)
    for name in ("GEN7DOF", "synth_gen7dof_s0", "UR5", "CSPR3DOF", "synth_ur_s2"):
        case = Case(name)
        variants = [capi.F_MVC_IN_CURVES, capi.F_MVC_IN_CURVES | capi.F_CURVES_IN_PLACE]
        if _vel_acc_only(name):
            variants.append(capi.F_MVC_IN_CURVES | capi.F_CURVES_IN_PLACE | capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES)
        oo = run_pipeline(oracle_ctx, [case], mvc=True, details=False)[0]
        for flags in variants:
            ho = run_pipeline(ctx, [case, case], mvc=True, details=False, extra_flags=flags)
            for h in ho:
                _compare(case, h, oo)
    case = Case("GEN7DOF")
    prob = capi.Problem.from_buffer_copy(bytes(case.problem))
    prob.flags |= capi.F_MVC_IN_CURVES
    with pytest.raises(capi.BatotpError):
        capi.Batch(ctx, prob, [case.n], case.n)              # a slot shorter than 1.5 x the knots
    b = capi.Batch(ctx, prob, [case.n], case.max_steps())
    b.upload_knots(0, [case.y], [case.sres])
    b.precompute(0); b.pointwise_mvc()
    first = np.stack(b.mvc(0))
    b.sweep(-1)
    with pytest.raises(capi.BatotpError):
        b.mvc(0)
    b.sweep(+1)
    b.pointwise_mvc()                                         # again: values back, curves invalid until the reverse sweep reruns
    helpers.assert_bit_equal(np.stack(b.mvc(0)), first, "pointwise values after re-evaluation")
    with pytest.raises(capi.BatotpError):
        b.sweep(+1)
    for which in (-1, +1):                                    # the curve slots now hold pointwise values, not curves
        with pytest.raises(capi.BatotpError):
            b.curve(0, which)
    b.sweep(-1)
    assert len(b.curve(0, -1)[0]) == case.expected["n_rev"]
    with pytest.raises(capi.BatotpError):
        b.curve(0, +1)                                        # still the stale forward curve
    b.sweep(+1)
    assert int(b.results()[0]["n_fwd"]) == case.expected["n_fwd"]
    b.close(); ctx.close()


def test_capacity_and_max_time_status(hip_ctx, oracle_ctx):
    case = Case("GEN7DOF")
    for ctx in (hip_ctx, oracle_ctx):
        out = run_pipeline(ctx, [case], max_steps=100, mvc=False, details=False)[0]
        assert out["result"]["status_rev"] & capi.ST_CAPACITY
        assert out["result"]["n_rev"] == 0
    import copy
    short = copy.copy(case)
    short.problem = capi.Problem.from_buffer_copy(bytes(case.problem))
    short.problem.max_integ_time = 1.0  # 100 steps of 0.01 s
    a = run_pipeline(hip_ctx, [short], mvc=False, details=False)[0]
    b = run_pipeline(oracle_ctx, [short], mvc=False, details=False)[0]
    assert a["result"]["status_rev"] & capi.ST_MAX_INTEG_TIME
    assert a["result"]["status_rev"] == b["result"]["status_rev"]
    # every field of the result row, on both error exits (steps taken, failure counts, statuses of the skipped forward sweep)
    for c, kw in ((case, dict(max_steps=100)), (short, {})):
        a = run_pipeline(hip_ctx, [c, c, c], mvc=False, details=False, **kw)
        b = run_pipeline(oracle_ctx, [c], mvc=False, details=False, **kw)[0]
        for q in a:
            for f in b["result"].dtype.names:
                assert q["result"][f] == b["result"][f], (f, kw)


def test_properties_at_full_size(hip_ctx):
    """size-independent properties on a BASELINE-size batch (no oracle involved):
    curves ascending in s and ending on the path end, forward curve never above the reverse curve
    (evaluated on the forward sites), T = integRes * steps, identical paths give identical bits"""
    case = Case("synth_ur_s7_100k")
    outs = run_pipeline(hip_ctx, [case] * 9, mvc=False, details=False)
    s_end = case.sres * (case.n - 1)
    for o in outs:
        r = o["result"]
        assert r["status_rev"] == 0 and r["status_fwd"] == 0
        for s, sd in (o["rev"], o["fwd"]):
            assert s[0] == 0.0 and s[-1] == s_end and np.all(np.diff(s) > 0) and np.all(sd >= 0)
        assert r["t_total"] == case.problem.integ_res * r["steps_fwd"]
        rev_on_fwd = np.interp(o["fwd"][0], o["rev"][0], o["rev"][1])
        assert np.all(o["fwd"][1] <= rev_on_fwd * (1 + 1e-12) + 1e-300)
    for o in outs[1:]:
        assert_bit_equal(o["fwd"][1], outs[0]["fwd"][1], "replicated path")


def test_pointwise_kernel_slices_batches_beyond_2_to_32_lanes(hip_lib, oracle_ctx):
    """5600 paths x 1e5 knots x 8 lanes per knot is more than one launch may cover (grid x block is a 32-bit
    quantity): the lane-group-per-knot kernel is launched in slices; first, middle and last path must all be right"""
    hip_ctx = capi.Context(hip_lib, 0)
    hip_ctx.set_sweep_group(8)
    case = Case("synth_ur_s7_100k")
    prob = capi.Problem.from_buffer_copy(bytes(case.problem))
    prob.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
    B = 5600
    assert B * case.n * 8 > 2 ** 32
    b = capi.Batch(hip_ctx, prob, [case.n] * B, 1024)
    for p0 in range(0, B, 50):
        k = min(50, B - p0)
        b.upload_knots(p0, [case.y] * k, [case.sres] * k)
    b.precompute(1)
    b.pointwise_mvc()
    o = run_pipeline(oracle_ctx, [case], details=False)[0]["mvc"]
    for p in (0, 1, B // 2, 4097, B - 1):
        assert_bit_equal(np.stack(b.mvc(p)), o, f"pointwise MVC of path {p}")
    b.close()


def test_overlapped_pointwise_evaluation_gives_the_same_results(hip_lib, oracle_ctx):
    """set_overlap: the per-knot evaluation runs beside the sweeps on a second stream; every output is unchanged,
    also when a batch is stepped repeatedly (the next precompute has to wait for it)"""
    ctx = capi.Context(hip_lib, 0)
    ctx.set_overlap(True)
    for name in ("synth_ur_s2", "synth_cspr_s3"):
        case = Case(name)
        oo = run_pipeline(oracle_ctx, [case])[0]
        for _ in range(2):
            ho = run_pipeline(ctx, [case, case])
            _compare(case, ho[0], oo)
            _compare(case, ho[1], oo)
    case = Case("synth_ur_s2")
    b = capi.Batch(ctx, case.problem, [case.n], case.max_steps())
    b.upload_knots(0, [case.y], [case.sres])
    for _ in range(3):
        b.precompute(0); b.pointwise_mvc(); b.sweep(-1); b.sweep(+1)
    assert_bit_equal(np.stack(b.mvc(0)), run_pipeline(oracle_ctx, [case], details=False)[0]["mvc"], "mvc after repeated overlapped steps")
    b.close()


def test_edge_minimum_knots_and_short_sweep(hip_ctx, oracle_ctx):
    """4 knots (the minimum) and a path crossed in fewer than 4 steps (nPts<4 re-interpolation, ba.cpp:1171-1184)"""
    prob = capi.make_problem(2, 0, flags=capi.F_JNT_ACC_ON, jnt_vel_max=[1e3, 1e3], jnt_acc_max=[1e6, 1e6], integ_res=0.05)
    y = np.array([[0.0, 0.1, 0.2, 0.3], [0.0, 0.05, 0.1, 0.15]])
    outs = []
    for ctx in (hip_ctx, oracle_ctx):
        b = capi.Batch(ctx, prob, [4, 4, 4], 64)
        for k in range(3):
            b.upload_knots(k, [y * (k + 1)], [0.1 * (k + 1)])
        b.precompute(0); b.pointwise_mvc(); b.sweep(-1); b.sweep(1)
        outs.append((b.results(), [b.curve(k, -1) for k in range(3)], [b.curve(k, 1) for k in range(3)], [np.stack(b.mvc(k)) for k in range(3)]))
        b.close()
    (rh, revh, fwdh, mh), (ro, revo, fwdo, mo) = outs
    assert np.array_equal(rh, ro)
    assert np.any(rh["status_fwd"] & capi.ST_SHORT) or np.all(rh["steps_fwd"] >= 3)
    for k in range(3):
        assert_bit_equal(revh[k][1], revo[k][1], "rev sdot"); assert_bit_equal(fwdh[k][0], fwdo[k][0], "fwd s")
        assert_bit_equal(fwdh[k][1], fwdo[k][1], "fwd sdot"); assert_bit_equal(mh[k], mo[k], "mvc")


def test_uploaded_sites_and_coefficients_path(hip_ctx, oracle_ctx):
    """the marshalling entry points BA::sweep uses (sites / coefficients / curve uploaded from a Traj),
    including non-uniform knot sites (loaded instead of computed)"""
    case = Case("GEN7DOF")
    oo = run_pipeline(oracle_ctx, [case])[0]
    prob = case.problem
    n = case.n
    for stretch in (False, True):
        res = []
        for ctx in (hip_ctx, oracle_ctx):
            b = capi.Batch(ctx, prob, [n], case.max_steps())
            sites = case.sres * np.arange(n, dtype=np.float64)
            if stretch:
                sites = sites * (1.0 + 1e-3 * np.sin(np.arange(n)))   # no longer sres*k
                sites[0] = 0.0
            vf = 1.0 / case.sres
            b.upload_path_sites(0, sites, vf, vf * vf, 0)
            for ch in range(prob.n_channels):
                b.upload_coeffs(0, ch, oo["coef"][ch])
            b.sweep(-1)
            s, sd = b.curve(0, -1)
            b2 = capi.Batch(ctx, prob, [n], case.max_steps())
            b2.upload_path_sites(0, sites, vf, vf * vf, 0)
            for ch in range(prob.n_channels):
                b2.upload_coeffs(0, ch, oo["coef"][ch])
            b2.upload_curve(0, s, sd)
            b2.sweep(+1)
            res.append((s, sd, *b2.curve(0, 1), b2.results()[0]["t_total"]))
            b.close(); b2.close()
        for a, c in zip(res[0][:4], res[1][:4]):
            assert_bit_equal(a, c, f"uploaded path (stretch={stretch})")
        assert res[0][4] == res[1][4]
        if not stretch:
            assert_bit_equal(res[0][1], oo["rev"][1], "uploaded == precomputed")


def test_compact_batch_gets_its_site_array_when_sites_are_uploaded(hip_ctx, oracle_ctx):
    """compact batches keep no knot-site array (uniform sites are computed); uploading the sites of ONE path creates it for
    all: the stretched path and its untouched neighbour both equal the oracle's"""
    case = Case("GEN7DOF")
    n = case.n
    flags = capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
    sites = case.sres * np.arange(n, dtype=np.float64) * (1.0 + 1e-3 * np.sin(np.arange(n)))
    sites[0] = 0.0
    vf = 1.0 / case.sres
    out = []
    for ctx, fl in ((hip_ctx, flags), (oracle_ctx, 0)):
        prob = capi.Problem.from_buffer_copy(bytes(case.problem))
        prob.flags |= fl
        b = capi.Batch(ctx, prob, [n, n], 4 * case.max_steps())
        b.upload_knots(0, [case.y, case.y], [case.sres, case.sres])
        b.precompute(0)
        b.upload_path_sites(1, sites, vf, vf * vf, 0)
        b.pointwise_mvc(); b.sweep(-1); b.sweep(+1)
        out.append(([b.curve(k, w) for k in (0, 1) for w in (-1, 1)], [np.stack(b.mvc(k)) for k in (0, 1)], b.results().copy()))
        b.close()
    for a, c in zip(out[0][0], out[1][0]):
        assert_bit_equal(a[0], c[0], "s"); assert_bit_equal(a[1], c[1], "sdot")
    for a, c in zip(out[0][1], out[1][1]):
        assert_bit_equal(a, c, "pointwise values")
    assert out[0][2].tobytes() == out[1][2].tobytes()
    assert out[0][2][0]["n_fwd"] == case.expected["n_fwd"] and out[0][2][1]["n_fwd"] != 0


def test_product_batest_end_to_end_on_gpu(tmp_path):
    """the real drop-in: batotp_amd/host/_build/batest (host BA library + HIP library) writes the same files
    as the reference binary"""
    import filecmp, os, shutil, subprocess
    exe = os.path.join(helpers.ROOT, "batotp_amd", "host", "_build", "batest")
    assert os.path.exists(exe), "build() must produce batest"
    for name in helpers.FULL_CASES:
        work = tmp_path / name
        work.mkdir()
        src = os.path.join(helpers.GOLD, name)
        for f in os.listdir(src):
            if not f.startswith("ref_") and f not in ("knots.npz", "expected.json"):
                shutil.copy(os.path.join(src, f), work / f)
        r = subprocess.run([exe, "config.dat"], cwd=work, capture_output=True, text=True)
        assert r.returncode == 0, (name, r.stdout[-2000:])
        assert filecmp.cmp(work / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False), name
        assert filecmp.cmp(work / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False), name


@pytest.mark.parametrize("ff", [1, 0])
def test_two_paths_per_wavefront_of_the_one_path_kernel(hip_lib, oracle_ctx, ff):
    """k_sweep1 with a second path in lanes 32..63 (NP = 2: the cable robot in serial form, every channel as pairs -- what a GPU's
    whole share of BASELINE config 5 runs through): ragged batches of an odd number of paths (the last wavefront has an empty half),
    paths that differ in length and in where they bisect, with and without the certified fast-forward -- result rows and both
    curves of every path equal the oracle's run of that path alone"""
    for name in ("CSPR3DOF", "synth_cspr_s3", "synth_cspr_s5", "synth_cspr_s9_dup", "synth_cspr_s11_decim"):
        case = Case(name)
        n = case.n
        lens = sorted({n, max(8, int(0.37 * n)), max(8, int(0.6 * n)), max(8, int(0.81 * n))}) + [n]
        want = {}
        for m in sorted(set(lens)):
            class _C:
                name = f"{case.name}:{m}"
            c = _C()
            c.y, c.sres, c.problem, c.n = np.ascontiguousarray(case.y[:, :m]), case.sres, case.problem, m
            c.max_steps = case.max_steps
            want[m] = run_pipeline(oracle_ctx, [c], mvc=False, details=False)[0]
        ctx = capi.Context(hip_lib, 0)
        helpers.set_layout(ctx, "64x2")
        ctx.set_fast_forward(ff)
        prob = capi.Problem.from_buffer_copy(bytes(case.problem))
        prob.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
        b = capi.Batch(ctx, prob, lens, case.max_steps())
        b.upload_knots(0, [np.ascontiguousarray(case.y[:, :m]) for m in lens], [case.sres] * len(lens))
        b.optimize()
        assert b.last_sweep_launch(-1)[:2] == (64, 2) and b.last_sweep_launch(+1)[:2] == (64, 2), (name, b.last_sweep_launch(-1))
        res = b.results()
        for k, m in enumerate(lens):
            w = want[m]
            for f in res.dtype.names:
                assert res[k][f] == w["result"][f], (name, ff, k, m, f, res[k], w["result"])
            for which, key in ((-1, "rev"), (1, "fwd")):
                s_, sd_ = b.curve(k, which)
                assert_bit_equal(s_, w[key][0], f"{name} ff {ff} path {k} {key}.s")
                assert_bit_equal(sd_, w[key][1], f"{name} ff {ff} path {k} {key}.sdot")
        b.close(); ctx.close()


@pytest.mark.parametrize("layout", [0, "flat4", 32, 64, "64x2"])
def test_ragged_batches_are_swept_longest_path_first(hip_lib, oracle_ctx, layout):
    """a batch whose paths differ in length is swept in the order of decreasing knot count
    (batotp_hip_set_path_order 1, the default: launch slot k runs path order[k]) -- longest-processing-time-first for the kernels
    with a wavefront per path, paths of similar length in one wavefront for the others.  Result rows and curves of every path are
    those of the static order (mode 0) and the oracle's, whatever slot ran it"""
    case = Case("synth_gen7dof_s0")
    rng = np.random.default_rng(99)
    lens = [int(v) for v in rng.integers(40, case.n, 11)] + [case.n, 17, case.n]
    ys = [np.ascontiguousarray(case.y[:, :n]) for n in lens]
    want = {}
    for n in sorted(set(lens)):
        class _C:
            name = f"prefix{n}"
        c = _C()
        c.y, c.sres, c.problem, c.n = np.ascontiguousarray(case.y[:, :n]), case.sres, case.problem, n
        c.max_steps = case.max_steps
        want[n] = run_pipeline(oracle_ctx, [c], mvc=False, details=False)[0]
    got = {}
    for mode in (1, 0):
        ctx = capi.Context(hip_lib, 0)
        helpers.set_layout(ctx, layout)
        ctx.set_path_order(mode)
        prob = capi.Problem.from_buffer_copy(bytes(case.problem))
        prob.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
        b = capi.Batch(ctx, prob, lens, case.max_steps())
        b.upload_knots(0, ys, [case.sres] * len(ys))
        b.optimize()
        res = b.results()
        got[mode] = (res.tobytes(), [(b.curve(k, -1), b.curve(k, +1)) for k in range(len(lens))])
        for k, n in enumerate(lens):
            w = want[n]
            for f in res.dtype.names:
                assert res[k][f] == w["result"][f], (layout, mode, k, n, f)
            for which, key in ((0, "rev"), (1, "fwd")):
                assert_bit_equal(got[mode][1][k][which][0], w[key][0], f"layout {layout} mode {mode} path {k} {key}.s")
                assert_bit_equal(got[mode][1][k][which][1], w[key][1], f"layout {layout} mode {mode} path {k} {key}.sdot")
        b.close(); ctx.close()
    assert got[0][0] == got[1][0]


def test_paths_of_one_batch_integrate_with_their_own_steps(hip_lib, oracle_ctx):
    """batotp_hip_set_path_integ_res: the automatic integration resolution of the reference (ba.cpp:493-556, class default
    ba.h:309) gives every path its own _integRes.  A batch whose paths integrate with different steps equals, row for row and
    point for point, the oracle run of each path alone with that step -- in every sweep kernel (one path per wavefront, the
    8-lane layout with nested loops, k_sweep8) and through the pointwise evaluation"""
    steps = [0.01, 0.004, 0.0075, 0.02, 0.01, 0.004]
    for name, extra in (("synth_gen7dof_s0", capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES), ("synth_gen7dof_s0", 0), ("synth_cspr_s3", 0), ("UR5", 0)):
        case = Case(name)
        want = []
        for h in steps:
            cs = Case(name)
            cs.problem.integ_res = h
            cs.max_steps = lambda: int(case.max_steps() * 0.01 / min(steps)) + 64
            want.append(run_pipeline(oracle_ctx, [cs], mvc=True, details=False)[0])
        for layout in (0, 8, "flat4", "oldflat4", 64, 32, "64x2"):
            ctx = capi.Context(hip_lib, 0)
            helpers.set_layout(ctx, layout)
            prob = capi.Problem.from_buffer_copy(bytes(case.problem))
            prob.flags |= extra
            cap = int(case.max_steps() * 0.01 / min(steps)) + 64
            b = capi.Batch(ctx, prob, [case.n] * len(steps), cap)
            b.upload_knots(0, [case.y] * len(steps), [case.sres] * len(steps))
            b.set_path_integ_res(0, steps)
            helpers.precompute_with_trig(ctx, b, prob, len(steps), None)
            b.pointwise_mvc()
            mvc = [np.stack(b.mvc(k)) for k in range(len(steps))]
            b.sweep(-1); b.sweep(+1)
            res = b.results()
            for k, w in enumerate(want):
                for f in res.dtype.names:
                    assert res[k][f] == w["result"][f], (name, layout, k, f, res[k], w["result"])
                assert_bit_equal(mvc[k], w["mvc"], f"{name} layout {layout} path {k}: pointwise values")
                for which, key in ((-1, "rev"), (1, "fwd")):
                    s, sd = b.curve(k, which)
                    assert_bit_equal(s, w[key][0], f"{name} layout {layout} path {k} {key}.s")
                    assert_bit_equal(sd, w[key][1], f"{name} layout {layout} path {k} {key}.sdot")
            b.close(); ctx.close()
