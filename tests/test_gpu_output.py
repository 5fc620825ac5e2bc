"""GPU (-m gpu): the device output stage against the reference binary's traj_out.dat and against the
oracle, through the C-ABI, bit for bit (fp64, tolerance 0)."""
import numpy as np
import pytest

import helpers
from helpers import OUTPUT_CASES, Case, run_to_output, assert_output_equals_reference_file, assert_bit_equal, output_params
from batotp_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [0, capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES])
@pytest.mark.parametrize("name", OUTPUT_CASES)
def test_hip_output_matches_reference_traj_out(hip_ctx, oracle_ctx, name, flags):
    case = Case(name)
    f = case.problem.flags
    cable_serial = (f & capi.F_TRQ_ON) and (f & capi.F_PARALLEL) and (f & capi.F_PAR2SER)
    if flags and (f & (capi.F_CART_VEL_ON | capi.F_CART_ACC_ON | capi.F_TRQ_ON)) and not cable_serial:
        pytest.skip("pairs: velocity/acceleration-only problems, and the cable robot in serial form (the other robots' host trig tables and the "
                    "two-link arm's output stage read the sample array)")
    ho, hb = run_to_output(hip_ctx, [case], extra_flags=flags)
    oo, ob = run_to_output(oracle_ctx, [case])
    assert_output_equals_reference_file(case, ho)
    assert int(ho.n_pts[0]) == int(oo.n_pts[0]) and ho.sres[0] == oo.sres[0]
    assert (ho.n_theta, ho.n_cart, ho.n_trq) == (oo.n_theta, oo.n_cart, oo.n_trq)
    assert_bit_equal(ho.rows(0), oo.rows(0), f"{name}: output trajectory (fp64; joints, Cartesian rows, torques)")
    for x in (ho, oo):
        x.close()
    hb.close(); ob.close()


def _variants(base):
    """parameter sets reaching every branch: plain, smoothing only, re-interpolation only, both, even / large windows"""
    out = []
    for out_res, smooth in ((base.integ_res, 1.0), (base.integ_res, 5.0), (base.integ_res * 0.8, 1.0), (base.integ_res * 0.8, 5.0),
                            (base.integ_res * 2.5, 4.0), (base.integ_res * 0.5, 9.0), (base.integ_res, 1.6)):
        out.append(capi.OutputParams(base.n_joints, base.path_type, base.integ_res, out_res, smooth))
    return out


@pytest.mark.parametrize("name", ["synth_cspr_s3", "CSPR3DOF_par"])
def test_cable_robot_every_branch_matches_the_oracle(hip_ctx, oracle_ctx, name):
    """CART path of the 3-cable robot: Cartesian rows, cable lengths and recomputed cable tensions, every branch"""
    case = Case(name)
    base = output_params(name)
    outs = []
    for ctx in (hip_ctx, oracle_ctx):
        b = capi.Batch(ctx, case.problem, [case.n, case.n], case.max_steps())
        b.upload_knots(0, [case.y, case.y], [case.sres, case.sres])
        b.optimize()
        outs.append(b)
    hb, ob = outs
    for prm in _variants(base):
        h, o = capi.Output(hb, prm, 0, 2), capi.Output(ob, prm, 0, 2)
        what = f"{name} out_res={prm.out_res} smooth={prm.out_smooth_fact}"
        assert (h.n_theta, h.n_cart, h.n_trq) == (3, 3, 3)
        assert np.array_equal(h.n_pts, o.n_pts), what
        for k in range(2):
            assert_bit_equal(h.rows(k), o.rows(k), f"{what}: path {k}")
        h.close(); o.close()
    hb.close(); ob.close()


@pytest.mark.parametrize("name", ["KUKA-LWR-IV", "RR", "RR_acc", "KUKA_trq"])
def test_forward_kinematics_robots_every_branch_matches_the_oracle(hip_ctx, oracle_ctx, name):
    """JOINT paths of the robots with forward kinematics: joint rows, Cartesian rows by Robot::fwdKin at
    the output points and, with torque constraints, the serial-robot torque recomputation (reference ba.cpp:1791-1827: clamped
    splines, Robot::dynRR or the chain model of BASELINE config 3) -- every branch of the stage, two paths of different length"""
    case = Case(name)
    short = Case(name)
    short.y = np.ascontiguousarray(short.y[:, :max(40, case.n // 3)])
    base = output_params(name) if name in OUTPUT_CASES else capi.OutputParams(case.problem.n_joints, capi.PATH_JOINT, case.problem.integ_res, 0.008, 5.0)
    outs = []
    for ctx in (hip_ctx, oracle_ctx):
        b = capi.Batch(ctx, case.problem, [case.n, short.n], case.max_steps())
        b.upload_knots(0, [case.y], [case.sres]); b.upload_knots(1, [short.y], [short.sres])
        helpers.precompute_with_trig(ctx, b, case.problem, 2)
        b.sweep(-1); b.sweep(+1)
        outs.append(b)
    hb, ob = outs
    assert hb.results().tobytes() == ob.results().tobytes()
    nJ = case.problem.n_joints
    trq = bool(case.problem.flags & capi.F_TRQ_ON)
    for prm in _variants(base):
        h, o = capi.Output(hb, prm, 0, 2), capi.Output(ob, prm, 0, 2)
        what = f"{name} out_res={prm.out_res} smooth={prm.out_smooth_fact}"
        assert (h.n_theta, h.n_cart, h.n_trq) == (nJ, 3, nJ if trq else 0) == (o.n_theta, o.n_cart, o.n_trq), what
        assert np.array_equal(h.n_pts, o.n_pts) and h.sres.tobytes() == o.sres.tobytes(), what
        for k in range(2):
            assert_bit_equal(h.rows(k), o.rows(k), f"{what}: path {k}")
        h.close(); o.close()
    hb.close(); ob.close()


def test_forward_kinematics_with_the_device_libm_is_close(hip_ctx, oracle_ctx):
    """without BATOTP_F_HOST_TRIG the tool point uses the device libm: same point counts, Cartesian rows within 1e-12 of the
    host-trig result (the documented tolerance mode; joint rows do not depend on it)"""
    case = Case("KUKA-LWR-IV")
    prob = capi.Problem.from_buffer_copy(bytes(case.problem))
    prob.flags &= ~capi.F_HOST_TRIG
    rows = []
    for p in (case.problem, prob):
        b = capi.Batch(hip_ctx, p, [case.n], case.max_steps())
        b.upload_knots(0, [case.y], [case.sres])
        b.optimize()
        o = capi.Output(b, output_params(case.name), 0, 1)
        rows.append(o.rows(0))
        o.close(); b.close()
    nJ = case.problem.n_joints
    assert_bit_equal(rows[0][:nJ], rows[1][:nJ], "joint rows")
    assert rows[0].shape == rows[1].shape and np.max(np.abs(rows[0][nJ:] - rows[1][nJ:])) < 1e-12


def test_every_branch_and_a_ragged_batch_match_the_oracle(hip_ctx, oracle_ctx):
    cases = [Case("synth_gen7dof_s0"), Case("synth_gen7dof_s0")]
    short = Case("synth_gen7dof_s0")
    short.y = np.ascontiguousarray(short.y[:, :400])   # a shorter path in the same batch
    cases.insert(1, short)
    base = output_params("synth_gen7dof_s0")
    outs = []
    for ctx in (hip_ctx, oracle_ctx):
        b = capi.Batch(ctx, cases[0].problem, [c.n for c in cases], max(c.max_steps() for c in cases))
        for k, c in enumerate(cases):
            b.upload_knots(k, [c.y], [c.sres])
        b.optimize()
        outs.append(b)
    hb, ob = outs
    for prm in _variants(base):
        h, o = capi.Output(hb, prm, 0, 3), capi.Output(ob, prm, 0, 3)
        what = f"out_res={prm.out_res} smooth={prm.out_smooth_fact}"
        assert np.array_equal(h.n_pts, o.n_pts), what
        assert h.sres.tobytes() == o.sres.tobytes(), what
        for k in range(3):
            assert_bit_equal(h.theta(k), o.theta(k), f"{what}: path {k}")
        # a sub-range of the batch gives the same trajectories
        h1 = capi.Output(hb, prm, 1, 2)
        assert_bit_equal(h1.theta(1), o.theta(2), f"{what}: sub-range")
        for x in (h, o, h1):
            x.close()
    hb.close(); ob.close()


def test_chunked_output_equals_one_chunk(hip_lib, oracle_ctx):
    """a tiny scratch budget forces one chunk per path: the result must not depend on the chunking"""
    case = Case("synth_ur_s2")
    ctx = capi.Context(hip_lib, 0)
    ctx.set_workspace_budget(output_bytes=1 << 20)
    ho, hb = run_to_output(ctx, [case, case, case])
    oo, ob = run_to_output(oracle_ctx, [case])
    for k in range(3):
        assert_bit_equal(ho.theta(k), oo.theta(0), f"chunked output, path {k}")
    assert_output_equals_reference_file(case, ho, 2)
    ho.close(); oo.close(); hb.close(); ob.close()


def test_baseline_size_path_and_failed_paths(hip_ctx, oracle_ctx):
    """a 1e5-knot path (hundreds of thousands of output points) and a path whose sweep hit the curve capacity"""
    case = Case("synth_ur_s7_100k")
    prm = capi.OutputParams(case.problem.n_joints, capi.PATH_JOINT, case.problem.integ_res, 0.008, 5.0)
    outs = []
    for ctx in (hip_ctx, oracle_ctx):
        b = capi.Batch(ctx, case.problem, [case.n], case.max_steps())
        b.upload_knots(0, [case.y], [case.sres])
        b.optimize()
        outs.append((capi.Output(b, prm, 0, 1), b))
    (h, hb), (o, ob) = outs
    assert int(h.n_pts[0]) == int(o.n_pts[0]) > 10000
    assert_bit_equal(h.theta(0), o.theta(0), "1e5-knot path")
    for x in (h, o, hb, ob):
        x.close()
    small = Case("synth_gen7dof_s0")
    b = capi.Batch(hip_ctx, small.problem, [small.n], 64)   # capacity far too small: the sweep reports it
    b.upload_knots(0, [small.y], [small.sres])
    b.optimize()
    out = capi.Output(b, output_params("synth_gen7dof_s0"), 0, 1)
    assert int(out.n_pts[0]) == 0 and out.theta(0).size == 0
    out.close(); b.close()


def test_long_series_on_the_lanes_of_a_wavefront(hip_ctx, oracle_ctx):
    """series of more than 16 386 values take spline_lanes.hip.h (64 chunks per series, warm-ups compared bit for bit), shorter
    ones the lane-per-series kernel: a 1e5-knot path, a prefix of it just above and one just below the limit and a short path in
    one batch, through the branches that build series (s(t) always; every channel with re-interpolation)"""
    case = Case("synth_ur_s7_100k")
    ys = [case.y, np.ascontiguousarray(case.y[:, :68000]), np.ascontiguousarray(case.y[:, :66000]), np.ascontiguousarray(case.y[:, :600]), case.y]
    outs = []
    for ctx in (hip_ctx, oracle_ctx):
        b = capi.Batch(ctx, case.problem, [y.shape[1] for y in ys], case.max_steps())
        for k, y in enumerate(ys):
            b.upload_knots(k, [y], [case.sres])
        b.optimize()
        outs.append(b)
    hb, ob = outs
    res = hb.results()
    fwd = [int(r["n_fwd"]) for r in res]
    assert fwd[0] > fwd[1] > 16386 > fwd[2] > 15000 > fwd[3], fwd
    base = capi.OutputParams(case.problem.n_joints, capi.PATH_JOINT, case.problem.integ_res, 0.008, 5.0)
    for prm in [base] + _variants(base)[2:6]:
        h, o = capi.Output(hb, prm, 0, len(ys)), capi.Output(ob, prm, 0, len(ys))
        what = f"out_res={prm.out_res} smooth={prm.out_smooth_fact}"
        assert np.array_equal(h.n_pts, o.n_pts), what
        for k in range(len(ys)):
            assert_bit_equal(h.theta(k), o.theta(k), f"{what}: path {k}")
        h.close(); o.close()
    hb.close(); ob.close()


@pytest.mark.parametrize("n_paths", [1, 2, 63, 64, 65, 200])
def test_segment_cursor_never_moves_back_on_every_path_of_a_chunk(hip_ctx, oracle_ctx, n_paths):
    """findInterpSegs' cursor (reference spline.cpp:56-99) as the stage runs it -- one wavefront per path, through the stage's own launch
    function: raw segment indices that step BACK (an s(t) spline that overshoots) on every path of a chunk, lengths around the 256 sites a
    wavefront takes per round, empty paths in between.  (Round 5 launched this kernel on a lane-per-path grid: only paths 0 ..
    ceil(K / 64) - 1 were processed and no test saw it, every test curve being monotone.)"""
    rng = np.random.default_rng(77 + n_paths)
    segs = []
    for k in range(n_paths):
        n = int(rng.choice([0, 1, 3, 255, 256, 257, 511, 1024, 1500, 4099]))
        raw = np.cumsum(rng.integers(0, 3, n)) - rng.integers(0, 4, n) * (rng.random(n) < 0.3)   # mostly rising, with dips
        raw = np.maximum(raw, 0).astype(np.int32)
        if n >= 3:
            raw[n // 2] = raw[n // 2 - 1] + 5      # a certain step back on every path: the site after a jump
            raw[n // 2 + 1] = raw[n // 2 - 1]
        segs.append(raw)
    if sum(len(x) for x in segs) == 0:
        segs[0] = np.array([3, 1, 2], dtype=np.int32)
    want = [np.maximum.accumulate(x) if len(x) else x for x in segs]
    assert all(len(x) < 3 or not np.array_equal(w, x) for w, x in zip(want, segs))
    for ctx in (hip_ctx, oracle_ctx):
        got = capi.out_segmax_kat(ctx, segs)
        for k in range(n_paths):
            assert np.array_equal(got[k], want[k]), (n_paths, k, len(segs[k]))
