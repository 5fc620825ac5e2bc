"""GPU (-m gpu): the device resampler against the golden knots and the oracle, through
the C-ABI, bit for bit (fp64, tolerance 0)."""
import numpy as np
import pytest

import helpers
from helpers import RESAMPLE_CASES, ResampleCase, Case, assert_bit_equal
from batotp_amd import capi, pathgen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", RESAMPLE_CASES)
def test_hip_resample_matches_golden_knots(hip_ctx, name):
    c = ResampleCase(name)
    r = capi.Resampled(hip_ctx, c.params, [c.x], [c.sres_in])
    assert int(r.status[0]) == 0
    assert int(r.n_knots[0]) == c.y.shape[1]
    assert r.sres[0] == c.sres
    assert_bit_equal(r.knots(0), c.y, f"{name}: resampled knots")
    r.close()


def _same(h, o, what):
    assert np.array_equal(h.status, o.status), what
    assert np.array_equal(h.n_knots, o.n_knots), what
    assert h.sres.tobytes() == o.sres.tobytes(), what
    for k in range(h.n_paths):
        assert_bit_equal(h.knots(k), o.knots(k), f"{what}: path {k}")


def test_ragged_batch_with_failing_and_duplicated_paths(hip_ctx, oracle_ctx):
    c = ResampleCase("synth_gen7dof_s0")
    xs = [c.x, np.repeat(c.x[:, :1], 16, axis=1), np.repeat(c.x, 2, axis=1), c.x[:, :37].copy(), c.x[:, ::-1].copy()]
    # a path whose last two points coincide (the tail rule of remClosePts)
    t = c.x[:, :200].copy()
    t[:, -1] = t[:, -2]
    xs.append(t)
    sr = [c.sres_in] * len(xs)
    h = capi.Resampled(hip_ctx, c.params, xs, sr)
    o = capi.Resampled(oracle_ctx, c.params, xs, sr)
    assert int(h.status[1]) != 0  # all points identical: the reference performs no optimisation
    _same(h, o, "ragged gen7dof batch")
    assert_bit_equal(h.knots(2), c.y, "duplicated taught points are dropped first")


@pytest.mark.parametrize("name", ["synth_gen7dof_s0", "synth_cspr_s3"])
def test_running_arc_lengths_with_a_row_per_chain(hip_ctx, oracle_ctx, name):
    """k_rs_scan_rows adds the step lengths of two paths per wavefront, 16 points of a chain per block: taught paths whose
    lengths sit around the block edges, a failing path as either partner of a wavefront, an odd number of paths"""
    c = ResampleCase(name)
    n = c.x.shape[1]
    dead = np.repeat(c.x[:, :1], 16, axis=1)
    lens = [4, 5, 16, 17, 18, 31, 32, 33, 34, 47, 48, 49, 50, 64, 65, 66, 129, n // 3, n]
    assert n > max(lens[:-2])
    xs = [dead] + [c.x[:, :k].copy() for k in lens] + [dead, c.x[:, n // 2:].copy(), c.x[:, ::-1].copy()]
    xs.insert(3, dead)                       # ... and as the second path of a wavefront
    if len(xs) % 2 == 0:
        xs.append(c.x[:, 7:107].copy())
    whole = [k for k, x in enumerate(xs) if x.shape[1] == n and np.array_equal(x, c.x)][0]
    sr = [c.sres_in] * len(xs)
    h = capi.Resampled(hip_ctx, c.params, xs, sr)
    o = capi.Resampled(oracle_ctx, c.params, xs, sr)
    assert int(h.status[0]) != 0 and int(h.status[3]) != 0 and int(h.status[-1]) == 0 and len(xs) % 2 == 1
    assert np.array_equal(h.status, o.status) and np.array_equal(h.n_knots, o.n_knots) and h.sres.tobytes() == o.sres.tobytes()
    for k in range(h.n_paths):
        if not int(h.status[k]):            # (what the rows of a refused path hold is nobody's business)
            assert_bit_equal(h.knots(k), o.knots(k), f"{name}: block edges of the row scan: path {k}")
    assert_bit_equal(h.knots(whole), c.y, "the whole golden path inside the batch")


@pytest.mark.parametrize("name", ["KUKA-LWR-IV", "RR"])
def test_forward_kinematics_robots_ragged_batch(hip_ctx, oracle_ctx, name):
    """JOINT paths of the robots with forward kinematics: the tool point is recomputed before the first pass
    (Cartesian constraint on) and after each pass, with the cos / sin of the host libm; ragged batch incl. a degenerate path"""
    c = ResampleCase(name)
    n = c.x.shape[1]
    rng = np.random.default_rng(5)
    wob = c.x.copy()
    wob[:c.params.n_joints] += rng.normal(0, 0.05, (c.params.n_joints, n))       # another path through the same region
    xs = [c.x, c.x[:, :n // 2].copy(), c.x[:, ::-1].copy(), np.repeat(c.x[:, :1], 16, axis=1), wob, np.repeat(c.x, 2, axis=1)]
    sr = [c.sres_in] * len(xs)
    h = capi.Resampled(hip_ctx, c.params, xs, sr)
    o = capi.Resampled(oracle_ctx, c.params, xs, sr)
    assert int(h.status[3]) != 0 and int(h.status[0]) == 0
    _same(h, o, f"ragged {name} batch")
    assert_bit_equal(h.knots(0), c.y, "golden knots inside the batch")
    assert_bit_equal(h.knots(5), c.y, "duplicated taught points are dropped first")
    # Cartesian constraints off: the taught Cartesian rows pass through the first pass untouched (ba.cpp:258-262)
    prm = capi.ResampleParams.from_buffer_copy(bytes(c.params))
    prm.flags &= ~(capi.F_CART_VEL_ON | capi.F_CART_ACC_ON)
    h2 = capi.Resampled(hip_ctx, prm, xs[:3], sr[:3])
    o2 = capi.Resampled(oracle_ctx, prm, xs[:3], sr[:3])
    _same(h2, o2, f"{name} without Cartesian constraints")


def test_pose_paths_ragged_batch(hip_ctx, oracle_ctx):
    """path type BOTH (the UR5 example): six pose rows in, seven out -- BA::aa2qVect on the device (quaternion of every
    taught point with the host's sincos of the half angle, sequential hemisphere alignment per path), then both passes on
    joints AND poses; ragged batch incl. orientations that flip hemisphere and a (near-)zero rotation"""
    c = ResampleCase("UR5")
    nJ = c.params.n_joints
    assert c.params.path_type == capi.PATH_BOTH and c.params.n_cart == 6 and c.y.shape[0] == nJ + 7
    n = c.x.shape[1]
    flip = c.x.copy()
    flip[nJ + 3:nJ + 6, n // 2:] *= -1.0           # the same rotations by the opposite axis-angle vectors: q and -q ...
    zero = c.x.copy()
    zero[nJ + 3:nJ + 6, : n // 3] = 0.0            # no rotation at all on the first third (the theta < 1e-6 branch)
    xs = [c.x, c.x[:, : n // 2].copy(), flip, zero, c.x[:, ::-1].copy()]
    sr = [c.sres_in] * len(xs)
    h = capi.Resampled(hip_ctx, c.params, xs, sr)
    o = capi.Resampled(oracle_ctx, c.params, xs, sr)
    assert not h.status.any()
    _same(h, o, "ragged pose batch")
    assert_bit_equal(h.knots(0), c.y, "golden knots inside the batch")


def test_forward_kinematics_with_the_device_libm_is_close(hip_ctx):
    """without BATOTP_F_HOST_TRIG the tool point comes from the device libm (last-bit differences against glibc): the
    documented tolerance mode -- same knot count on the golden path, knots within 1e-9"""
    c = ResampleCase("KUKA-LWR-IV")
    prm = capi.ResampleParams.from_buffer_copy(bytes(c.params))
    assert prm.flags & capi.F_HOST_TRIG
    prm.flags &= ~capi.F_HOST_TRIG
    r = capi.Resampled(hip_ctx, prm, [c.x], [c.sres_in])
    assert int(r.status[0]) == 0 and int(r.n_knots[0]) == c.y.shape[1]
    assert np.max(np.abs(r.knots(0) - c.y)) < 1e-9
    r.close()


@pytest.mark.parametrize("kind", ["gen7", "ur", "cspr"])
def test_baseline_size_paths_match_oracle(hip_ctx, oracle_ctx, kind):
    """BASELINE.json-sized taught paths (tens of thousands of points, ~1e5 knots), several seeds per batch"""
    if kind == "gen7":
        base = ResampleCase("synth_gen7dof_s0")
        xs = [pathgen.gen7dof_fine(s, 400 + 50 * s) for s in (11, 12, 13)]
    elif kind == "ur":
        base = ResampleCase("synth_ur_s2")
        xs = [pathgen.ur_like_fine(s, 300 + 40 * s) for s in (21, 22)]
    else:
        base = ResampleCase("synth_cspr_s3")
        xs = [pathgen.cspr_fine(s, 100 + 20 * s) for s in (31, 32, 33)]
    nJ, nC = base.params.n_joints, base.params.n_cart
    full = []
    for x in xs:
        # the taught file stores float32: widen the way the reader does
        x = x.astype(np.float32).astype(np.float64)
        f = np.zeros((nJ + nC, x.shape[1]))
        if kind == "cspr":
            f[nJ:] = x
        else:
            f[:nJ] = x
        full.append(f)
    sr = [base.sres_in] * len(full)
    h = capi.Resampled(hip_ctx, base.params, full, sr)
    o = capi.Resampled(oracle_ctx, base.params, full, sr)
    assert not h.status.any()
    _same(h, o, f"{kind} baseline-size batch")


def test_resampled_knots_feed_the_hot_path_on_the_device(hip_ctx):
    """resample -> upload_knots_device -> precompute -> sweeps, nothing leaves HBM in between; the
    traversal time and step counts are the golden case's"""
    name = "synth_cspr_s3"
    c, rc = Case(name), ResampleCase(name)
    r = capi.Resampled(hip_ctx, rc.params, [rc.x, rc.x], [rc.sres_in] * 2)
    b = capi.Batch(hip_ctx, c.problem, [int(n) for n in r.n_knots], c.max_steps())
    b.upload_knots_device(0, 2, r.device_ptr(), list(r.sres))
    b.optimize()
    res = b.results()
    for k in range(2):
        assert int(res[k]["n_rev"]) == c.expected["n_rev"] and int(res[k]["n_fwd"]) == c.expected["n_fwd"]
        assert f"{float(res[k]['t_total']):.2f}" == f"{c.expected['t_total_print']:.2f}"
        s, sd = b.curve(k, +1)
        assert helpers.f32_digest(s, sd) == c.expected["sha256_fwd"]
    b.close()
    r.close()


def test_knot_checksums_computed_in_hbm_equal_the_checkers(hip_ctx, oracle_ctx):
    """batotp_hip_resampled_checksums: the per-path sums the host library compares between its two evaluations of the resampler,
    computed on the device by k_rs_checksum (grid-stride, one atomic add per wavefront) against the checker's serial sums -- ragged batch
    incl. a path that ends with a status (sum 0) and BASELINE-size paths"""
    c = ResampleCase("synth_gen7dof_s0")
    same = c.x.copy(); same[:, :] = same[:, :1]
    big = [_big_gen7_taught_points(s) for s in (1, 2)]
    xs = [c.x, c.x[:, : c.x.shape[1] // 2].copy(), same, c.x[:, ::-1].copy()] + big
    sr = [c.sres_in] * len(xs)
    h = capi.Resampled(hip_ctx, c.params, xs, sr)
    o = capi.Resampled(oracle_ctx, c.params, xs, sr)
    _same(h, o, "checksum batch")
    hs, os_ = h.checksums(), o.checksums()
    assert np.array_equal(hs, os_), (hs, os_)
    assert int(hs[2]) == 0 and int(h.status[2]) != 0 and len(set(int(v) for v in hs)) == len(xs)
    h.close(); o.close()


def test_one_path_calls_can_keep_a_checksum_of_every_intermediate_stage(hip_lib):
    """batotp_hip_set_resample_trace: the diagnostic behind the two-evaluations guard of BA::interpInputData -- eight stage checksums of a
    one-path call, the same from call to call, the last one the checksum of the knots; a call with more than one path keeps none"""
    ctx = capi.Context(hip_lib, 0)
    ctx.set_resample_trace(True)
    traces = []
    for name in ("synth_gen7dof_s0", "synth_cspr_s3", "UR5_pos3"):
        c = ResampleCase(name)
        per_case = []
        for _ in range(3):
            r = capi.Resampled(ctx, c.params, [c.x], [c.sres_in])
            t = r.trace()
            assert int(t[7]) == int(r.checksums()[0]) and all(int(v) != 0 for v in t), (name, t)
            # the arrays of stages 2 and 3 are kept on the host as well: second derivatives of the taught points, the emitted points
            n_taught_kept = r.trace_data(2).size // c.y.shape[0]
            assert r.trace_data(2).size % c.y.shape[0] == 0 and 4 <= n_taught_kept <= c.x.shape[1] and r.trace_data(3).size % c.y.shape[0] == 0
            assert r.trace_data(0).size == 0 and not np.isnan(r.trace_data(2)).any() and not np.isnan(r.trace_data(3)).any()
            assert_bit_equal(r.knots(0), c.y, name)
            per_case.append(t.tobytes())
            r.close()
        assert len(set(per_case)) == 1, name
        traces.append(per_case[0])
    assert len(set(traces)) == 3
    c = ResampleCase("synth_gen7dof_s0")
    r = capi.Resampled(ctx, c.params, [c.x, c.x], [c.sres_in] * 2)
    with pytest.raises(capi.BatotpError):
        r.trace()
    r.close()
    ctx.close()


def _big_gen7_taught_points(seed):
    """taught points of a GEN7DOF path of ~5e4 knots (bench.py's generator), widened to the resampler's rows"""
    import bench
    taught, _ = bench.taught_points_f32("gen7", [seed], 50000)
    return bench.widen("gen7", taught[0])


@pytest.mark.parametrize("compact", [False, True])
def test_block_upload_of_paths_that_carry_more_rows_than_the_batch_keeps(hip_ctx, oracle_ctx, compact):
    """batotp_hip_upload_knots_device_rows: the resampler leaves joint AND Cartesian rows per path, a problem without Cartesian limits
    keeps the joint rows only -- one call for a ragged block of paths, rows and pairs layout: coefficients / curves / result rows equal
    to a batch filled path by path from host arrays, and to the oracle's"""
    name = "synth_gen7dof_s0"
    c, rc = Case(name), ResampleCase(name)
    nJ = c.problem.n_joints
    xs = [rc.x, rc.x[:, : rc.x.shape[1] // 2].copy(), rc.x[:, ::-1].copy()]
    r = capi.Resampled(hip_ctx, rc.params, xs, [rc.sres_in] * 3)
    src_rows = r.knots(0).shape[0]
    assert src_rows > nJ and not r.status.any()
    prob = capi.Problem.from_buffer_copy(bytes(c.problem))
    prob.n_cart = 0                                   # (no Cartesian limit: no Cartesian channels are carried)
    prob.flags |= capi.F_NO_SAMPLES | (capi.F_COMPACT_SPLINES if compact else 0)
    nk = [int(n) for n in r.n_knots]
    rows = []
    for mode in ("block", "per_path", "oracle"):
        ctx = oracle_ctx if mode == "oracle" else hip_ctx
        pr = capi.Problem.from_buffer_copy(bytes(prob))
        if mode == "oracle":
            pr.flags &= ~capi.F_COMPACT_SPLINES
        b = capi.Batch(ctx, pr, nk, 4 * c.max_steps())
        if mode == "block":
            b.upload_knots_device_rows(0, 3, r.device_ptr(), src_rows, list(r.sres))
        else:
            for k in range(3):
                b.upload_knots(k, [np.ascontiguousarray(r.knots(k)[:nJ])], [float(r.sres[k])])
        b.optimize()
        rows.append((b.results(), [b.curve(k, +1) for k in range(3)]))
        b.close()
    for other in rows[1:]:
        assert rows[0][0].tobytes() == other[0].tobytes()
        for k in range(3):
            assert_bit_equal(rows[0][1][k][0], other[1][k][0], f"path {k} s")
            assert_bit_equal(rows[0][1][k][1], other[1][k][1], f"path {k} sdot")
    # too few source rows, host pointers: refused
    b = capi.Batch(hip_ctx, prob, nk, 64)
    with pytest.raises(capi.BatotpError):
        b.upload_knots_device_rows(0, 3, r.device_ptr(), nJ - 1, list(r.sres))
    b.close(); r.close()


@pytest.mark.parametrize("name", ["synth_cspr_s3", "synth_gen7dof_s0", "synth_ur_s2", "GEN7DOF", "CSPR3DOF", "UR5", "UR5_pos3", "KUKA-LWR-IV", "KUKA_cartacc", "RR", "RR_acc"])
def test_product_batch_driver_on_gpu(tmp_path, name):
    """batotp_amd/host/_build/batest_batch (BA::optimizeBatch over the HIP library, device resampler and device output
    stage where the configuration allows them) writes the reference binary's files for every copy of the path"""
    import filecmp, os, shutil, subprocess
    exe = os.path.join(helpers.ROOT, "batotp_amd", "host", "_build", "batest_batch")
    assert os.path.exists(exe), "build() must produce batest_batch"
    src = os.path.join(helpers.GOLD, name)
    for f in os.listdir(src):
        if not f.startswith("ref_") and not f.endswith(".npz") and not f.endswith(".json"):
            shutil.copy(os.path.join(src, f), tmp_path / f)
    r = subprocess.run([exe, "config.dat", "5"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(tmp_path / d / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False), name
        assert filecmp.cmp(tmp_path / d / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False), name


def test_product_batch_driver_over_all_devices_of_the_box(tmp_path):
    """batest_batch --all-devices: BA::useAllDevices + BA::optimizeBatch over every GPU the box has (contiguous blocks of paths, a
    host thread and a device context per GPU; on a one-GPU box that is one block on device 0) -- every path comes back in place
    with the single-path files"""
    import filecmp, os, shutil, subprocess
    exe = os.path.join(helpers.ROOT, "batotp_amd", "host", "_build", "batest_batch")
    name = "synth_gen7dof_s0"
    src = os.path.join(helpers.GOLD, name)
    for f in os.listdir(src):
        if not f.startswith("ref_") and not f.endswith(".npz") and not f.endswith(".json"):
            shutil.copy(os.path.join(src, f), tmp_path / f)
    r = subprocess.run([exe, "config.dat", "9", "--all-devices"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "9 paths, 0 failed" in r.stdout
    n_gpus = hip_device_count()
    assert n_gpus >= 1 and (n_gpus == 1 or f"sharded over {min(n_gpus, 9)} devices" in r.stdout)
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(tmp_path / d / "s-sdot.dat", os.path.join(src, "ref_s-sdot.dat"), shallow=False)
        assert filecmp.cmp(tmp_path / d / "traj_out.dat", os.path.join(src, "ref_traj_out.dat"), shallow=False)


def hip_device_count():
    return capi.load_hip().device_count()


def test_product_batch_driver_grows_a_shared_curve_buffer(tmp_path):
    """is_sdotOut = 0 (one curve buffer per path, BATOTP_F_CURVES_IN_PLACE) and a forward curve that does not fit the first
    capacity guess while the reverse curve does: the forward kernel gives up 64 points before unread reverse points, with
    steps_fwd below the capacity; BA::optimizeBatch reads that as out of room and runs the batch again -- same trajectory as
    the single-path driver (round-2 advisor finding)"""
    import filecmp, os, re, shutil, subprocess
    from test_batest_cpu import _edit_config, _stage
    build = os.path.join(helpers.ROOT, "batotp_amd", "host", "_build")
    src = os.path.join(helpers.GOLD, "GEN7DOF")
    one, many = tmp_path / "one", tmp_path / "many"
    for d, sdot_out in ((one, "1"), (many, "0")):
        d.mkdir()
        _stage(src, d)
        _edit_config(d / "config.dat", {"integRes": "0.00076", "outRes": "0.004", "is_sdotOut": sdot_out})
    r1 = subprocess.run([os.path.join(build, "batest"), "config.dat"], cwd=one, capture_output=True, text=True)
    assert r1.returncode == 0, r1.stdout[-2000:]
    fwd = int(re.search(r"fwd\. integ\.:\s*(\d+) steps", r1.stdout).group(1))
    rev = int(re.search(r"rev\. integ\.:\s*(\d+) steps", r1.stdout).group(1))
    cap = 8 * 231 + 4096
    assert rev + 1 < cap <= fwd + 66, (rev, fwd, cap)
    r = subprocess.run([os.path.join(build, "batest_batch"), "config.dat", "3"], cwd=many, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "running the batch again" in r.stdout and "3 paths, 0 failed" in r.stdout
    for d in ("out_first", "out_last"):
        assert filecmp.cmp(many / d / "traj_out.dat", one / "traj_out.dat", shallow=False)


def test_chunked_resampling_equals_one_chunk(hip_lib, oracle_ctx):
    """a tiny scratch budget forces several chunks of paths: the result must not depend on the chunking"""
    c = ResampleCase("synth_cspr_s3")
    xs = [c.x, c.x[:, :900].copy(), c.x[:, ::-1].copy(), c.x[:, 500:2500].copy(), c.x]
    sr = [c.sres_in] * len(xs)
    ctx = capi.Context(hip_lib, 0)
    ctx.set_workspace_budget(resample_bytes=6 << 20)
    h = capi.Resampled(ctx, c.params, xs, sr)
    o = capi.Resampled(oracle_ctx, c.params, xs, sr)
    _same(h, o, "chunked cspr batch")
    assert_bit_equal(h.knots(4), c.y, "last path of a chunked batch")


def test_knots_of_an_older_resample_call_are_refused(hip_lib):
    """the knots live in a workspace the context reuses: after the next resample call (or ctx_trim) the older object
    answers BATOTP_ERR_STATE instead of handing out memory that now belongs to someone else"""
    c = ResampleCase("synth_gen7dof_s1_vel")
    ctx = capi.Context(hip_lib, 0)
    first = capi.Resampled(ctx, c.params, [c.x], [c.sres_in])
    assert_bit_equal(first.knots(0), c.y, "first call")
    second = capi.Resampled(ctx, c.params, [c.x, c.x], [c.sres_in] * 2)
    with pytest.raises(capi.BatotpError):
        first.knots(0)
    with pytest.raises(capi.BatotpError):
        first.device_ptr()
    assert_bit_equal(second.knots(1), c.y, "second call")
    ctx.trim()
    with pytest.raises(capi.BatotpError):
        second.knots(0)
    third = capi.Resampled(ctx, c.params, [c.x], [c.sres_in])   # the workspaces are allocated again on demand
    assert_bit_equal(third.knots(0), c.y, "after trim")


import os
from test_oracle_resample import AUTORES_CASES, autores_params


@pytest.mark.parametrize("name", AUTORES_CASES)
def test_device_resampler_automatic_integration_resolution(hip_ctx, oracle_ctx, name):
    """the automatic integration resolution (reference ba.cpp:462-470, 493-556; class default ba.h:309) in the device
    resampler, per path: a ragged batch (the case's path, its first half, the path reversed) against the oracle, and the
    case's own path against tests/golden/<case>/autores.npz (knots, integration step, s weights, scale type)"""
    c = ResampleCase(name)
    z = np.load(os.path.join(helpers.GOLD, name, "autores.npz"))
    p = autores_params(c, Case(name).problem)
    n = c.x.shape[1]
    xs = [c.x, c.x[:, :max(8, n // 2)].copy(), c.x[:, ::-1].copy()]
    sr = [c.sres_in] * len(xs)
    h = capi.Resampled(hip_ctx, p, xs, sr)
    o = capi.Resampled(oracle_ctx, p, xs, sr)
    _same(h, o, f"{name}: automatic integration resolution")
    hi, hw, hs = h.auto()
    oi, ow, os_ = o.auto()
    assert hi.tobytes() == oi.tobytes() and hw.tobytes() == ow.tobytes() and np.array_equal(hs, os_), (hi, oi, hw, ow, hs, os_)
    assert np.array_equal(np.array([hi[0]]).view(np.uint64), np.array([float(z["integ_res"])]).view(np.uint64))
    assert hw[0].tobytes() == np.asarray(z["s_weights"], dtype=np.float64).tobytes() and int(hs[0]) == int(z["scale_type"])
    assert_bit_equal(h.knots(0), np.ascontiguousarray(z["y"]), f"{name}: knots under the automatic integration resolution")
    h.close(); o.close()


def test_short_paths_are_stretched_to_four_points(hip_ctx, oracle_ctx):
    """BA::interpTrajLinear (reference ba.cpp:182-183, 773-774, 2768-2794): a path that remClosePts leaves with two or three points,
    and a path so short that interpSpecial emits fewer than four, are stretched onto four points and go on"""
    c = ResampleCase("synth_gen7dof_s0")
    a = np.repeat(c.x[:, :3], 4, axis=1)                       # 12 taught points, 3 distinct
    b = np.repeat(c.x[:, :2], 5, axis=1)                       # 10 taught points, 2 distinct
    tiny = c.x[:, :6].copy()
    tiny[:, 1:] = tiny[:, :1] + (tiny[:, 1:] - tiny[:, :1]) * 0.02   # six points within a fraction of one resolution step
    one = np.repeat(c.x[:, :1], 8, axis=1)                     # a single distinct point: nothing to optimise
    xs, sr = [a, b, tiny, one, c.x], [c.sres_in] * 5
    h = capi.Resampled(hip_ctx, c.params, xs, sr)
    o = capi.Resampled(oracle_ctx, c.params, xs, sr)
    _same(h, o, "short paths")
    assert int(h.status[3]) != 0 and int(h.status[4]) == 0
    assert_bit_equal(h.knots(4), c.y, "the full path beside them is untouched")
    h.close(); o.close()


@pytest.mark.parametrize("workload,knots,n_seeds", [("gen7", 6000, 64), ("cspr", 8000, 48), ("kuka7trq", 6000, 32)])
def test_concurrent_one_path_resampling_equals_the_oracle(workload, knots, n_seeds):
    """Regression for the red driver run of round 4: BA::interpInputData of ONE path goes through batotp_hip_resample with a
    batch of one -- `baknots`, a process and a HIP context per path, 64 of them at a time on one GPU (what bench.py and the
    BASELINE-size tests did when one of 1024 such calls came back with different knots and rc 0).  Every knots.bin must be
    the ORACLE resampler's (oracle/_build/dump_knots: the same host shell over the CPU checker), bit for bit; a call that
    fails must say so through its exit code."""
    import concurrent.futures as cf
    import hashlib
    import os
    import subprocess
    import tempfile
    import bench

    def digest(tool, seed):
        w = bench.WORKLOADS[workload]
        n_coarse = max(8, int(round(knots / w["knots_per_coarse"])))
        theta, cart, tres = w["gen"](seed, n_coarse)
        with tempfile.TemporaryDirectory() as work:
            pathgen.write_traj_bin(os.path.join(work, "path.dat"), tres, theta, cart)
            pathgen.write_config(os.path.join(work, "config.dat"), **w["cfg"])
            r = subprocess.run([tool, "config.dat"], cwd=work, capture_output=True, text=True)
            assert r.returncode == 0, (workload, seed, os.path.basename(tool), r.stdout[-600:])
            return hashlib.sha256(open(os.path.join(work, "knots.bin"), "rb").read()).hexdigest()

    seeds = [31000 + k for k in range(n_seeds)]
    jobs = min(64, os.cpu_count() or 1)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        want = list(ex.map(lambda s: digest(bench.ORACLE_KNOTS, s), seeds))
    for rnd in range(1):
        with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
            got = list(ex.map(lambda s: digest(bench.BAKNOTS, s), seeds))
        bad = [s for s, a, b in zip(seeds, want, got) if a != b]
        assert not bad, f"{workload}: one-path device resampling differs from the oracle's knots for seeds {bad} (round {rnd}, {jobs} concurrent processes)"
