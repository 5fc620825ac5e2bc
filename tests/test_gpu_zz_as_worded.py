"""BASELINE.json's batch configurations AS WORDED, at full size -- collected LAST among the GPU tests (file name), so that
`pytest -x` cannot hide the rest of the suite behind minutes-long tests.

The HIP batch (taught points -> knots by the device resampler -> precompute -> both sweeps) against the oracle chain that is
independent of the product from the taught points on: the ORACLE's resampler (oracle/batotp_oracle_resample.c through
oracle/_build/dump_knots), then the oracle's precompute and sweeps.  Round 4's version of this test fed the oracle with knots
from `baknots` -- the product's own one-path device route -- and went red when one of 1024 concurrent one-path calls returned
different knots (tests/test_gpu_resample.py::test_concurrent_one_path_resampling_equals_the_oracle is the regression for that
route)."""
import concurrent.futures as cf
import os

import numpy as np
import pytest

import helpers
from batotp_amd import capi
from helpers import assert_bit_equal, run_pipeline

pytestmark = pytest.mark.gpu


def _batch_as_worded(hip_lib, oracle_ctx, config, n_paths, sample, distinct):
    """a BASELINE batch configuration as bench.py builds it (distinct seeded paths, taught points -> knots by the device
    resampler): size-independent properties for every path; for EVERY distinct path the knots bit-equal to the ORACLE
    resampler's and the result rows equal to the oracle's (fed by the oracle's resampler); curves for a sample"""
    import bench
    ctx = capi.Context(hip_lib, 0)
    c = bench.CONFIGS[config]
    seeds = [7000 + k for k in range(min(distinct, n_paths))]
    inp = bench.Inputs(ctx, c["workload"], c["knots"], seeds)
    K = inp.K
    prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
    if prob.flags & capi.F_NO_SAMPLES:
        prob.flags |= capi.F_COMPACT_SPLINES
    cap = int(int(inp.n_knots.max()) * bench.WORKLOADS[c["workload"]]["cap"] * 2) + 1024
    b = capi.Batch(ctx, prob, [int(inp.n_knots[p % K]) for p in range(n_paths)], cap)
    inp.fill(b, n_paths)
    b.precompute(0); b.sweep(-1); b.sweep(+1)
    res = b.results()
    ok = ((res["status_rev"] | res["status_fwd"]) & ~np.uint32(capi.ST_BISECT_FAIL)) == 0
    # every distinct path through the oracle (one path per host thread, chunks bounded in memory): the result ROWS of all of
    # them must be equal -- in particular the set of paths that end with an error status (cable tensions the limits do not
    # admit: the reference grinds through such a path and returns -1) is exactly the oracle's, not "at most 3 %"
    import hashlib
    # The oracle chain for EVERY distinct path (round 6; round 5 checked every fifth path of cfg 4): the oracle's resampler runs in this
    # process over all host threads (oracle/abi_shim.c: OpenMP over the paths of a batch) instead of a dump_knots process per path, and so
    # do its precompute and sweeps below.  Independent of the device from the taught points on.
    checked = list(range(K))
    hosts = {}
    if inp.on_device:
        step = 256
        for k0 in range(0, K, step):
            ks = checked[k0:k0 + step]
            ors = capi.Resampled(oracle_ctx, inp.prm, [bench.widen(c["workload"], inp.taught[k]) for k in ks], [inp.sres_in] * len(ks))
            assert not np.any(ors.status), (config, "oracle resampler status")
            for i, k in enumerate(ks):
                hosts[k] = (np.ascontiguousarray(ors.knots(i)[: inp.keep]), float(ors.sres[i]))
            ors.close()
    else:
        with cf.ThreadPoolExecutor(max_workers=min(len(checked), os.cpu_count() or 1, 64)) as ex:
            hosts = dict(zip(checked, ex.map(inp.oracle_knots, checked)))      # CPU only: no device call in the checker's chain
    # resampling at BASELINE size, device against oracle: the knots the batch holds, bit for bit
    digs = inp.device_knot_digests(K)
    for k in checked:
        dig, n, sres = digs[k]
        assert n == hosts[k][0].shape[1] and sres == hosts[k][1], (config, "knot count / spacing of distinct path", k, n, hosts[k][0].shape[1])
        assert dig == hashlib.sha256(np.ascontiguousarray(hosts[k][0]).tobytes()).hexdigest(), (config, "knots of distinct path", k)
    pr = capi.Problem.from_buffer_copy(bytes(inp.prob))
    chunk = max(1, min(len(checked), int((24 << 30) / (1.5 * bench.bytes_per_path(pr, bench.WORKLOADS[c["workload"]]["C"], float(inp.n_knots.mean()), cap)))))
    for k0 in range(0, len(checked), chunk):
        ks = checked[k0:k0 + chunk]
        ob = capi.Batch(oracle_ctx, pr, [hosts[k][0].shape[1] for k in ks], cap)
        for i, k in enumerate(ks):
            ob.upload_knots(i, [hosts[k][0]], [hosts[k][1]])
        bench.prepare_dynamics(ob, pr, len(ks))
        ob.precompute(0); ob.sweep(-1); ob.sweep(+1)
        orows = ob.results()
        ob.close()
        for i, k in enumerate(ks):
            assert res[k] == orows[i], (config, "distinct path", k, res[k], orows[i])
    assert ok.mean() > 0.9, f"{(~ok).sum()} of {n_paths} paths failed: the generator is meant to produce mostly feasible paths"
    assert np.all(res["t_total"][ok] == prob.integ_res * res["steps_fwd"][ok])           # T is quantised to the step
    assert np.all(res["n_fwd"][ok] == res["steps_fwd"][ok] + 1) and np.all(res["n_rev"][ok] == res["steps_rev"][ok] + 1)
    for p in range(K, n_paths):                                                          # tiled copies give identical rows
        assert res[p] == res[p % K]
    for p in sample:
        y, sres = hosts[p % K]
        class _C:
            name = f"{config}:{p}"
        cs = _C()
        cs.y, cs.sres, cs.problem, cs.n = y, sres, inp.prob, y.shape[1]
        cs.max_steps = lambda: cap
        oo = run_pipeline(oracle_ctx, [cs], mvc=False, details=False)[0]
        for f in res.dtype.names:
            assert res[p][f] == oo["result"][f], (config, p, f)
        if ok[p]:
            for which, key in ((-1, "rev"), (1, "fwd")):
                s, sd = b.curve(p, which)
                assert_bit_equal(s, oo[key][0], f"{config} path {p} {key}.s")
                assert_bit_equal(sd, oo[key][1], f"{config} path {p} {key}.sdot")
                assert s[0] == 0.0 and np.all(np.diff(s) > 0)
    # the same batch again with every bisection iteration checked (batotp_hip_set_fast_forward 0): every result row and the
    # sampled curves must come out the same -- the certified fast-forward never changes a result, at full size either
    kept = {p: (b.curve(p, -1), b.curve(p, 1)) for p in sample if ok[p]}
    ctx.set_fast_forward(False)
    b.precompute(0); b.sweep(-1); b.sweep(+1)
    res2 = b.results()
    for f in res.dtype.names:
        assert np.array_equal(res2[f], res[f]), (config, "fast-forward off", f)
    for p, (rev, fwd) in kept.items():
        for which, before in ((-1, rev), (1, fwd)):
            s, sd = b.curve(p, which)
            assert_bit_equal(s, before[0], f"{config} path {p} curve {which} s, fast-forward off")
            assert_bit_equal(sd, before[1], f"{config} path {p} curve {which} sdot, fast-forward off")
    if prob.flags & capi.F_COMPACT_SPLINES and not (prob.flags & (capi.F_TRQ_ON | capi.F_CART_VEL_ON | capi.F_CART_ACC_ON)):
        # ... and through the batch kernel of the headline (k_sweep8: 8 lanes per path, 8 paths per wavefront, flat loop) with the
        # certificate phase of its REVERSE sweep at the automatic hold, at hold 1 and switched off (batotp_hip_set_cert_hold), and with
        # every fast-forward off: the rows of every path and the sampled curves are those of the one-path-per-wavefront kernel above
        for cert, ff in ((-1, True), (1, True), (0, True), (-1, False)):
            ctx.set_fast_forward(ff)
            ctx.set_sweep_group(8); ctx.set_paths_per_wave(8); ctx.set_sweep_hold(4, 8); ctx.set_cert_hold(cert)
            b.precompute(0); b.sweep(-1); b.sweep(+1)
            assert b.last_sweep_launch(-1) == (8, 8, 4) and b.last_sweep_launch(+1) == (8, 8, 8)
            res3 = b.results()
            for f in res.dtype.names:
                assert np.array_equal(res3[f], res[f]), (config, "k_sweep8, certificate hold", cert, "fast-forward", ff, f)
            for p, (rev, fwd) in kept.items():
                for which, before in ((-1, rev), (1, fwd)):
                    s, sd = b.curve(p, which)
                    assert_bit_equal(s, before[0], f"{config} path {p} curve {which} s, k_sweep8 cert {cert} ff {ff}")
                    assert_bit_equal(sd, before[1], f"{config} path {p} curve {which} sdot, k_sweep8 cert {cert} ff {ff}")
        ctx.set_sweep_group(0); ctx.set_paths_per_wave(0); ctx.set_sweep_hold(-2, -2); ctx.set_cert_hold(-1); ctx.set_fast_forward(True)
    b.close()
    if (prob.flags & capi.F_PARALLEL) and (prob.flags & capi.F_PAR2SER):
        # the layout bench.py runs this configuration in -- every channel as (value, second derivative) pairs, one curve buffer per
        # path with the pointwise values in it -- through k_sweep1 with ONE and with TWO paths per wavefront (what a GPU's whole
        # share of 4096 paths gets): the result rows of every path are those of the coefficient-row batch above
        lean = capi.Problem.from_buffer_copy(bytes(prob))
        lean.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES | capi.F_CURVES_IN_PLACE | capi.F_MVC_IN_CURVES
        cap2 = max(cap, int(1.5 * int(inp.n_knots.max())) + 64)
        for ppw in (1, 2):
            ctx.set_fast_forward(True)
            ctx.set_sweep_group(64)
            ctx.set_paths_per_wave(ppw)
            b2 = capi.Batch(ctx, lean, [int(inp.n_knots[p % K]) for p in range(n_paths)], cap2)
            inp.fill(b2, n_paths)
            b2.precompute(0); b2.pointwise_mvc(); b2.sweep(-1); b2.sweep(+1)
            assert b2.last_sweep_launch(+1)[:2] == (64, ppw)
            r2 = b2.results()
            for f in res.dtype.names:
                assert np.array_equal(r2[f], res[f]), (config, "pairs for all channels", ppw, f)
            b2.close()
    ctx.trim()
    ctx.close()


def test_cfg4_batch_as_worded(hip_lib, oracle_ctx):
    """BASELINE config 4 as worded: GEN7DOF, N = 50k, a batch of 1024 randomised (distinct) paths"""
    _batch_as_worded(hip_lib, oracle_ctx, "cfg4", 1024, [0, 1, 511, 777, 1023], distinct=1024)


def test_cfg5_share_as_worded(hip_lib, oracle_ctx):
    """BASELINE config 5 as worded, one GPU's share of the 4096-path batch at 8 GPUs: 512 cable-robot paths of 200k knots"""
    _batch_as_worded(hip_lib, oracle_ctx, "cfg5", 512, [0, 77, 300], distinct=96)
