#!/usr/bin/env python3
"""Focused stress test for the kernel the traced wrong-knots event pointed at (profiles/r06_i_*): the Thomas solve of ONE long series,
by the wavefront-per-series kernel (spline_lanes.hip.h) and by the lane-per-series kernel (kernels.hip.h: thomas_series), through
batotp_hip_spline_lanes_kat -- J processes at a time (a HIP context each, as `baknots` has), every process solving the same series over
and over for T seconds and comparing every result with its first one and the two kernels with each other, optionally beside a resident
batch that sweeps in a loop (what the soak of tools/repro_concurrent_resample.py has).

Both kernels are deterministic functions of their input: ANY difference between two calls is a fault below the source level.  A mismatch
is characterised: which kernel, how many values, first / last index, whether the differing range is one lane's chunk.

usage: stress_spline_lanes.py [--jobs 64] [--seconds 60] [--n 34000] [--resident 1024]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(k, n, seconds, q):
    from batotp_amd import capi
    ctx = capi.Context(capi.load_hip(), 0)
    rng = np.random.default_rng(100 + k)
    t = np.linspace(0, 40, n)
    y = np.sin(t * rng.uniform(0.5, 2.0)) * rng.uniform(1, 5) + 0.01 * rng.standard_normal(n)
    first, first_seq, redone0 = capi.spline_lanes_kat(ctx, y)
    calls, bad = 1, []
    if first.tobytes() != first_seq.tobytes():
        bad.append(dict(call=0, kind="lanes != sequential on the first call", n_diff=int((first.view(np.uint64) != first_seq.view(np.uint64)).sum())))
    t0 = time.time()
    while time.time() - t0 < seconds:
        sol, seq, redone = capi.spline_lanes_kat(ctx, y)
        calls += 1
        for name, arr, ref in (("wavefront-per-series", sol, first), ("lane-per-series", seq, first_seq)):
            if arr.tobytes() != ref.tobytes():
                d = np.nonzero(arr.view(np.uint64) != ref.view(np.uint64))[0]
                Lc = (n - 2) // 64
                bad.append(dict(call=calls, kernel=name, n_diff=int(d.size), first=int(d[0]), last=int(d[-1]), chunk_first=int((d[0] - 1) // Lc),
                                chunk_last=int((d[-1] - 1) // Lc), redone=redone, max_abs=float(np.nanmax(np.abs(arr[d] - ref[d]))),
                                nan=int(np.isnan(arr[d]).sum())))
    q.put(dict(worker=k, calls=calls, bad=bad, redone_first=redone0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=60.0)
    ap.add_argument("--n", type=int, default=34000)
    ap.add_argument("--resident", type=int, default=1024, help="paths of a resident batch sweeping meanwhile (0: none)")
    a = ap.parse_args()
    stop = threading.Event()
    thr = None
    if a.resident:
        import bench
        from batotp_amd import capi
        hip = capi.Context(capi.load_hip(), 0)
        inp = bench.Inputs(hip, "gen7", 50000, [7000 + k for k in range(a.resident)])
        prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
        prob.flags |= capi.F_COMPACT_SPLINES
        cap = int(int(inp.n_knots.max()) * bench.WORKLOADS["gen7"]["cap"] * 2) + 1024
        b = capi.Batch(hip, prob, [int(v) for v in inp.n_knots], cap)
        inp.fill(b, a.resident)
        steps = [0]

        def spin():
            while not stop.is_set():
                b.precompute(0); b.sweep(-1); b.sweep(+1)
                steps[0] += 1
        thr = threading.Thread(target=spin)
        thr.start()
        print(f"resident batch of {a.resident} paths sweeping", flush=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(k, a.n, a.seconds, q)) for k in range(a.jobs)]
    t0 = time.time()
    for p in procs:
        p.start()
    res = [q.get(timeout=a.seconds * 4 + 600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    stop.set()
    if thr:
        thr.join()
    calls = sum(r["calls"] for r in res)
    bad = [dict(worker=r["worker"], **b_) for r in res for b_ in r["bad"]]
    print(f"{a.jobs} processes x {a.seconds:.0f} s: {calls} calls of each kernel on a series of {a.n} values in {time.time() - t0:.0f} s, {len(bad)} results differ")
    for b_ in bad[:40]:
        print(json.dumps(b_))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
