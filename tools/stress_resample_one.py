#!/usr/bin/env python3
"""Stress test of the WHOLE one-path device resampler for the transient wrong-knots event (profiles/r06_i_*), at many times the call rate of
the soak (tools/repro_concurrent_resample.py starts a process per call; here a process stays and calls in a loop).

J processes at a time, a HIP context each, every process resampling ITS OWN path (bench.py's GEN7DOF generator, ~5e4 knots, the soak's size)
with the stage trace on (batotp_hip_set_resample_trace); every call's eight stage checksums are compared with those of the process's first
call.  Both events of the soak were the SECOND evaluation of a fresh process, so the three ages of a context are all exercised: a process
lives --life seconds and is replaced by a new one (wave after wave until --seconds are over), and inside a process the context is closed and
re-created every --calls-per-context calls.  Optionally beside a resident batch that sweeps in a loop.

The resampler is a deterministic function of its input: ANY difference is a fault below the source level.  A mismatch is characterised: age of
the process and of the context in calls, the first stage whose checksum differs, and -- from the host copies of the stage-2 and stage-3 arrays
the trace keeps -- how many values differ, in which rows, first / last index, the size of the difference; both arrays go to --out as .npy.

usage: stress_resample_one.py [--alternate] [--jobs 64] [--seconds 300] [--life 60] [--calls-per-context 8] [--knots 50000] [--resident 1024] [--out gpurun_out/stress_rs]
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

STAGES = ["taught points", "their sites", "their second derivatives", "emitted points", "stage-1 points", "stage-1 sites", "stage-1 second derivatives", "knots"]


def worker(k, wave, knots, life, per_ctx, out, q, alternate=False):
    import bench
    from batotp_amd import capi
    seed = 7000 + k
    # --alternate: two different paths take turns, so that every address of the workspaces holds the OTHER path's data from the call
    # before -- a repetition of ONE path cannot see a stale read (a cache line or a buffer that was not refreshed holds the right values
    # anyway); the soak's wrong evaluations were both the second of their process, the one whose workspaces have just moved
    seeds = [seed, seed + 5000] if alternate else [seed]
    taught, sres_in = bench.taught_points_f32("gen7", seeds, knots)
    xs = [bench.widen("gen7", t) for t in taught]
    prm = bench.resample_params("gen7", capi.Problem())      # (a joint path of a generic robot: the pose matrix is not used)
    lib = capi.load_hip()
    calls, ctxs, bad = 0, 0, []
    refs, refs_data = [None] * len(xs), [None] * len(xs)
    t0 = time.time()
    while time.time() - t0 < life:
        ctx = capi.Context(lib, 0)
        ctx.set_resample_trace(True)
        ctxs += 1
        for j in range(per_ctx):
            w = calls % len(xs)
            x, ref, ref_data = xs[w], refs[w], refs_data[w]
            r = capi.Resampled(ctx, prm, [x], [sres_in])
            t = r.trace()
            calls += 1
            if ref is None:
                refs[w], refs_data[w] = t.copy(), (r.trace_data(2), r.trace_data(3), int(r.n_knots[0]), float(r.sres[0]))
            elif t.tobytes() != ref.tobytes():
                first = int(np.nonzero(t != ref)[0][0])
                rec = dict(worker=k, wave=wave, seed=seeds[w], call_of_process=calls, call_of_context=j + 1, context_of_process=ctxs,
                           first_stage=first, first_stage_name=STAGES[first], stages_differ=[int(v) for v in np.nonzero(t != ref)[0]],
                           n_knots=[ref_data[2], int(r.n_knots[0])], sres=[ref_data[3], float(r.sres[0])], status=int(r.status[0]))
                for st, good in ((2, ref_data[0]), (3, ref_data[1])):
                    got = r.trace_data(st)
                    tag = f"stage{st}"
                    if got.size != good.size:
                        rec[tag] = dict(sizes=[int(good.size), int(got.size)])
                    else:
                        d = np.nonzero(got.view(np.uint64) != good.view(np.uint64))[0]
                        if d.size:
                            C = x.shape[0]
                            n = got.size // C
                            rec[tag] = dict(n_diff=int(d.size), per_row={int(c): int(((d // n) == c).sum()) for c in np.unique(d // n)},
                                            first_in_row=int(d[0] % n), last_in_row=int(d[-1] % n), n_per_row=int(n),
                                            max_abs=float(np.nanmax(np.abs(got[d] - good[d]))), nan=int(np.isnan(got[d]).sum()),
                                            first_values=[[float(good[i]), float(got[i])] for i in d[:4]])
                        else:
                            rec[tag] = dict(n_diff=0)
                    if not bad:                                            # (the first event of a process keeps its arrays: ~10 MB)
                        np.save(os.path.join(out, f"w{wave}_k{k}_call{calls}_{tag}_good.npy"), good)
                        np.save(os.path.join(out, f"w{wave}_k{k}_call{calls}_{tag}_bad.npy"), got)
                bad.append(rec)
            r.close()
        ctx.close()
    q.put(dict(worker=k, calls=calls, contexts=ctxs, bad=bad))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--life", type=float, default=60.0)
    ap.add_argument("--calls-per-context", type=int, default=8)
    ap.add_argument("--knots", type=int, default=50000)
    ap.add_argument("--alternate", action="store_true", help="every process alternates between two different paths (sees stale reads)")
    ap.add_argument("--resident", type=int, default=1024, help="paths of a resident batch sweeping meanwhile (0: none)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stress_rs"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    stop = threading.Event()
    thr = None
    if a.resident:
        import bench
        from batotp_amd import capi
        hip = capi.Context(capi.load_hip(), 0)
        inp = bench.Inputs(hip, "gen7", a.knots, [7000 + k for k in range(a.resident)])
        prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
        prob.flags |= capi.F_COMPACT_SPLINES
        cap = int(int(inp.n_knots.max()) * bench.WORKLOADS["gen7"]["cap"] * 2) + 1024
        b = capi.Batch(hip, prob, [int(v) for v in inp.n_knots], cap)
        inp.fill(b, a.resident)

        def spin():
            while not stop.is_set():
                b.precompute(0); b.sweep(-1); b.sweep(+1)
        thr = threading.Thread(target=spin)
        thr.start()
        print(f"resident batch of {a.resident} paths sweeping", flush=True)
    mpc = mp.get_context("spawn")
    t0 = time.time()
    wave, calls, ctxs, procs_n, bad = 0, 0, 0, 0, []
    while time.time() - t0 < a.seconds:
        q = mpc.Queue()
        procs = [mpc.Process(target=worker, args=(k, wave, a.knots, a.life, a.calls_per_context, a.out, q, a.alternate)) for k in range(a.jobs)]
        for p in procs:
            p.start()
        res = [q.get(timeout=a.life * 4 + 600) for _ in procs]
        for p in procs:
            p.join(timeout=60)
        calls += sum(r["calls"] for r in res)
        ctxs += sum(r["contexts"] for r in res)
        procs_n += len(procs)
        wb = [b_ for r in res for b_ in r["bad"]]
        bad += wb
        print(f"wave {wave}: {sum(r['calls'] for r in res)} calls in {sum(r['contexts'] for r in res)} contexts of {len(procs)} processes, {len(wb)} differ, {time.time() - t0:.0f} s", flush=True)
        for b_ in wb:
            print(json.dumps(b_), flush=True)
        wave += 1
    stop.set()
    if thr:
        thr.join()
    print(f"total: {calls} traced one-path resamples ({a.knots} knots) in {ctxs} contexts of {procs_n} processes, {a.jobs} at a time, {time.time() - t0:.0f} s: {len(bad)} differ from their process's first")
    json.dump(dict(calls=calls, contexts=ctxs, processes=procs_n, bad=bad), open(os.path.join(a.out, "summary.json"), "w"), indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
