#!/usr/bin/env python3
"""stage-by-stage run of one chunk of BASELINE config 5 with every channel as pairs (debugging aid: a marker after every stage)
   usage: python3 tools/repro_pairs.py PATHS KNOTS [rows]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from batotp_amd import capi

B, knots = int(sys.argv[1]), int(sys.argv[2])
rows = len(sys.argv) > 3 and sys.argv[3] == "rows"
lib = capi.Library(os.environ['REPRO_LIB']) if os.environ.get('REPRO_LIB') else capi.load_hip()
hip = capi.Context(lib, 0)
K = int(os.environ.get('REPRO_K', '128'))
inp = bench.Inputs(hip, "cspr", knots, [1000 + k for k in range(K)])
prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
if not rows:
    prob.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
cap = int(int(inp.n_knots.max()) * bench.WORKLOADS["cspr"]["cap"]) * 2 + 1024
say = lambda *a: print(*a, file=sys.stderr, flush=True)
say("inputs", inp.n_knots[:4], "cap", cap, "bytes/path", bench.bytes_per_path(prob, 18, float(inp.n_knots.mean()), cap))
batch = capi.Batch(hip, prob, [int(inp.n_knots[p % K]) for p in range(B)], cap)
say("batch created")
inp.fill(batch, B); hip.synchronize(); say("filled")
for name, f in (("precompute1", lambda: batch.precompute(1)), ("precompute2", lambda: batch.precompute(2)), ("pointwise", batch.pointwise_mvc),
                ("rev", lambda: batch.sweep(-1)), ("fwd", lambda: batch.sweep(+1))):
    t = time.perf_counter(); f(); hip.synchronize(); say(name, "ok %.1f ms" % (1e3 * (time.perf_counter() - t)))
r = batch.results()
say("status", np.unique(r["status_rev"] | r["status_fwd"], return_counts=True), "T", r["t_total"][:3])
