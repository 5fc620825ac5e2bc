#!/usr/bin/env python3
"""Times the device output stage (batotp_hip_output) on a swept batch of bench.py's workloads.

    python tools/bench_output.py --workload gen7 --paths 512 --knots 100000

Prints one JSON line (best of the calls after the first, which allocates the context's workspace).  Parity is the tests'
business (tests/test_gpu_output.py); bench.py compares the rows with the oracle's on every default run."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from batotp_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="gen7")
ap.add_argument("--paths", type=int, default=512)
ap.add_argument("--knots", type=int, default=100000)
ap.add_argument("--distinct", type=int, default=64)
ap.add_argument("--calls", type=int, default=3)
a = ap.parse_args()

hip = capi.Context(capi.load_hip(), 0)
if a.paths > a.distinct:
    hip.set_path_order(0)
inp = bench.Inputs(hip, a.workload, a.knots, [1000 + k for k in range(a.distinct)])
prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
cap = int(int(inp.n_knots.max()) * bench.WORKLOADS[a.workload]["cap"]) + 1024
b = capi.Batch(hip, prob, [int(inp.n_knots[p % inp.K]) for p in range(a.paths)], cap)
inp.fill(b, a.paths)
b.optimize()
wcfg = bench.WORKLOADS[a.workload]["cfg"]
prm = capi.OutputParams(prob.n_joints, capi.PATH_JOINT if wcfg["path_type"] == "JOINT" else capi.PATH_CART, prob.integ_res, 0.008, 5.0)
ms = []
for _ in range(a.calls):
    o = capi.Output(b, prm, 0, a.paths)
    ms.append(o.ms())
    pts = int(o.n_pts.sum())
    o.close()
print(json.dumps({"workload": a.workload, "paths": a.paths, "points": pts, "first_ms": ms[0], "ms": min(ms[1:] or ms),
                  "points_per_s": pts / (min(ms[1:] or ms) * 1e-3)}))
