#!/usr/bin/env python3
"""Cycle attribution inside the sweep kernel (diagnostic build of the library: -DBK_PROFILE_SECTIONS).

    hipcc ... -DBK_PROFILE_SECTIONS -shared -o /tmp/libdiag.so batotp_amd/csrc/batotp_hip.hip
    python tools/sweep_sections.py --lib /tmp/libdiag.so --paths 16384

Per path the kernel leaves 4 numbers in the first doubles of the K3 output array: cycles in the velocity limit, in the
spline evaluation, in the bisection / constraint checks, and the total."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from batotp_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None, help="diagnostic build of the library (-DBK_PROFILE_SECTIONS)"); ap.add_argument("--workload", default="ur6"); ap.add_argument("--paths", type=int, default=16384)
ap.add_argument("--knots", type=int, default=100000); ap.add_argument("--distinct", type=int, default=16)
a = ap.parse_args()
hip = capi.Context(capi.Library(a.lib) if a.lib else capi.load_hip(), 0)
base = [bench.make_knots(a.workload, 1000 + k, a.knots) for k in range(a.distinct)]
nk = [base[p % a.distinct][0].shape[1] for p in range(a.paths)]
cap = int(max(nk) * bench.WORKLOADS[a.workload]["cap"]) + 1024
prob = base[0][2]
if prob.flags & capi.F_NO_SAMPLES:
    prob.flags |= capi.F_COMPACT_SPLINES
b = capi.Batch(hip, prob, nk, cap)
for p in range(a.paths):
    b.upload_knots(p, [base[p % a.distinct][0]], [base[p % a.distinct][1]])
b.precompute(0)
for d, name in ((-1, "reverse"), (+1, "forward")):
    b.sweep(d)
    raw = b.mvc(0)[0][: 4 * a.paths].reshape(a.paths, 4)
    tot = raw[:, 3].sum()
    print(f"{name}: {b.kernel_ms(3 if d < 0 else 4):.1f} ms; share of wavefront cycles: velocity limit {raw[:,0].sum()/tot:.3f}, "
          f"spline evaluation {raw[:,1].sum()/tot:.3f}, bisection + checks {raw[:,2].sum()/tot:.3f}, "
          f"rest (RK bookkeeping, stores, control) {1 - raw[:,:3].sum()/tot:.3f}")
