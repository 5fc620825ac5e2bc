#!/usr/bin/env python3
"""Cycle attribution inside the one-path-per-wavefront sweep kernel (diagnostic build: -DBK_PROFILE_SECTIONS).

    hipcc ... -DBK_PROFILE_SECTIONS -shared -o /tmp/libdiag.so batotp_amd/csrc/batotp_hip.hip
    python tools/sweep1_sections.py --lib /tmp/libdiag.so --workload ur6 --paths 1

Per path the kernel leaves 8 numbers in the first doubles of the K3 output array: cycles in the velocity limit, the spline
evaluation, the first constraint check, the bisection passes, the whole step loop; stages, stages that bisect, bisection passes."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from batotp_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True); ap.add_argument("--workload", default="ur6"); ap.add_argument("--paths", type=int, default=1)
ap.add_argument("--knots", type=int, default=50000); ap.add_argument("--distinct", type=int, default=1)
a = ap.parse_args()
hip = capi.Context(capi.Library(a.lib), 0)
hip.set_sweep_group(64)
base = [bench.make_knots(a.workload, 1000 + k, a.knots) for k in range(a.distinct)]
nk = [base[p % a.distinct][0].shape[1] for p in range(a.paths)]
cap = int(max(nk) * bench.WORKLOADS[a.workload]["cap"]) + 1024
prob = base[0][2]
if prob.flags & capi.F_NO_SAMPLES:
    prob.flags |= capi.F_COMPACT_SPLINES
b = capi.Batch(hip, prob, nk, cap)
for p in range(a.paths):
    b.upload_knots(p, [base[p % a.distinct][0]], [base[p % a.distinct][1]])
b.precompute(1)
bench.prepare_dynamics(b, prob, a.paths)   # serial robots with a chain model: cos / sin of the joint angles from the host
if prob.dyn_dim:
    b.precompute(2)
for d, name in ((-1, "reverse"), (+1, "forward")):
    b.sweep(d)
    raw = b.mvc(0)[0][: 8 * a.paths].reshape(a.paths, 8)
    tot = raw[:, 4].sum()
    st, bis, ps = raw[:, 5].sum(), raw[:, 6].sum(), raw[:, 7].sum()
    print(f"{name}: {b.kernel_ms(3 if d < 0 else 4):.1f} ms; {tot / st:.0f} cycles per stage; shares: velocity limit {raw[:,0].sum()/tot:.3f}, "
          f"spline evaluation {raw[:,1].sum()/tot:.3f}, first check {raw[:,2].sum()/tot:.3f}, bisection passes {raw[:,3].sum()/tot:.3f}, "
          f"rest {1 - raw[:,:4].sum()/tot:.3f}; stages that bisect {bis/st:.3f}, passes per bisecting stage {ps/max(bis,1):.2f}, "
          f"cycles per pass {raw[:,3].sum()/max(ps,1):.0f}")
    if a.paths == 1:
        ex = b.mvc(0)[0][8:12]
        if ex[3] > 0:
            print(f"   certified fast-forward of the bisection: {ex[3]:.0f} of {bis:.0f} bisecting stages ended by it (fast-forward + one real check); "
                  f"cycles per bisecting stage: first iteration + x* and error band {ex[0]/bis:.0f}, fast-forward loop {ex[1]/bis:.0f}, "
                  f"the last candidate's real check {ex[2]/bis:.0f}")
