#!/usr/bin/env python3
"""What a wavefront of k_sweep8 spends its passes and cycles on (diagnostic build of the library: -DS8_PROFILE).

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -Iinclude -Ibatotp_amd/csrc -DS8_PROFILE -shared \
          -o /tmp/libs8prof.so batotp_amd/csrc/batotp_hip.hip
    python tools/sweep8_sections.py --lib /tmp/libs8prof.so --paths 16384

The headline batch of bench.py (GEN7DOF, N ~ 1e5, curves in place, pointwise values in the curve slots); per wavefront the
kernel leaves 16 counters (sweep8.hip.h, S8_PROFILE)."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from batotp_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lib", required=True)
ap.add_argument("--workload", default="gen7")
ap.add_argument("--paths", type=int, default=16384)
ap.add_argument("--knots", type=int, default=100000)
ap.add_argument("--distinct", type=int, default=256)
ap.add_argument("--hold", type=int, nargs=2, default=None)
ap.add_argument("--cert-hold", type=int, default=None, help="batotp_hip_set_cert_hold (reverse sweep: the certificate phase)")
a = ap.parse_args()

lib = capi.Library(a.lib)
hip = capi.Context(lib, 0)
if a.hold:
    hip.set_sweep_hold(*a.hold)
if a.cert_hold is not None:
    hip.set_cert_hold(a.cert_hold)
if a.paths > a.distinct:
    hip.set_path_order(0)   # tiled copies of a path must not become neighbours in a wavefront (they would run in lockstep)
inp = bench.Inputs(hip, a.workload, a.knots, [1000 + k for k in range(a.distinct)])
prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
prob.flags |= capi.F_COMPACT_SPLINES | capi.F_CURVES_IN_PLACE | capi.F_MVC_IN_CURVES
cap = int(int(inp.n_knots.max()) * bench.WORKLOADS[a.workload]["cap"]) + 1024
b = capi.Batch(hip, prob, [int(inp.n_knots[p % inp.K]) for p in range(a.paths)], cap)
inp.fill(b, a.paths)
b.precompute(0)
fn = lib.lib.batotp_hip_debug_sweep8_counters
fn.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int64]
names = ["passes", "prologue blocks", "paths served by them", "check blocks", "paths inside them", "cycles prologue", "cycles check",
         "cycles loop", "step-end blocks", "segment-change blocks", "cursor walks", "fast-forward blocks", "literal-form blocks", "live paths x passes", "cycles certificate", "paths certified"]
for rep in range(2):
    for d, name in ((-1, "reverse"), (+1, "forward")):
        b.sweep(d)
        if rep == 0:
            continue
        lanes, ppw, hold = b.last_sweep_launch(d)
        nw = (a.paths + ppw - 1) // ppw
        raw = np.zeros(nw * 16)
        lib.check(fn(b.handle, raw.ctypes.data_as(C.POINTER(C.c_double)), raw.size), "debug counters")
        c = raw.reshape(nw, 16).sum(axis=0)
        res = b.results()
        steps = float((res["steps_rev"] if d < 0 else res["steps_fwd"]).sum())
        wstages = 6.0 * steps / ppw              # wavefront-stages: a stage of the ppw paths of a wavefront
        print(f"{name}: {b.kernel_ms(3 if d < 0 else 4):.1f} ms, launch (lanes, paths per wavefront, hold) = {(lanes, ppw, hold)}, {nw} wavefronts, "
              f"{steps:.3e} integration steps")
        for k, nm in enumerate(names):
            print(f"   {nm:26s} {c[k]:.4e}   per wavefront-stage {c[k] / wstages:9.3f}")
        print(f"   paths per prologue block {c[2] / max(c[1], 1):.2f}, paths per check block {c[4] / max(c[3], 1):.2f}, live paths per pass {c[13] / max(c[0], 1):.2f}")
        print(f"   cycles per certificate block {c[14] / max(c[11], 1):.0f}, paths per certificate block {c[15] / max(c[11], 1):.2f}, "
              f"checks per path-stage {c[4] / (6.0 * steps):.3f}, share of loop cycles: certificate {c[14] / c[7]:.3f}")
        print(f"   cycles per prologue block {c[5] / max(c[1], 1):.0f}, per check block {c[6] / max(c[3], 1):.0f}; share of loop cycles: prologue "
              f"{c[5] / c[7]:.3f}, check {c[6] / c[7]:.3f}")
