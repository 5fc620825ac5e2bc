#!/usr/bin/env python3
"""Run the hot path once or a few times on the GPU for profiling (rocprofv3 -- python3 tools/run_hotpath.py ...)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from batotp_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="ur6"); ap.add_argument("--paths", type=int, default=256)
ap.add_argument("--knots", type=int, default=100000); ap.add_argument("--distinct", type=int, default=8)
ap.add_argument("--reps", type=int, default=1); ap.add_argument("--group", type=int, default=0)
ap.add_argument("--ppw", type=int, default=0)
ap.add_argument("--coefficient-rows", action="store_true")
ap.add_argument("--lib", default=None, help="alternative build of the library (experiments)")
ap.add_argument("--hold", type=int, nargs=2, default=None, help="sweep loop form, reverse forward (batotp_hip_set_sweep_hold)")
ap.add_argument("--flat-form", type=int, default=None, help="0 = flat instantiation of k_sweep, 1 = k_sweep8 (batotp_hip_set_flat_form)")
ap.add_argument("--fast-forward", type=int, default=None, help="batotp_hip_set_fast_forward: 0 off, 1 on (default)")
ap.add_argument("--tag", default="", help="label printed in front of every line")
ap.add_argument("--variants", default=None, help="A/B of sweep forms on ONE resident batch: comma-separated form:holdRev:holdFwd[:ppw[:group[:certHold]]] "
                "(form 0 = k_sweep's flat instantiation, 1 = k_sweep8; hold -1 = nested loops; certHold: batotp_hip_set_cert_hold); prints kernel times and digests of rows / curves")
a = ap.parse_args()
hip = capi.Context(capi.Library(a.lib) if a.lib else capi.load_hip(), 0)
hip.set_sweep_group(a.group)
if a.paths > a.distinct:
    hip.set_path_order(0)   # tiled copies of a path must not become neighbours in a wavefront (they would run in lockstep)
if a.ppw and hasattr(hip, "set_paths_per_wave"):
    hip.set_paths_per_wave(a.ppw)
base = [bench.make_knots(a.workload, 1000 + k, a.knots) for k in range(a.distinct)]
nk = [base[p % a.distinct][0].shape[1] for p in range(a.paths)]
cap = int(max(nk) * bench.WORKLOADS[a.workload]["cap"]) + 1024
prob = base[0][2]
if (prob.flags & capi.F_NO_SAMPLES) and not a.coefficient_rows:
    prob.flags |= capi.F_COMPACT_SPLINES
b = capi.Batch(hip, prob, nk, cap)
for p in range(a.paths):
    b.upload_knots(p, [base[p % a.distinct][0]], [base[p % a.distinct][1]])
bench.prepare_dynamics(b, prob, a.paths)
if a.hold:
    hip.set_sweep_hold(*a.hold)
if a.flat_form is not None:
    hip.set_flat_form(a.flat_form)
if a.fast_forward is not None:
    hip.set_fast_forward(a.fast_forward)
import hashlib
for _ in range(a.reps):
    t = time.perf_counter(); b.precompute(0); b.pointwise_mvc(); b.sweep(-1); b.sweep(1); dt = time.perf_counter() - t
    r = b.results()
    dig = hashlib.sha256(np.ascontiguousarray(r).tobytes()).hexdigest()[:12]
    print(a.tag, f"rows {dig} launch rev {b.last_sweep_launch(-1)} fwd {b.last_sweep_launch(1)}")
    print(a.tag, f"step {dt*1e3:.1f} ms  pre {b.kernel_ms(1):.1f} mvc {b.kernel_ms(2):.1f} rev {b.kernel_ms(3):.1f} fwd {b.kernel_ms(4):.1f}  "
          f"wp/s {sum(nk)/dt:.3e}  steps rev {int(r['steps_rev'].sum())} fwd {int(r['steps_fwd'].sum())} status {int((r['status_rev']|r['status_fwd']).max())}")

if a.variants:
    def digest():
        r = b.results()
        h = hashlib.sha256(np.ascontiguousarray(r).tobytes())
        for p in sorted({0, a.paths // 3, a.paths - 1}):
            for which in (-1, 1):
                s_, sd_ = b.curve(p, which)
                h.update(s_.tobytes()); h.update(sd_.tobytes())
        return h.hexdigest()[:12], int(r["steps_rev"].sum()), int(r["steps_fwd"].sum()), int((r["status_rev"] | r["status_fwd"]).max())
    for v in a.variants.split(","):
        f = [int(x) for x in v.split(":")]
        hip.set_flat_form(f[0]); hip.set_sweep_hold(f[1], f[2])
        if len(f) > 3: hip.set_paths_per_wave(f[3])
        if len(f) > 4: hip.set_sweep_group(f[4])
        if len(f) > 5: hip.set_cert_hold(f[5])
        best = [1e30, 1e30]
        for _ in range(max(1, a.reps)):
            b.sweep(-1); best[0] = min(best[0], b.kernel_ms(3))
            b.sweep(1); best[1] = min(best[1], b.kernel_ms(4))
        print(f"variant {v:12s} rev {best[0]:9.1f} ms  fwd {best[1]:9.1f} ms  launch {b.last_sweep_launch(-1)} {b.last_sweep_launch(1)}  digest {digest()}", flush=True)
