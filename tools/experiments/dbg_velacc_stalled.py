"""vel/acc-only paths that stall (one joint has a negative acceleration limit: the bisection fails wherever it moves) through the nested and the
flat sweep loop (patch applied), coefficient rows and compact splines; prints failure counts and a digest of results."""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from batotp_amd import capi
from test_gpu_fuzz import _random_knots

lib = capi.Library(sys.argv[1]) if len(sys.argv) > 1 else capi.load_hip()
for seed in range(3):
    rng = np.random.default_rng(500 + seed)
    nJ = 3
    amax = [float(rng.uniform(5, 30)), -1.0, float(rng.uniform(5, 30))]   # a negative limit: no admissible sddot wherever that joint moves
    prob = capi.make_problem(nJ, 0, flags=capi.F_JNT_ACC_ON, jnt_vel_max=list(rng.uniform(1, 6, nJ)), jnt_acc_max=amax,
                             integ_res=0.01, max_integ_time=1e5)
    ys = [_random_knots(rng, nJ, int(rng.integers(40, 300)), rng.uniform(0.5, 2.0)) for _ in range(7)]
    ys[3][1, :] = 0.25          # this path's second joint stands still: never limited
    for k, f in ((0, 3e-6), (1, 1e-5), (2, 3e-5), (4, 1e-4)):
        ys[k][1] *= f            # hovers around the zero-velocity threshold: fails at some stages only
    sres = [float(rng.uniform(0.02, 0.1)) for _ in ys]
    first = {}
    for compact in (0, 1):
        for hr in (-1, 0, 2, 3, 4, 5, 6, 8):
            ctx = capi.Context(lib, 0)
            ctx.set_sweep_group(8); ctx.set_paths_per_wave(8); ctx.set_sweep_hold(hr, hr)
            p2 = capi.Problem.from_buffer_copy(bytes(prob))
            if compact:
                p2.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
            b = capi.Batch(ctx, p2, [y.shape[1] for y in ys], 8000)
            for k, y in enumerate(ys):
                b.upload_knots(k, [y], [sres[k]])
            b.precompute(0); b.sweep(-1); b.sweep(+1)
            r = b.results()
            h = hashlib.sha256(r.tobytes())
            for k in range(len(ys)):
                for d in (-1, 1):
                    s, sd = b.curve(k, d)
                    h.update(s.tobytes()); h.update(sd.tobytes())
            dg = h.hexdigest()[:12]
            first.setdefault(compact, dg)
            print("seed", seed, "compact", compact, "hold", hr, r["n_bisect_fail_rev"], r["n_bisect_fail_fwd"], r["steps_rev"], dg,
                  "same" if dg == first[compact] else "DIFFERENT", flush=True)
            b.close(); ctx.close()
