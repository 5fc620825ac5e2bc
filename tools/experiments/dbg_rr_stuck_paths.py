"""see README.md in this directory; needs the patch applied (batotp_hip_set_sweep_hold)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from batotp_amd import capi
from test_gpu_fuzz import _random_knots
seed = 0
rng = np.random.default_rng(3000 + seed)
base = helpers.Case("RR").problem
prob = capi.Problem.from_buffer_copy(bytes(base))
prob.flags = capi.F_TRQ_ON | capi.F_HOST_TRIG | (capi.F_JNT_ACC_ON if seed % 2 else 0)
for j in range(2):
    prob.jnt_vel_max[j] = float(rng.uniform(100, 400)); prob.jnt_acc_max[j] = float(rng.uniform(500, 2000))
    prob.jnt_trq_max[j] = float(rng.uniform(5, 40)); prob.jnt_trq_min[j] = -float(rng.uniform(5, 40))
ys = [_random_knots(rng, 2, int(rng.integers(20, 400)), rng.uniform(20, 120)) for _ in range(int(rng.integers(2, 9)))]
ys = [np.ascontiguousarray(np.vstack([y, np.zeros((prob.n_cart, y.shape[1]))])) for y in ys]
sres = [float(rng.uniform(0.2, 2.0)) for _ in ys]
hip_lib = capi.Library(sys.argv[1])
def run(ctx):
    b = capi.Batch(ctx, prob, [y.shape[1] for y in ys], 30000)
    for k, y in enumerate(ys):
        b.upload_knots(k, [y], [sres[k]])
    b.precompute(1)
    for k in range(len(ys)):
        b.upload_rr_trig(k, helpers.rr_trig(b.samples(k, 0)[0], b.samples(k, 1)[0]))
    b.precompute(2)
    b.sweep(-1)
    r = b.results()
    cur = [(b.curve(k, -1), b.curve(k, 1)) for k in range(len(ys))]
    b.close()
    return r, cur
for lanes, ppw, hr, hf in ((8, 8, -1, -1), (8, 8, 6, -1), (8, 8, 4, -1)):
    ctx = capi.Context(hip_lib, 0)
    ctx.set_sweep_group(lanes); ctx.set_paths_per_wave(ppw); ctx.set_sweep_hold(hr, hf)
    r, c = run(ctx)
    print(sys.argv[1][-12:], (lanes, ppw, hr, hf), r["n_bisect_fail_rev"], [hex(x) for x in r["status_rev"]])
    ctx.close()
