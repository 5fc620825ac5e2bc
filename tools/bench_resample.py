#!/usr/bin/env python3
"""Times the device path resampler (batotp_hip_resample) on bench.py's workloads.

    python tools/bench_resample.py --workload ur6 --paths 1024 --knots 100000

Prints one JSON line: knots per second and milliseconds (cold = first call of the process, which allocates the
context's workspaces; warm = best of the following calls).  Parity is the tests' business (tests/test_gpu_resample.py);
bench.py compares the knots with the host resampler's on every run."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batotp_amd import capi, pathgen  # noqa: E402

KINDS = {
    "ur6": dict(gen=lambda s, n: pathgen.ur_like_fine(s, n), per_coarse=210.8, nJ=6, nC=3, robot=capi.ROBOT_GENJNT, ptype=capi.PATH_JOINT,
                sw=(0, 1, 0), scale=1, tres=(0.3, 0.3), cres=(0.02, 0.02), flags=0, sres=0.01),
    "gen7": dict(gen=lambda s, n: pathgen.gen7dof_fine(s, n), per_coarse=58.2, nJ=7, nC=3, robot=capi.ROBOT_GENJNT, ptype=capi.PATH_JOINT,
                 sw=(0, 1, 0), scale=1, tres=(0.1, 0.1), cres=(0.02, 0.02), flags=0, sres=0.01),
    "cspr": dict(gen=lambda s, n: pathgen.cspr_fine(s, n), per_coarse=217.0, nJ=3, nC=3, robot=capi.ROBOT_CSPR3DOF, ptype=capi.PATH_CART,
                 sw=(0, 0, 1), scale=2, tres=(0.01, 0.01), cres=(0.01, 0.01), flags=capi.F_CART_VEL_ON, sres=0.005),
}


def params(kind, pmat=None):
    k = KINDS[kind]
    p = capi.ResampleParams()
    p.n_joints, p.n_cart, p.robot_type, p.path_type, p.scale_type, p.flags = k["nJ"], k["nC"], k["robot"], k["ptype"], k["scale"], k["flags"]
    for i in range(3):
        p.s_weights[i] = k["sw"][i]
    p.theta_norm_res, p.theta_norm_res2 = k["tres"]
    p.cart_norm_res, p.cart_norm_res2 = k["cres"]
    p.jnt_thresh = p.cart_thresh = 1e-6
    if pmat is not None:
        for i in range(9):
            p.pmat[i] = pmat[i]
    return p


def taught(kind, seed, n_knots):
    k = KINDS[kind]
    n_coarse = max(8, int(round(n_knots / k["per_coarse"])))
    x = k["gen"](seed, n_coarse).astype(np.float32).astype(np.float64)
    full = np.zeros((k["nJ"] + k["nC"], x.shape[1]))
    if k["ptype"] == capi.PATH_CART:
        full[k["nJ"]:] = x
    else:
        full[: k["nJ"]] = x
    return full


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ur6", choices=sorted(KINDS))
    ap.add_argument("--paths", type=int, default=1024)
    ap.add_argument("--knots", type=int, default=100000)
    ap.add_argument("--distinct", type=int, default=16)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--budget-gb", type=float, default=0.0, help="scratch bytes per chunk of paths (batotp_hip_set_workspace_budget); 0 = the library's rule")
    args = ap.parse_args()

    pmat = None
    if args.workload == "cspr":
        z = np.load(os.path.join(ROOT, "tests", "golden", "synth_cspr_s3", "resample.npz"))
        pmat = list(capi.ResampleParams.from_bytes(z["params"].tobytes()).pmat)
    prm = params(args.workload, pmat)
    K = min(args.distinct, args.paths)
    base = [taught(args.workload, 2000 + k, args.knots) for k in range(K)]
    xs = [base[p % K] for p in range(args.paths)]
    sr = [KINDS[args.workload]["sres"]] * args.paths
    pts = int(sum(x.shape[1] for x in xs))

    hip = capi.Context(capi.load_hip(), 0)
    if args.budget_gb > 0:
        hip.set_workspace_budget(resample_bytes=int(args.budget_gb * (1 << 30)))
    best, knots, bad = None, 0, 0
    for rep in range(args.reps):
        t0 = time.perf_counter()
        r = capi.Resampled(hip, prm, xs, sr)
        wall = time.perf_counter() - t0
        ms = r.ms()
        knots = int(r.n_knots.sum())
        bad = int((r.status != 0).sum())
        if best is None or ms < best[0]:
            best = (ms, wall)
        r.close()

    print(json.dumps({
        "workload": args.workload, "paths": args.paths, "taught_points": pts, "knots": knots, "failed_paths": bad,
        "device_ms": round(best[0], 2), "call_wall_ms": round(best[1] * 1e3, 1),
        "device_knots_per_s": round(knots / (best[0] * 1e-3), 1),
    }))


if __name__ == "__main__":
    main()
