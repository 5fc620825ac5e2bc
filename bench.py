#!/usr/bin/env python3
"""bench.py -- constraint-evaluated waypoints/sec of the batotp hot path on MI355X.

One "step" = one pass of the whole hot path over one batch of synthetic paths whose knot values are
already resident in HBM:
    per-knot precompute (spline coefficients, knot samples [, dynamics])      -> K1/K2
    per-knot max-admissible-sdot evaluation with bisection                    -> K3
    reverse sweep + forward sweep                                             -> K4
Workload (BASELINE.json configs[1]): UR5-like 6-DOF path, joint velocity + acceleration limits only,
N ~ 100k knots per path; the batch holds --paths such paths per GPU (north_star: "synthetic N-point,
B-path batches"), weak scaling over GPUs, no collective in the hot path, one RCCL all_gather of the
per-path result table at the end of every step.

Prints ONE JSON line on rank 0.  Launch: python bench.py [--gpus N --steps K --warmup W], or through
torch.distributed.run for N > 1.
"""
import argparse
import concurrent.futures as cf
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from batotp_amd import capi, pathgen  # noqa: E402
from batotp_amd import dist as bdist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DUMP_KNOTS = os.path.join(ROOT, "batotp_amd", "host", "_build", "baknots")  # the product's host resampler as a tool (no device call)

WORKLOADS = {
    # name: (fine-path generator, config kwargs, coarse points per 1000 knots)
    "ur6": dict(C=6, gen=lambda seed, n: (pathgen.ur_like_fine(seed, n), None, 0.01), knots_per_coarse=210.8,
                cfg=dict(robot="GENJNT", is_parallel=0, n_joints=6, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                         degrees=1, jnt_vel=[160] * 6, jnt_acc_on=1, jnt_acc=[573, 573, 573, 1146, 1146, 1146], integ_res=0.008,
                         max_integ_time=2000000.0, theta_res=0.3, theta_res2=0.3)),
    "gen7": dict(C=7, gen=lambda seed, n: (pathgen.gen7dof_fine(seed, n), None, 0.01), knots_per_coarse=58.2,
                 cfg=dict(robot="GENJNT", is_parallel=0, n_joints=7, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                          degrees=0, jnt_vel=[5] * 7, jnt_acc_on=1, jnt_acc=[10] * 7, integ_res=0.01, max_integ_time=2000000.0,
                          theta_res=0.1, theta_res2=0.1)),
    # BASELINE configs[2]: KUKA LWR IV+ 7-DOF, joint velocity / acceleration limits of the shipped example + the rated joint
    # torques with the chain dynamics of include/batotp_models.h (the reference has no model for this robot: DESIGN.md 5)
    "kuka7trq": dict(C=35, gen=lambda seed, n: (pathgen.kuka_like_fine(seed, n), None, 0.01), knots_per_coarse=235.0,
                     cfg=dict(robot="KUKA", is_parallel=0, n_joints=7, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                              degrees=1, jnt_vel=[110, 110, 128, 128, 204, 184, 184], jnt_acc_on=1,
                              jnt_acc=[137.5, 157.1, 213.3, 213.3, 510.0, 460.0, 613.3], trq_on=1,
                              trq_max=[176, 176, 100, 100, 100, 38, 38], trq_min=[float("nan")] * 7, cart_vel_on=0, cart_vel=0.6,
                              integ_res=0.005, max_integ_time=2000000.0, theta_res=0.3, theta_res2=0.3)),
    # BASELINE configs[4]: cable robot, cable velocity/acceleration/tension limits + Cartesian speed, isPar2Ser=1
    "cspr": dict(C=18, gen=lambda seed, n: (None, pathgen.cspr_fine(seed, n), 0.005), knots_per_coarse=217.0,
                 cfg=dict(robot="CSPR3DOF", is_parallel=1, n_joints=3, n_cart=3, traj_file="path.dat", is_bin=1, path_type="CART",
                          degrees=0, jnt_vel=[4] * 3, jnt_acc_on=1, jnt_acc=[8] * 3, trq_on=1, trq_max=[12] * 3, trq_min=[1] * 3,
                          cart_vel_on=1, cart_vel=4.0, cart_acc_on=0, cart_acc=100.0, integ_res=0.01, max_integ_time=2000000.0,
                          s_weights=(0, 0, 1), scale_type=2, theta_res=0.01, theta_res2=0.01, cart_res=0.01, cart_res2=0.01,
                          par2ser=1)),
}


def make_knots(workload: str, seed: int, n_target: int):
    """one synthetic path -> (y [C_in][N], sres, problem) through the host resampler of the BA library"""
    w = WORKLOADS[workload]
    n_coarse = max(8, int(round(n_target / w["knots_per_coarse"])))
    theta, cart, tres = w["gen"](seed, n_coarse)
    with tempfile.TemporaryDirectory() as work:
        pathgen.write_traj_bin(os.path.join(work, "path.dat"), tres, theta, cart)
        pathgen.write_config(os.path.join(work, "config.dat"), **w["cfg"])
        r = subprocess.run([DUMP_KNOTS, "config.dat"], cwd=work, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("host resampling failed: " + r.stdout[-1000:])
        kb = open(os.path.join(work, "knots.bin"), "rb").read()
        N, nJ, nC = (int(v) for v in np.frombuffer(kb, "<i8", 3, 0))
        sres = float(np.frombuffer(kb, "<f8", 1, 24)[0])
        y = np.frombuffer(kb, "<f8", (nJ + nC) * N, 32).reshape(nJ + nC, N)
        prob = capi.Problem.from_buffer_copy(open(os.path.join(work, "problem.bin"), "rb").read())
    if not (prob.flags & (capi.F_CART_VEL_ON | capi.F_CART_ACC_ON | capi.F_PARALLEL)):
        y = y[:nJ]           # no Cartesian limit, no cable robot: no Cartesian channels are carried (BA::deviceSweep does the same)
        prob.n_cart = 0
        if not (prob.flags & capi.F_TRQ_ON):
            prob.flags |= capi.F_NO_SAMPLES  # knot samples (traj.theta/thetaD/thetaD2) only feed the dynamics model
    return np.ascontiguousarray(y), sres, prob, (theta, cart, tres)


def resample_params(workload: str, prob) -> "capi.ResampleParams":
    """struct batotp_resample_params of a workload's configuration (the fields write_config puts into config.dat)"""
    cfg = WORKLOADS[workload]["cfg"]
    r = capi.ResampleParams()
    r.n_joints, r.n_cart = cfg["n_joints"], cfg["n_cart"]
    r.robot_type = {"GENJNT": capi.ROBOT_GENJNT, "CSPR3DOF": capi.ROBOT_CSPR3DOF}[cfg["robot"]]
    r.path_type = capi.PATH_JOINT if cfg["path_type"] == "JOINT" else capi.PATH_CART
    r.scale_type = cfg.get("scale_type", 1)
    r.flags = (capi.F_CART_VEL_ON if cfg.get("cart_vel_on") else 0) | (capi.F_CART_ACC_ON if cfg.get("cart_acc_on") else 0)
    for i, w in enumerate(cfg.get("s_weights", (0, 1, 0))):
        r.s_weights[i] = w
    r.theta_norm_res, r.theta_norm_res2 = cfg.get("theta_res", 0.1), cfg.get("theta_res2", 0.1)
    r.cart_norm_res, r.cart_norm_res2 = cfg.get("cart_res", 0.02), cfg.get("cart_res2", 0.02)
    r.jnt_thresh, r.cart_thresh = cfg.get("jnt_thresh", 1e-6), cfg.get("cart_thresh", 1e-6)
    for i in range(9):
        r.pmat[i] = prob.pmat[i]
    return r


def measure_resampler(hip, workload, base, n_paths):
    """SURVEY.md 8f-1 beside the hot path: the taught points of the bench paths through batotp_hip_resample;
    its knots must be the ones the host resampler produced for the hot path (bit for bit)"""
    cfg = WORKLOADS[workload]["cfg"]
    nJ, nC = cfg["n_joints"], cfg["n_cart"]
    prm = resample_params(workload, base[0][2])
    xs = []
    for _, _, _, (theta, cart, _) in base:
        n = (theta if theta is not None else cart).shape[1]
        x = np.zeros((nJ + nC, n))
        if theta is not None:
            x[:nJ] = theta.astype(np.float32).astype(np.float64)   # the taught file stores float32
        if cart is not None:
            x[nJ:] = cart.astype(np.float32).astype(np.float64)
        xs.append(x)
    K = len(base)
    tiled = [xs[p % K] for p in range(n_paths)]
    sres_in = [float(np.float32(base[0][3][2]))] * n_paths
    best = None
    for _ in range(2):   # the second call finds the context's workspaces allocated
        r = capi.Resampled(hip, prm, tiled, sres_in)
        ms = r.ms()
        same = all(np.array_equal(r.knots(k)[: base[k][0].shape[0]], base[k][0]) and r.sres[k] == base[k][1] for k in range(min(K, 4)))
        knots = int(r.n_knots.sum())
        r.close()
        best = ms if best is None else min(best, ms)
    hip.trim()
    return {"paths": n_paths, "knots": knots, "ms": best, "knots_per_s": knots / (best * 1e-3),
            "identical_to_host_resampler": bool(same),
            "what": "remClosePts + adjust_s x2 + interpSpecial + uniform re-evaluation on the device (taught points resident)"}


def run_step(batch, has_dyn):
    batch.precompute(0)
    batch.pointwise_mvc()
    batch.sweep(-1)
    batch.sweep(+1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="ur6", choices=sorted(WORKLOADS))
    ap.add_argument("--paths", type=int, default=0,
                    help="paths per GPU (0 = what fills the GPU for the workload: 16384 ur6, 11264 gen7, 2048 cspr)")
    ap.add_argument("--knots", type=int, default=100000, help="target knots per path")
    ap.add_argument("--distinct", type=int, default=32, help="distinct seeded paths per GPU (tiled to --paths)")
    ap.add_argument("--group", type=int, default=0, help="lanes per path in the sweep kernel (0 = automatic)")
    ap.add_argument("--ppw", type=int, default=0, help="paths per wavefront in the sweep kernel (0 = automatic)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="run the per-knot evaluation (K3) beside the sweeps on a second stream (measured: slower, the SIMDs are already saturated)")
    ap.add_argument("--no-resample", action="store_true", help="skip the side measurement of the device path resampler")
    ap.add_argument("--no-output", action="store_true", help="skip the side measurement of the device output stage")
    ap.add_argument("--no-flat-loop", action="store_true", help="skip the side measurement of the optional flat reverse-sweep loop")
    ap.add_argument("--no-seven-dof", action="store_true", help="skip the GEN7DOF (7-DOF) measurement reported beside the default workload")
    ap.add_argument("--coefficient-rows", action="store_true",
                    help="keep four coefficients per knot and channel instead of the compact (value, second derivative) form")
    args = ap.parse_args()
    if args.paths <= 0:
        # ur6 / gen7: 8 paths in each of ~2048 wavefronts (2 per SIMD) is where the sweep kernel peaks, memory permitting
        args.paths = {"ur6": 16384, "gen7": 11264, "cspr": 2048}[args.workload]

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or os.environ.get("BATOTP_BENCH_FORCE_DIST") == "1"   # the latter: exercise RCCL on one GPU
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for N > 1"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    hip = capi.Context(capi.load_hip(), local_rank)  # raises if the HIP extension or the GPU is missing
    hip.set_sweep_group(args.group)
    hip.set_paths_per_wave(args.ppw)
    hip.set_overlap(args.overlap)

    # ---- synthetic inputs: K distinct host-resampled paths per GPU, tiled so that consecutive paths
    # (the 64/G paths that share a wavefront) are all different
    K = max(1, min(args.distinct, args.paths))
    seeds = [1000 + rank * K + k for k in range(K)]
    with cf.ThreadPoolExecutor(max_workers=min(K, os.cpu_count() or 1)) as ex:
        base = list(ex.map(lambda s: make_knots(args.workload, s, args.knots), seeds))
    prob = base[0][2]
    resample_info = None
    if rank == 0 and not args.no_resample:
        resample_info = measure_resampler(hip, args.workload, base, min(args.paths, 1024))
    if (prob.flags & capi.F_NO_SAMPLES) and not args.coefficient_rows:
        prob.flags |= capi.F_COMPACT_SPLINES  # same results, half the spline bytes per knot: room for more paths per GPU
    B = args.paths
    C = WORKLOADS[args.workload]["C"]
    while True:
        n_knots = [base[p % K][0].shape[1] for p in range(B)]
        total_knots = int(sum(n_knots))
        cap = int(max(n_knots) * {"ur6": 0.5, "gen7": 2.2, "cspr": 0.6}[args.workload]) + 1024
        try:
            batch = capi.Batch(hip, prob, n_knots, cap)
            break
        except capi.BatotpError as e:
            # the default sizes fill most of the 288 GB: on a GPU with less free memory run a smaller batch (reported in
            # config.paths_per_gpu) rather than nothing
            if "-5" not in str(e) and "Alloc" not in str(e) and "alloc" not in str(e) and "memory" not in str(e):
                raise
            if B <= 1024:
                raise
            B = max(1024, (B * 3 // 4) // 1024 * 1024)
            print(f"bench: batch did not fit, retrying with {B} paths", file=sys.stderr)
    for p in range(B):
        y, sres = base[p % K][0], base[p % K][1]
        batch.upload_knots(p, [y], [sres])
    hip.synchronize()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    gathered = None
    for _ in range(args.warmup):
        run_step(batch, False)
        gathered = bdist.gather_results(batch.results(), dev if use_dist else None)
    res = batch.results()
    bad = int(np.count_nonzero((res["status_rev"] | res["status_fwd"]) & ~np.uint32(capi.ST_BISECT_FAIL))) if args.warmup else 0
    if bad:
        raise RuntimeError(f"{bad} paths ended with an error status: raise the curve capacity")

    kernel_ms = {1: 0.0, 2: 0.0, 3: 0.0, 4: 0.0}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(batch, False)
        for k in kernel_ms:
            kernel_ms[k] += batch.kernel_ms(k)   # HIP events on the stream the kernels were launched on
        gathered = bdist.gather_results(batch.results(), dev if use_dist else None)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tk = torch.tensor([total_knots], dtype=torch.int64, device=dev)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        job_knots = int(tk.item())
    else:
        job_knots = total_knots
    for k in kernel_ms:
        kernel_ms[k] /= max(args.steps, 1)

    res = batch.results()
    steps_rev, steps_fwd = int(res["steps_rev"].sum()), int(res["steps_fwd"].sum())

    # ---- roofline of the dominant kernel (algorithmic bytes of SURVEY.md 8d, compact (y, M) figure):
    # a sweep reads the spline data of every knot once, 16*C bytes, and writes 16 bytes per integrated
    # point; the forward sweep also reads the reverse curve, 16 bytes per point
    bytes_rev = 16.0 * C * total_knots + 16.0 * (steps_rev + B)
    bytes_fwd = 16.0 * C * total_knots + 16.0 * (steps_fwd + B) + 16.0 * (steps_rev + B)
    # the dominant kernel is k_sweep; a step launches it twice (reverse, forward): per-launch averages, which is what
    # a rocprofv3 --stats summary of this command shows for the kernel (profiles/)
    dom_bytes = 0.5 * (bytes_rev + bytes_fwd)
    dom_ms = 0.5 * (kernel_ms[3] + kernel_ms[4])
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_sweep_traffic.json")
    if os.path.exists(pmc_path):
        try:
            traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "constraint-evaluated waypoints/sec + traversal-time err vs CPU ref",
        "value": job_knots * args.steps / elapsed,
        "unit": "waypoints/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": f"synthetic: {K} distinct seeded spline paths per GPU resampled by the host BA library, tiled to {B} paths",
        "config": {"workload": {"ur6": f"cfg2 UR5-like 6-DOF vel+acc, N~{args.knots} knots/path, batch of independent paths",
                                "gen7": f"cfg4 GEN7DOF 7-DOF vel+acc, N~{args.knots} knots/path, batch of independent paths",
                                "cspr": f"cfg5 CSPR3DOF cable tensions + vel/acc + Cartesian speed, N~{args.knots} knots/path"}[args.workload],
                   "paths_per_gpu": B, "knots_per_path_mean": total_knots / B, "channels": C, "lanes_per_path": args.group,
                   "regions": "K1+K2 precompute, K3 pointwise, K4 reverse+forward sweep", "parallelism": f"paths sharded x{world}"},
        "kernel_ms": {"precompute": kernel_ms[1], "pointwise_mvc": kernel_ms[2], "sweep_rev": kernel_ms[3], "sweep_fwd": kernel_ms[4]},
        "steps_per_knot": {"rev": steps_rev / total_knots, "fwd": steps_fwd / total_knots},
        # the three timed regions with their algorithmic bytes per waypoint (compact (y, M) figures):
        # R1 per-knot work (K1+K2+K3) 16*C + 24, R2 both sweeps 32*C + 16*(2*rho_rev + rho_fwd), R3 = R1 + R2
        "regions": (lambda r1_ms, r2_ms, rho_r, rho_f: {
            "R1_per_knot": {"ms": r1_ms, "waypoints_per_s": total_knots / (r1_ms * 1e-3), "bytes_per_waypoint": 16 * C + 24,
                            "algorithmic_GBps": total_knots * (16 * C + 24) / (r1_ms * 1e-3) / 1e9},
            "R2_sweeps": {"ms": r2_ms, "waypoints_per_s": total_knots / (r2_ms * 1e-3),
                          "bytes_per_waypoint": 32 * C + 16 * (2 * rho_r + rho_f),
                          "algorithmic_GBps": total_knots * (32 * C + 16 * (2 * rho_r + rho_f)) / (r2_ms * 1e-3) / 1e9},
            "R3_total": {"ms": r1_ms + r2_ms, "waypoints_per_s": total_knots / ((r1_ms + r2_ms) * 1e-3),
                         "bytes_per_waypoint": 48 * C + 24 + 16 * (2 * rho_r + rho_f)},
        })(kernel_ms[1] + kernel_ms[2], kernel_ms[3] + kernel_ms[4], steps_rev / total_knots, steps_fwd / total_knots),
        "stage_evals_per_s": 7.0 * (steps_rev + steps_fwd) / ((kernel_ms[3] + kernel_ms[4]) * 1e-3),
        "hbm_bytes_resident": batch.nbytes(),
        "gathered_rows": int(gathered.shape[0]) if gathered is not None else 0,
        "roofline": {"bound": "hbm", "kernel": "k_sweep (2 launches per step: reverse, forward; per-launch averages)", "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                     "reverse": {"ms": kernel_ms[3], "algorithmic_bytes": bytes_rev}, "forward": {"ms": kernel_ms[4], "algorithmic_bytes": bytes_fwd}},
    }

    # ---- SURVEY.md 8f-2 beside the hot path: the output stage (constant-time trajectories from the forward curves)
    # of the first paths of the batch on the device; untimed side measurement
    wcfg = WORKLOADS[args.workload]["cfg"]
    out_prm = capi.OutputParams(prob.n_joints, capi.PATH_JOINT if wcfg["path_type"] == "JOINT" else capi.PATH_CART, prob.integ_res, 0.008, 5.0)
    hip_out_theta0 = None
    covered = (wcfg["robot"] == "GENJNT" and not (prob.flags & capi.F_TRQ_ON)) or \
              (wcfg["robot"] == "CSPR3DOF" and (prob.flags & capi.F_TRQ_ON) and (prob.flags & capi.F_PARALLEL))
    if rank == 0 and not args.no_output and covered:
        n_out_paths = min(B, 512)
        best = None
        for _ in range(2):  # the second call finds the context's workspace allocated
            o = capi.Output(batch, out_prm, 0, n_out_paths)
            ms, pts = o.ms(), int(o.n_pts.sum())
            hip_out_theta0 = o.rows(0)
            o.close()
            best = ms if best is None else min(best, ms)
        out["output_stage"] = {"paths": n_out_paths, "points": pts, "ms": best, "points_per_s": pts / (best * 1e-3),
                               "out_res": out_prm.out_res, "out_smooth_fact": out_prm.out_smooth_fact,
                               "what": "s(t) spline + re-sampling at constant time steps + joint spline evaluation + smoothing / "
                                       "down-sampling (+ re-interpolation when out_res < integ_res) on the device"}

    # ---- optional loop form of the reverse sweep (batotp_hip_set_sweep_hold, off by default: DESIGN.md 4): the same
    # batch once more with it, results compared byte for byte with the timed runs'; untimed side measurement
    if rank == 0 and not args.no_flat_loop and not (prob.flags & (capi.F_TRQ_ON | capi.F_CART_VEL_ON | capi.F_CART_ACC_ON)):
        ref_rows = batch.results().tobytes()
        hip.set_sweep_group(8)
        hip.set_sweep_hold(4, -1)
        batch.sweep(-1)
        batch.sweep(+1)
        flat_rev, flat_fwd = batch.kernel_ms(3), batch.kernel_ms(4)
        same = batch.results().tobytes() == ref_rows
        hip.set_sweep_hold(-1, -1)
        hip.set_sweep_group(args.group)
        step_est = 1e3 * elapsed / max(args.steps, 1) - kernel_ms[3] - kernel_ms[4] + flat_rev + flat_fwd
        out["flat_reverse_loop"] = {"hold_reverse": 4, "sweep_rev_ms": flat_rev, "sweep_fwd_ms": flat_fwd,
                                    "ms_per_step_with_it": step_est, "waypoints_per_s_with_it": total_knots / (step_est * 1e-3),
                                    "result_rows_identical": bool(same),
                                    "what": "one loop for stages and bisection passes in the reverse sweep (paths of a wavefront drift apart); "
                                            "not the default, not part of value"}

    # ---- the same workload as ONE trajectory (BASELINE configs[1] wording): inherently sequential,
    # reported for transparency next to the batch figure
    if rank == 0:
        y1, sres1 = base[0][0], base[0][1]
        b1 = capi.Batch(hip, prob, [y1.shape[1]], cap)
        b1.upload_knots(0, [y1], [sres1])
        run_step(b1, False)
        t1 = time.perf_counter()
        run_step(b1, False)
        dt1 = time.perf_counter() - t1
        out["single_trajectory"] = {"knots": int(y1.shape[1]), "ms": 1e3 * dt1, "waypoints_per_s": y1.shape[1] / dt1,
                                    "kernel_ms": {"precompute": b1.kernel_ms(1), "pointwise_mvc": b1.kernel_ms(2),
                                                  "sweep_rev": b1.kernel_ms(3), "sweep_fwd": b1.kernel_ms(4)}}
        b1.close()

    # ---- CPU baseline: the oracle (bit-identical port of the reference's path) on the host cores, on a
    # bounded sample of the same workload, one path per thread
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        ora_lib = capi.load_oracle()
        cores = os.cpu_count() or 1
        octx = capi.Context(ora_lib, 0)

        def cpu_batch(n_paths, passes=1):
            """n_paths paths of the workload through the oracle, OpenMP: one path per host thread.
            The last of `passes` passes is timed (the first one also pays the page faults of fresh memory)."""
            nk = [base[i % K][0].shape[1] for i in range(n_paths)]
            b = capi.Batch(octx, prob, nk, cap)
            for i in range(n_paths):
                b.upload_knots(i, [base[i % K][0]], [base[i % K][1]])
            for _ in range(passes):
                t = time.perf_counter()
                b.precompute(0); b.pointwise_mvc(); b.sweep(-1); b.sweep(+1)
                dt = time.perf_counter() - t
            rr = b.results()
            th0 = None
            if hip_out_theta0 is not None:
                oo = capi.Output(b, out_prm, 0, 1)
                th0 = oo.rows(0)
                oo.close()
            b.close()
            return dt, sum(nk), rr, th0

        t_one, n_one, _, th0 = cpu_batch(1, passes=2)      # one path = one busy thread
        n_sample = int(max(cores, min(2 * cores, (args.cpu_seconds / max(t_one, 1e-3)) * cores)))
        n_sample = max(cores, (n_sample // cores) * cores)
        wall, n_wp, rows, _ = cpu_batch(n_sample, passes=2)
        if th0 is not None:
            out["output_stage"]["identical_to_oracle"] = bool(th0.tobytes() == hip_out_theta0.tobytes())
        out["cpu_baseline"] = {"value": n_wp / wall, "unit": "waypoints/s", "cores": cores, "kind": "port",
                               "single_thread_value": n_one / t_one,
                               "sample": f"{n_sample} paths of the same workload (N~{n_one}), OpenMP one path per thread on {cores} "
                                         f"host threads, same regions (K1-K4), oracle/ C restatement at -O2 -ffp-contract=off"}
        # traversal-time error vs the CPU reference on the sampled paths (T is quantised to integRes)
        m = min(n_sample, B)
        out["traversal_time_err_s"] = float(np.max(np.abs(res["t_total"][:m] - rows["t_total"][:m])))
        out["step_count_mismatches"] = int(np.count_nonzero(res["steps_fwd"][:m] != rows["steps_fwd"][:m]))

    batch.close()
    hip.trim()
    hip.close()
    if rank == 0:
        if resample_info is not None:
            out["resample"] = resample_info
        # BASELINE.json's target is worded for a 7-DOF robot at N = 100k: the same measurement on the GEN7DOF workload
        # (child process, after this one has released the GPU memory), reported beside the UR6 line
        if world == 1 and args.workload == "ur6" and not args.no_seven_dof:
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", "gen7", "--steps", "1", "--warmup", "1", "--no-cpu-baseline",
                   "--no-resample", "--no-output", "--knots", str(args.knots)]
            r = subprocess.run(cmd, capture_output=True, text=True)
            try:
                g = json.loads(r.stdout.strip().splitlines()[-1])
                out["seven_dof"] = {"value": g["value"], "unit": g["unit"], "ms_per_step": g["ms_per_step"], "config": g["config"],
                                    "kernel_ms": g["kernel_ms"], "steps_per_knot": g["steps_per_knot"], "roofline_frac": g["roofline"]["frac"],
                                    "flat_reverse_loop": g.get("flat_reverse_loop")}
            except Exception as e:  # the main line must not depend on the side measurement
                out["seven_dof"] = {"error": f"{type(e).__name__}: {r.stderr[-300:]}"}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
