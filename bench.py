#!/usr/bin/env python3
"""bench.py -- constraint-evaluated waypoints/sec of the batotp hot path on MI355X.

One "step" = one pass of the whole hot path over one batch of synthetic paths whose knot values are already resident
in HBM:
    per-knot precompute (spline coefficients, knot samples [, dynamics])      -> K1/K2
    per-knot max-admissible-sdot evaluation with bisection                    -> K3
    reverse sweep + forward sweep                                             -> K4
value = knots of all paths of all ranks / time (BASELINE.json: "constraint-evaluated waypoints/sec").

Configurations (--config):
    fill7 (default)  GEN7DOF 7-DOF vel+acc, N ~ 100k knots per path, as many paths as fill one GPU (weak scaling over
                     GPUs) -- BASELINE.json's target is worded for "a 7-DOF robot at N=100k" on "B-path batches";
                     the default line also carries the four BASELINE configs AS WORDED (`as_worded`: cfg2..cfg5) and the
                     side measurements of the stages either side of the path
    fill6            the same with the UR5-like 6-DOF workload of cfg 2 (round 1's headline)
    cfg2             UR5 6-DOF vel+acc, N = 100k, ONE trajectory (replicas only over GPUs)
    cfg3             KUKA-LWR-IV 7-DOF with torque limits, N = 100k, ONE trajectory (replicas only)
    cfg4             GEN7DOF, N = 50k, B = 1024 paths in total, sharded over the GPUs (strong scaling)
    cfg5             CSPR3DOF cable robot with tension limits, N = 200k, B = 4096 in total, sharded (strong scaling)
Multi-GPU: one process per GPU.  `python bench.py --gpus N` starts its own N ranks (torch.distributed.run as a child
process, before this process touches a GPU); started under torch.distributed.run it is a rank.  Paths are sharded over
ranks, no collective inside the hot path, one RCCL all_gather of the 64-byte result rows per step.

Prints ONE JSON line on rank 0.
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
# the PRODUCT's one-path route as a tool: BA::prepareKnots -> batotp_hip_resample with a batch of one (runs on the GPU)
BAKNOTS = os.path.join(ROOT, "batotp_amd", "host", "_build", "baknots")
# TEST INFRASTRUCTURE, cpu_baseline leg only: the same host shell linked against the CPU checker -- the ORACLE's resampler
# (oracle/batotp_oracle_resample.c), the independent restatement every CPU-side comparison of this file is fed from
ORACLE_KNOTS = os.path.join(ROOT, "oracle", "_build", "dump_knots")
METRIC = "constraint-evaluated waypoints/sec + traversal-time err vs CPU ref"
AS_WORDED_STEPS = 3     # timed steps of each BASELINE configuration as worded beside the headline (median reported as well)

WORKLOADS = {
    # name: fine-path generator, config kwargs, knots per coarse point, capacity of a curve in points per knot
    "ur6": dict(C=6, gen=lambda seed, n: (pathgen.ur_like_fine(seed, n), None, 0.01), knots_per_coarse=210.8, cap=0.5,
                cfg=dict(robot="GENJNT", is_parallel=0, n_joints=6, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                         degrees=1, jnt_vel=[160] * 6, jnt_acc_on=1, jnt_acc=[573, 573, 573, 1146, 1146, 1146], integ_res=0.008,
                         max_integ_time=2000000.0, theta_res=0.3, theta_res2=0.3)),
    "gen7": dict(C=7, gen=lambda seed, n: (pathgen.gen7dof_fine(seed, n), None, 0.01), knots_per_coarse=58.2, cap=2.2,
                 cfg=dict(robot="GENJNT", is_parallel=0, n_joints=7, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                          degrees=0, jnt_vel=[5] * 7, jnt_acc_on=1, jnt_acc=[10] * 7, integ_res=0.01, max_integ_time=2000000.0,
                          theta_res=0.1, theta_res2=0.1)),
    # BASELINE configs[2]: KUKA LWR IV+ 7-DOF, joint velocity / acceleration limits of the shipped example + the rated joint
    # torques with the chain dynamics of include/batotp_models.h (the reference has no model for this robot: DESIGN.md 5)
    "kuka7trq": dict(C=35, gen=lambda seed, n: (pathgen.kuka_like_fine(seed, n), None, 0.01), knots_per_coarse=235.0, cap=1.2,
                     cfg=dict(robot="KUKA", is_parallel=0, n_joints=7, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                              degrees=1, jnt_vel=[110, 110, 128, 128, 204, 184, 184], jnt_acc_on=1,
                              jnt_acc=[137.5, 157.1, 213.3, 213.3, 510.0, 460.0, 613.3], trq_on=1,
                              trq_max=[176, 176, 100, 100, 100, 38, 38], trq_min=[float("nan")] * 7, cart_vel_on=0, cart_vel=0.6,
                              integ_res=0.005, max_integ_time=2000000.0, theta_res=0.3, theta_res2=0.3)),
    # BASELINE configs[4]: cable robot, cable velocity/acceleration/tension limits + Cartesian speed, isPar2Ser=1
    "cspr": dict(C=18, gen=lambda seed, n: (None, pathgen.cspr_fine(seed, n), 0.005), knots_per_coarse=217.0, cap=0.8,
                 cfg=dict(robot="CSPR3DOF", is_parallel=1, n_joints=3, n_cart=3, traj_file="path.dat", is_bin=1, path_type="CART",
                          degrees=0, jnt_vel=[4] * 3, jnt_acc_on=1, jnt_acc=[8] * 3, trq_on=1, trq_max=[12] * 3, trq_min=[1] * 3,
                          cart_vel_on=1, cart_vel=4.0, cart_acc_on=0, cart_acc=100.0, integ_res=0.01, max_integ_time=2000000.0,
                          s_weights=(0, 0, 1), scale_type=2, theta_res=0.01, theta_res2=0.01, cart_res=0.01, cart_res2=0.01,
                          par2ser=1)),
}

# BASELINE.json configs as worded.  paths = total over all GPUs ("strong"), or per GPU ("weak": what fills one GPU, or the
# single trajectory every GPU replicates)
CONFIGS = {
    # lean: BATOTP_F_CURVES_IN_PLACE (one curve buffer per path, the forward curve overwrites the reverse curve as in the
    # reference's Traj) + BATOTP_F_MVC_IN_CURVES (the pointwise values live in the curve slots until the sweeps start); compact
    # batches keep no site array: 14.6 instead of 21.3 MB per path -> 16 384 instead of 11 264 paths, every lane of 2048
    # wavefronts carries a path
    # distinct = paths: EVERY path of the headline batch is its own seeded path.  (Up to round 4 the batch was 2048 distinct paths
    # tiled x8.  Since round 5 the library sweeps a ragged batch in the order of decreasing knot count -- which would put the
    # eight identical copies of a path into one wavefront, where they run in perfect lockstep: 3.05e8 instead of 2.92e8
    # waypoints/s on the tiled batch, an artefact of the tiling and not a property of the kernels.)
    "fill7": dict(workload="gen7", knots=100000, paths=16384, scaling="weak", distinct=16384, lean=True,
                  what="GEN7DOF 7-DOF vel+acc, N~100k knots/path, batch of independent paths filling the GPU"),
    "fill6": dict(workload="ur6", knots=100000, paths=16384, scaling="weak", distinct=2048,
                  what="cfg2 shape: UR5-like 6-DOF vel+acc, N~100k knots/path, batch of independent paths filling the GPU"),
    "cfg2": dict(workload="ur6", knots=100000, paths=1, scaling="weak", distinct=1,
                 what="cfg2 as worded: UR5 6-DOF vel+acc, N=100k, single trajectory per GPU (replicas only)"),
    "cfg3": dict(workload="kuka7trq", knots=100000, paths=1, scaling="weak", distinct=1,
                 what="cfg3 as worded: KUKA-LWR-IV 7-DOF with torque limits (chain dynamics), N=100k, single trajectory per GPU "
                      "(replicas only)"),
    "cfg4": dict(workload="gen7", knots=50000, paths=1024, scaling="strong", distinct=1024,
                 what="cfg4 as worded: GEN7DOF, N=50k, batch of 1024 randomised paths sharded across the GPUs"),
    # 128 distinct paths tiled x32, swept in the order given (a tiled batch: measure()).  Round 5 also ran it with 2048 distinct
    # seeds: random cable-robot paths have a heavy tail of barely feasible ones -- the slowest of 2048 takes 1.9x the median's
    # integration steps at more than twice the time per step (long bisections) -- and a launch lasts as long
    # as its slowest path: 16.7 s instead of 6.3 s for the same 4096 x 2e5 knots.  The sample of round 4 is kept so
    # that the number stays comparable; candidates that cannot finish are swapped as before (`swapped_seeds`).
    # lean (round 5): one curve buffer per path and the pointwise values in it -- 62.7 instead of 67.6 MB per path, so that one GPU
    # holds its whole share (4096 paths) as ONE resident batch, which the library then sweeps with two paths per wavefront
    "cfg5": dict(workload="cspr", knots=200000, paths=4096, scaling="strong", distinct=128, lean=True,
                 what="cfg5 as worded: CSPR3DOF cable robot with cable-tension constraints, N=200k, batch of 4096 sharded "
                      "across the GPUs"),
    # the same configuration WITHOUT a curated sample: the first 2048 seeds as they come (4096 paths = those x2), only candidates that
    # cannot finish at all are swapped and counted.  Random cable-robot paths have a heavy tail of barely feasible ones, and a launch
    # lasts as long as its slowest path: this is the number a user with arbitrary paths sees (round 5: 16.7 s against 3.3 s)
    "cfg5_distinct2048": dict(workload="cspr", knots=200000, paths=4096, scaling="strong", distinct=2048, lean=True, steps=1,
                              what="cfg5 as worded with 2048 distinct seeds as they come (no curated sample): CSPR3DOF cable robot with "
                                   "cable-tension constraints, N=200k, batch of 4096 sharded across the GPUs"),
}


# ---------------------------------------------------------------------------------------------------------------------
# synthetic inputs
# ---------------------------------------------------------------------------------------------------------------------
def make_knots(workload: str, seed: int, n_target: int, tool: str = BAKNOTS):
    """one synthetic path -> (y [C_in][N], sres, problem, taught points) through BA::prepareKnots of the host library: with the
    default tool the product's one-path route (the device resampler with a batch of one), with ORACLE_KNOTS the checker's"""
    w = WORKLOADS[workload]
    n_coarse = max(8, int(round(n_target / w["knots_per_coarse"])))
    theta, cart, tres = w["gen"](seed, n_coarse)
    with tempfile.TemporaryDirectory() as work:
        pathgen.write_traj_bin(os.path.join(work, "path.dat"), tres, theta, cart)
        pathgen.write_config(os.path.join(work, "config.dat"), **w["cfg"])
        r = subprocess.run([tool, "config.dat"], cwd=work, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{os.path.basename(tool)} failed: " + r.stdout[-1000:])
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
    r.robot_type = {"GENJNT": capi.ROBOT_GENJNT, "CSPR3DOF": capi.ROBOT_CSPR3DOF, "KUKA": capi.ROBOT_KUKA}[cfg["robot"]]
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


def device_resamplable(workload: str) -> bool:
    """path kinds batotp_hip_resample covers (JOINT paths of GENJNT, CART paths of the cable robot)"""
    return WORKLOADS[workload]["cfg"]["robot"] in ("GENJNT", "CSPR3DOF")


def taught_points_f32(workload: str, seeds, n_target: int):
    """the taught points of one path per seed as the trajectory file stores them (float32): list of (theta or None, cart or
    None), and the file's resolution"""
    w = WORKLOADS[workload]
    n_coarse = max(8, int(round(n_target / w["knots_per_coarse"])))
    with cf.ThreadPoolExecutor(max_workers=min(len(seeds), os.cpu_count() or 1, 32)) as ex:
        got = list(ex.map(lambda s: w["gen"](s, n_coarse), seeds))
    return [(g[0], g[1]) for g in got], float(np.float32(got[0][2]))


def widen(workload: str, taught):
    """[nJ + nC][n] float64 rows the device resampler takes (theta rows, then Cartesian rows; absent ones zero)"""
    cfg = WORKLOADS[workload]["cfg"]
    nJ, nC = cfg["n_joints"], cfg["n_cart"]
    theta, cart = taught
    n = (theta if theta is not None else cart).shape[1]
    x = np.zeros((nJ + nC, n))
    if theta is not None:
        x[:nJ] = theta.astype(np.float32).astype(np.float64)
    if cart is not None:
        x[nJ:] = cart.astype(np.float32).astype(np.float64)
    return x


class Inputs:
    """K distinct synthetic paths of a workload for one rank: problem description, knot counts / spacing per distinct path
    and a way to put the knots of distinct path k into path p of a batch"""

    # distinct paths per call of the device resampler (the library cuts a call into chunks that fit its scratch budget itself; what bounds
    # this number is the host memory of the widened taught points -- gen7: 2.7 MB per path -- and the knots a call leaves resident:
    # 8 MB per path).  Round 5 used 256: 129 calls for the headline's two passes, each with a 61 ms floor (one walk wavefront per path)
    CHUNK = 1024

    def __init__(self, hip, workload, knots, seeds):
        self.hip, self.workload, self.K = hip, workload, len(seeds)
        self.seeds, self.knots_target = list(seeds), knots
        first = make_knots(workload, seeds[0], knots)        # the one-path route of the product: problem description, and the check below
        self.prob = first[2]
        self.on_device = device_resamplable(workload) and self.K > 4
        if self.on_device:
            # the device resampler produces the knots, chunk by chunk; here only sizes are kept (and the
            # float32 taught points, so that the second pass does not regenerate them)
            self.prm = resample_params(workload, self.prob)
            self.nC_in = self.prm.n_joints + self.prm.n_cart
            self.keep = first[0].shape[0]                   # channels the batch carries (Cartesian rows dropped when unused)
            self.n_knots, self.sres = np.zeros(self.K, np.int64), np.zeros(self.K)
            self.skipped = 0
            want = self.K
            if self.prob.flags & capi.F_PARALLEL:
                # random cable-robot paths can ask for cable tensions outside the limits even at rest: the sweep
                # of such a path crawls at the speed floor with a failing 100-iteration bisection at every stage until it runs
                # out of capacity (the reference grinds through it the same way and then returns -1).  The benchmark measures
                # feasible paths: candidates with a knot at which K3 finds no admissible sdot are skipped.
                self.seeds = self.seeds + [self.seeds[-1] + 1 + k for k in range(max(32, self.K // 16))]
                self.K = len(self.seeds)
            self.n_knots, self.sres = np.zeros(self.K, np.int64), np.zeros(self.K)
            self.taught, self.sres_in = taught_points_f32(workload, self.seeds, knots)
            self.one_path_check = None
            feasible = np.ones(self.K, bool)
            for k0, rs in self._chunks():
                m = rs.n_knots.shape[0]
                if np.any(rs.status):
                    raise RuntimeError(f"device resampler refused {int(np.count_nonzero(rs.status))} synthetic paths")
                self.n_knots[k0:k0 + m], self.sres[k0:k0 + m] = rs.n_knots, rs.sres
                if k0 == 0:
                    # the batch call and the one-path call (BA::interpInputData, a batch of one in another process) agree
                    self.one_path_check = bool(np.array_equal(rs.knots(0)[: self.keep], first[0]) and rs.sres[0] == first[1])
                if self.K > want:
                    # cable robot: which candidates have an admissible speed at every knot -- the per-knot evaluation (K3) of this
                    # chunk of candidates, through a small batch of its own (every channel as pairs: 288 B per knot)
                    pr = capi.Problem.from_buffer_copy(bytes(self.prob))
                    if (pr.flags & capi.F_PARALLEL) and (pr.flags & capi.F_PAR2SER):
                        pr.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
                    tb = capi.Batch(hip, pr, [int(n) for n in rs.n_knots], 8)
                    tb.upload_knots_device(0, m, rs.device_ptr(), list(rs.sres))
                    tb.precompute(0)
                    tb.pointwise_mvc()
                    for j in range(m):
                        feasible[k0 + j] = not np.isnan(tb.mvc(j)[1]).any()
                    tb.close()
                rs.close()
            hip.trim()
            if self.K > want:
                ok = [p for p in range(self.K) if feasible[p]]
                self.skipped = self.K - len(ok)
                if len(ok) < want:
                    raise RuntimeError(f"only {len(ok)} of {self.K} candidate paths are feasible")
                self.spares = [(self.seeds[p], self.taught[p], int(self.n_knots[p]), float(self.sres[p])) for p in ok[want:]]
                ok = ok[:want]
                self.seeds = [self.seeds[p] for p in ok]
                self.taught = [self.taught[p] for p in ok]
                self.n_knots, self.sres, self.K = self.n_knots[ok], self.sres[ok], want
            self._data = f"synthetic: {self.K} distinct seeded spline paths per GPU (taught points -> knots by the device resampler)"
        else:
            with cf.ThreadPoolExecutor(max_workers=min(self.K, os.cpu_count() or 1)) as ex:
                rest = list(ex.map(lambda s: make_knots(workload, s, knots), seeds[1:]))
            self.base = [first] + rest
            self.n_knots = np.array([b[0].shape[1] for b in self.base], np.int64)
            self.sres = np.array([b[1] for b in self.base])
            self._data = (f"synthetic: {self.K} distinct seeded spline paths per GPU resampled one by one through BA::interpInputData "
                          f"(the device resampler with a batch of one)")

    spares = []
    skipped = 0

    @property
    def data(self):
        return self._data + (f"; {self.skipped} candidate paths the cable-tension limits do not admit were replaced by other seeds" if self.skipped else "")

    def replace(self, distinct):
        """swap distinct paths whose sweep ended with an error (the constraints do not admit them) for spare candidates;
        returns how many could be replaced"""
        done = 0
        for k in distinct:
            if not self.on_device or not self.spares:
                break
            seed, taught, n, sres = self.spares.pop(0)
            self.seeds[k], self.taught[k], self.n_knots[k], self.sres[k] = seed, taught, n, sres
            self.skipped += 1
            done += 1
        return done

    def _chunks(self):
        for k0 in range(0, self.K, self.CHUNK):
            xs = [widen(self.workload, t) for t in self.taught[k0:k0 + self.CHUNK]]
            yield k0, capi.Resampled(self.hip, self.prm, xs, [self.sres_in] * len(xs))

    def oracle_knots(self, k):
        """CHECKER (cpu_baseline leg and tests only): (y, sres) of distinct path k from the ORACLE's resampler -- independent
        of the device resampler that produced the knots the batch holds"""
        return make_knots(self.workload, self.seeds[k], self.knots_target, tool=ORACLE_KNOTS)[:2]

    def device_knot_digests(self, D):
        """(sha256 of the knot rows, N, sres) of the first D distinct paths as the PRODUCT made them (device resampler: the batch
        call, or the one-path route) -- kept on the host so that the cpu_baseline leg can compare them with the oracle's knots
        after the GPU context is gone"""
        import hashlib
        out = []
        if not self.on_device:
            for k in range(min(D, self.K)):
                out.append((hashlib.sha256(np.ascontiguousarray(self.base[k][0]).tobytes()).hexdigest(), int(self.base[k][0].shape[1]), float(self.base[k][1])))
            return out
        xs = [widen(self.workload, t) for t in self.taught[: min(D, self.K)]]
        rs = capi.Resampled(self.hip, self.prm, xs, [self.sres_in] * len(xs))
        for k in range(len(xs)):
            y = np.ascontiguousarray(rs.knots(k)[: self.keep])
            out.append((hashlib.sha256(y.tobytes()).hexdigest(), int(y.shape[1]), float(rs.sres[k])))
        rs.close()
        return out

    def fill(self, batch, n_paths):
        """path p of the batch <- distinct path p % K"""
        K = self.K
        if not self.on_device:
            for p in range(n_paths):
                batch.upload_knots(p, [self.base[p % K][0]], [self.base[p % K][1]])
            return
        for k0, rs in self._chunks():
            m = rs.n_knots.shape[0]
            ptr = rs.device_ptr()
            off = np.concatenate([[0], np.cumsum(rs.n_knots)]) * self.nC_in
            if n_paths <= K:
                hi = min(k0 + m, n_paths)     # consecutive paths are contiguous in the resampler's output: ONE call for the block; the batch
                if hi > k0:                   # keeps the first `keep` of the nC_in rows of a path (batotp_hip_upload_knots_device_rows)
                    batch.upload_knots_device_rows(k0, hi - k0, ptr, self.nC_in, list(rs.sres[: hi - k0]))
            else:
                for j in range(m):
                    # the first `keep` rows of a path are the first keep * N doubles of its block
                    for p in range(k0 + j, n_paths, K):
                        batch.upload_knots_device(p, 1, ptr + 8 * int(off[j]), [float(rs.sres[j])])
            rs.close()
        self.hip.synchronize()


# ---------------------------------------------------------------------------------------------------------------------
# one measurement
# ---------------------------------------------------------------------------------------------------------------------
def run_step(batch):
    batch.precompute(0)
    batch.pointwise_mvc()
    batch.sweep(-1)
    batch.sweep(+1)


def prepare_dynamics(batch, prob, n_paths):
    """serial robots with a chain model (cfg 3): set the table and upload the host cosines / sines of the joint angles
    (the trig policy of DESIGN.md 2); part of the input preparation, outside the timed region like every upload"""
    if not (prob.flags & capi.F_TRQ_ON) or (prob.flags & capi.F_PARALLEL) or prob.robot_type == capi.ROBOT_RR:
        return None
    import math
    model = batch.L.builtin_serial_model(prob.robot_type)
    batch.set_serial_model(model)
    batch.precompute(1)
    unit = (3.14159265358979323846 / 180.0) if model.degrees else 1.0
    nJ = prob.n_joints
    for p in range(n_paths):
        trig = np.empty((2 * nJ, int(batch.n_knots[p])))
        for j in range(nJ):
            q = unit * batch.samples(p, j)[0]
            trig[j] = [math.cos(v) for v in q]       # libm's cos / sin (numpy's vectorised ones are not bit-identical)
            trig[nJ + j] = [math.sin(v) for v in q]
        batch.upload_joint_trig(p, trig)
    return model


def bytes_per_path(prob, C, n_mean, cap):
    """HBM bytes a path of n_mean knots occupies in a batch (batotp_hip_batch_create's arrays)"""
    cin, d = prob.n_joints + prob.n_cart, prob.dyn_dim
    mvc = 0 if (prob.flags & capi.F_MVC_IN_CURVES) else 24
    if prob.flags & capi.F_COMPACT_SPLINES:
        pairs_all = d or (prob.flags & (capi.F_CART_VEL_ON | capi.F_CART_ACC_ON))   # every channel as a pair, nothing else per knot
        per_knot = 16.0 * (C if pairs_all else cin) + mvc        # no site array for compact batches
    else:
        per_knot = 8.0 * cin + 8 + 32.0 * C + (0 if (prob.flags & capi.F_NO_SAMPLES) else 24.0 * cin) + 32.0 * d + mvc + 8.0 * max(cin, 4 * d)
        if d and not (prob.flags & capi.F_PARALLEL):
            per_knot += 16.0 * prob.n_joints      # joint trig tables of the chain model
    return per_knot * n_mean + (16.0 if prob.flags & capi.F_CURVES_IN_PLACE else 32.0) * cap


def plan_chunks(B, K, limit):
    """a rank's share of B paths in chunks of at most `limit` paths (what fits its HBM): equal chunks, whole multiples of the K
    distinct paths when there is more than one chunk (every chunk then holds the same distinct paths and runs through the same
    resident batch)"""
    n = 1
    while (B + n - 1) // n > limit:
        n += 1
    bc = (B + n - 1) // n
    if n > 1 and K <= limit:
        bc = max(K, (bc // K) * K)      # (more distinct paths than fit one chunk: every chunk holds the first bc of them)
    return [min(bc, B - i * bc) for i in range((B + bc - 1) // bc)] if B else []


def measure(hip, cfg_name, rank, world, steps, warmup, dist_ctx, paths_override=0, knots_override=0, group=0, ppw=0,
            coefficient_rows=False, distinct_override=0, keep=False, lean=False):
    """run one configuration on this rank's share; returns (result dict, kept objects or None)"""
    import torch
    c = CONFIGS[cfg_name]
    workload = c["workload"]
    knots = knots_override or c["knots"]
    total_paths = paths_override or c["paths"]
    if c["scaling"] == "strong":
        lo, hi = bdist.shard_range(total_paths, rank, world)
        B = hi - lo
    else:
        lo, B = 0, total_paths
    K = max(1, min(distinct_override or c["distinct"], max(B, 1)))
    seeds = [1000 + (rank * 100003 + lo) + k for k in range(K)]
    hip.set_sweep_group(group)
    hip.set_paths_per_wave(ppw)
    # a TILED batch (fewer distinct paths than paths: profiling passes, --distinct) of a velocity / acceleration-only problem is
    # swept in the order given: sorted by knot count, the identical copies of a path would become neighbours in ONE wavefront of
    # the 8-lane kernels and run in lockstep, which no real batch does.  (Problems with torque or Cartesian limits run one path
    # per wavefront: there the order only decides which paths share a SIMD, whether they are copies of each other or not.)
    wcfg0 = WORKLOADS[workload]["cfg"]
    multi_path_wavefronts = not (wcfg0.get("trq_on") or wcfg0.get("cart_vel_on") or wcfg0.get("cart_acc_on"))
    hip.set_path_order(0 if (B > K and multi_path_wavefronts) else 1)
    inp = Inputs(hip, workload, knots, seeds)
    prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
    if (prob.flags & capi.F_NO_SAMPLES) and not coefficient_rows:
        prob.flags |= capi.F_COMPACT_SPLINES  # same results, half the spline bytes per knot: room for more paths per GPU
    elif (prob.flags & capi.F_PARALLEL) and (prob.flags & capi.F_PAR2SER) and not coefficient_rows:
        # the cable robot in serial form: every channel (cables, platform position, a1..a4 of every row) as (value, second
        # derivative) pairs -- same results, 288 instead of 992 bytes per knot: twice the paths per chunk, two wavefronts per SIMD
        prob.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
    if c.get("lean") or lean:
        prob.flags |= capi.F_CURVES_IN_PLACE | capi.F_MVC_IN_CURVES   # same results, one curve buffer per path and nothing else per knot
    C = WORKLOADS[workload]["C"]
    cap = int(int(inp.n_knots.max()) * WORKLOADS[workload]["cap"]) + 1024
    if prob.flags & capi.F_MVC_IN_CURVES:
        cap = max(cap, int(1.5 * int(inp.n_knots.max())) + 64)     # the three pointwise values of every knot live in the curve slot

    # A batch must fit the free HBM.  A larger share is processed in chunks of whole multiples of the K distinct paths, every
    # chunk through the same resident device batch (chunk i holds the same K distinct paths as chunk 0, so re-running the
    # batch IS processing the next chunk: its inputs alias the same HBM).
    free_b, _ = torch.cuda.mem_get_info()
    fit = max(1, int(0.93 * (free_b - (8 << 30)) / bytes_per_path(prob, C, float(inp.n_knots.mean()), cap)))

    split = lambda limit: plan_chunks(B, K, limit)
    chunk_sizes = split(fit)
    batch = None
    while B and batch is None:
        try:
            batch = capi.Batch(hip, prob, [int(inp.n_knots[p % K]) for p in range(chunk_sizes[0])], cap)
        except capi.BatotpError as e:
            if "-5" not in str(e) or chunk_sizes[0] <= K:
                raise
            chunk_sizes = split(max(K, chunk_sizes[0] * 3 // 4))
            print(f"bench: batch did not fit, retrying with chunks of {chunk_sizes[0]} paths", file=sys.stderr)
    if B:
        inp.fill(batch, chunk_sizes[0])
        prepare_dynamics(batch, prob, chunk_sizes[0])
        hip.synchronize()
    knots_of = lambda s: int(sum(int(inp.n_knots[p % K]) for p in range(s)))

    def barrier():
        if dist_ctx is not None:
            dist_ctx["dist"].barrier()
        torch.cuda.synchronize()

    dev = dist_ctx["dev"] if dist_ctx is not None else None
    kernel_ms = {1: 0.0, 2: 0.0, 3: 0.0, 4: 0.0}
    state = {"gathered": None, "rows": None}

    def one_pass(timed, sizes):
        nonlocal batch
        rows = []
        for s in sizes:
            run_step(batch)
            if timed:
                for k in kernel_ms:
                    kernel_ms[k] += batch.kernel_ms(k)   # HIP events on the stream the kernels were launched on
            rows.append(batch.results()[:s])
        local = np.concatenate(rows) if rows else np.zeros(0, dtype=capi.RESULT_DTYPE)
        if timed:   # (warm-up passes stay free of collectives: ranks may repeat them independently of each other)
            state["gathered"] = bdist.gather_results(local, dev) if dist_ctx is not None else local
        state["rows"] = local

    def capacity_errors():
        rows0 = state["rows"]
        if rows0 is None or not rows0.shape[0]:
            return 0, 0.0
        bad = (rows0["status_rev"] | rows0["status_fwd"]) & ~np.uint32(capi.ST_BISECT_FAIL)
        worst = float(np.max(np.maximum(rows0["steps_rev"], rows0["steps_fwd"]) / inp.n_knots[np.arange(rows0.shape[0]) % K]))
        return int(np.count_nonzero(bad)), worst

    failed_paths = 0
    for attempt in range(3):
        for _ in range(max(warmup, 1) if attempt else warmup):
            one_pass(False, chunk_sizes[:1])       # warm-up: the first chunk
        bad, worst = capacity_errors() if (warmup or attempt) else (0, 0.0)
        if dist_ctx is not None:
            # every rank takes the same decision (the retry re-creates the batch; the timed passes that follow are collective)
            flag = torch.tensor([bad], dtype=torch.int64, device=dev)
            dist_ctx["dist"].all_reduce(flag, op=dist_ctx["dist"].ReduceOp.MAX)
            bad = int(flag.item())
        failed_paths = bad
        if not bad or attempt == 2:
            break
        rows0 = state["rows"]
        all_replaced = False
        if rows0 is not None and rows0.shape[0] and inp.spares:
            # paths the constraints do not admit (they crawl at the speed floor until the capacity is exhausted): other seeds take their place
            st0 = (rows0["status_rev"] | rows0["status_fwd"]) & ~np.uint32(capi.ST_BISECT_FAIL)
            offenders = sorted(set(int(p) % K for p in np.nonzero(st0)[0]))
            all_replaced = inp.replace(offenders) == len(offenders) and worst > 0.9 * cap / max(float(inp.n_knots.max()), 1.0)
        # distinct random paths differ in the integration steps they need per knot: the first retry also gives the curves
        # twice the room -- unless every offender ran into the capacity (a crawler) and was replaced by another seed; what still
        # fails after the retries is reported as failed (the reference would return -1 for it)
        if attempt == 0 and not all_replaced:
            cap = int(cap * 2)
        print(f"bench: {bad} paths ended with an error status (up to {worst:.2f} steps per knot), retrying with {cap} points per curve", file=sys.stderr)
        if batch is not None:
            batch.close()
            cap = max(cap, int(int(inp.n_knots.max()) * WORKLOADS[workload]["cap"]) + 1024)
            batch = capi.Batch(hip, prob, [int(inp.n_knots[p % K]) for p in range(chunk_sizes[0])], cap)
            inp.fill(batch, chunk_sizes[0])
            prepare_dynamics(batch, prob, chunk_sizes[0])
            hip.synchronize()
    local_knots = sum(knots_of(s) for s in chunk_sizes)     # (after the retries: replaced paths have other knot counts)
    barrier()
    t0 = time.perf_counter()
    step_s = []
    for _ in range(steps):
        ts = time.perf_counter()
        one_pass(True, chunk_sizes)      # ends with the result rows on the host (and the all_gather): a complete step
        step_s.append(time.perf_counter() - ts)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_ctx is not None:
        dist = dist_ctx["dist"]
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tk = torch.tensor([local_knots, B], dtype=torch.int64, device=dev)
        dist.all_reduce(tk, op=dist.ReduceOp.SUM)
        job_knots, job_paths = int(tk[0].item()), int(tk[1].item())
    else:
        job_knots, job_paths = local_knots, B
    for k in kernel_ms:
        kernel_ms[k] /= max(steps, 1)

    # the variable-length part of the gather: the forward curves of every path to rank 0, device to device
    # (size exchange + grouped send/recv); untimed side measurement of the sharded configurations
    curve_gather = None
    if dist_ctx is not None and c["scaling"] == "strong":     # (collective: a rank with an empty share takes part with batch = None)
        barrier()
        tg = time.perf_counter()
        got = bdist.gather_curves(batch, +1, dev, on_device=True)
        barrier()
        tg = time.perf_counter() - tg
        if rank == 0:
            pts = int(sum(int(t.shape[0]) for t in got[0]))
            curve_gather = {"ms": 1e3 * tg, "points": pts, "GB": 16e-9 * pts, "GBps": 16e-9 * pts / tg,
                            "what": "forward curves (s, sdot) of all paths of all ranks to rank 0: all_gather of the point counts + one "
                                    "grouped send/recv of packed double2 buffers"}
        got = None

    res = state["rows"] if state["rows"] is not None else np.zeros(0, dtype=capi.RESULT_DTYPE)
    steps_rev, steps_fwd = int(res["steps_rev"].sum()), int(res["steps_fwd"].sum())
    sr, sf = res["steps_rev"].astype(np.float64), res["steps_fwd"].astype(np.float64)

    # roofline of the dominant kernel (algorithmic bytes of SURVEY.md 8d, compact (y, M) figure): a sweep reads the spline
    # data of every knot once, 16*C bytes, and writes 16 bytes per integrated point; the forward sweep also reads the
    # reverse curve, 16 bytes per point.  k_sweep is launched twice per step and chunk: per-launch averages, which is what a
    # rocprofv3 --stats summary of this command shows for the kernel (profiles/)
    launches = max(len(chunk_sizes), 1)
    bytes_rev = 16.0 * C * local_knots + 16.0 * (steps_rev + B)
    bytes_fwd = 16.0 * C * local_knots + 16.0 * (steps_fwd + B) + 16.0 * (steps_rev + B)
    dom_bytes = 0.5 * (bytes_rev + bytes_fwd) / launches
    dom_ms = 0.5 * (kernel_ms[3] + kernel_ms[4]) / launches
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    layout = "compact" if (prob.flags & capi.F_COMPACT_SPLINES) else "rows"
    traffic, traffic_src, issue_rec = recorded_traffic(workload, chunk_sizes[0] if B else 0, layout)
    issue = None
    if issue_rec:
        # the issue side of the same kernel from the SQ counters of the matching recorded run: share of all SIMD cycles with a
        # VALU instruction in flight x share of the 64 lanes that carry work = fraction of the fp64 vector pipes' lane-cycles used
        vb, al = issue_rec["valu_busy_frac"], issue_rec["active_lanes_of_64"]
        issue = {"valu_busy_frac": 0.5 * (vb["reverse"] + vb["forward"]), "active_lanes_of_64": 0.5 * (al["reverse"] + al["forward"]),
                 "frac": 0.5 * (vb["reverse"] * al["reverse"] + vb["forward"] * al["forward"]) / 64.0,
                 "by_direction": {"reverse": vb["reverse"] * al["reverse"] / 64.0, "forward": vb["forward"] * al["forward"] / 64.0},
                 "source": issue_rec.get("source")}
    tk_ = max(local_knots, 1)
    r1_ms, r2_ms = max(kernel_ms[1] + kernel_ms[2], 1e-9), max(kernel_ms[3] + kernel_ms[4], 1e-9)
    rho_r, rho_f = steps_rev / tk_, steps_fwd / tk_
    out = {
        "metric": METRIC,
        "value": job_knots * steps / elapsed,
        "unit": "waypoints/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": warmup,
        "ms_per_step": 1e3 * elapsed / max(steps, 1),
        "ms_per_step_median": 1e3 * float(np.median(step_s)) if step_s else None,   # this rank's steps, one by one
        "ms_per_step_min_max": [1e3 * min(step_s), 1e3 * max(step_s)] if step_s else None,
        "higher_is_better": True,
        "scaling": c["scaling"],
        "vs_baseline": None,
        "dtype": "f64",
        "data": inp.data + ((f", tiled to {B} paths" + (" (swept in the order given)" if multi_path_wavefronts else "")) if B > K else ""),
        "config": {"workload": c["what"], "config": cfg_name, "paths_total": job_paths, "paths_per_gpu": B,
                   "chunks_per_step": len(chunk_sizes), "knots_per_path_mean": local_knots / max(B, 1), "channels": C,
                   "spline_layout": layout, "lanes_per_path": group, "distinct_paths_per_gpu": K,
                   "regions": "K1+K2 precompute, K3 pointwise, K4 reverse+forward sweep", "parallelism": f"paths sharded x{world}"},
        "kernel_ms": {"precompute": kernel_ms[1], "pointwise_mvc": kernel_ms[2], "sweep_rev": kernel_ms[3], "sweep_fwd": kernel_ms[4]},
        "steps_per_knot": {"rev": rho_r, "fwd": rho_f},
        "steps_per_path": {"rev_min_median_max": [float(np.min(sr)), float(np.median(sr)), float(np.max(sr))] if B else None,
                           "fwd_min_median_max": [float(np.min(sf)), float(np.median(sf)), float(np.max(sf))] if B else None},
        # the three timed regions with their algorithmic bytes per waypoint (compact (y, M) figures):
        # R1 per-knot work (K1+K2+K3) 16*C + 24, R2 both sweeps 32*C + 16*(2*rho_rev + rho_fwd), R3 = R1 + R2
        "regions": {
            "R1_per_knot": {"ms": r1_ms, "waypoints_per_s": local_knots / (r1_ms * 1e-3), "bytes_per_waypoint": 16 * C + 24,
                            "algorithmic_GBps": local_knots * (16 * C + 24) / (r1_ms * 1e-3) / 1e9},
            "R2_sweeps": {"ms": r2_ms, "waypoints_per_s": local_knots / (r2_ms * 1e-3), "bytes_per_waypoint": 32 * C + 16 * (2 * rho_r + rho_f),
                          "algorithmic_GBps": local_knots * (32 * C + 16 * (2 * rho_r + rho_f)) / (r2_ms * 1e-3) / 1e9},
            "R3_total": {"ms": r1_ms + r2_ms, "waypoints_per_s": local_knots / ((r1_ms + r2_ms) * 1e-3),
                         "bytes_per_waypoint": 48 * C + 24 + 16 * (2 * rho_r + rho_f)}},
        "stage_evals_per_s": 7.0 * (steps_rev + steps_fwd) / (r2_ms * 1e-3),
        "us_per_integration_step": (1e3 * (kernel_ms[3] + kernel_ms[4]) / max(steps_rev + steps_fwd, 1)) if B == 1 else None,
        "paths_with_error_status": int(np.count_nonzero((res["status_rev"] | res["status_fwd"]) & ~np.uint32(capi.ST_BISECT_FAIL))) if B else 0,
        # candidate paths the constraints do not admit (cable tensions outside their limits) that were replaced by other seeds
        # before the timed region -- the reference would grind through such a path and return -1
        "swapped_seeds": int(inp.skipped),
        # a launch of one-path wavefronts lasts as long as its slowest path: integration steps of the slowest / the mean path
        "slowest_over_mean_path": {"rev": float(np.max(sr) / max(np.mean(sr), 1.0)), "fwd": float(np.max(sf) / max(np.mean(sf), 1.0))} if B else None,
        "hbm_bytes_resident": batch.nbytes() if batch is not None else 0,
        "gathered_rows": int(state["gathered"].shape[0]) if state["gathered"] is not None else 0,
        # "bound" names the roofline the path is priced against (north_star: HBM GB/s vs peak); what the counters say limits the
        # kernel is in "limited_by" / "issue": the sweep is an initial-value problem, bound by fp64 instruction issue on two
        # wavefronts per SIMD, not by bytes (SURVEY.md 8d "honest expectation")
        "roofline": {"bound": "valu-issue" if B > 64 else "latency (one wavefront per path)", "priced_against": "hbm",
                     "limited_by": ("fp64 vector instruction issue on two wavefronts per SIMD" if B > 64 else
                                    "instruction issue of a lone wavefront per SIMD") + " (the counters in `issue`; HBM bytes are what the north_star prices)",
                     "issue": issue,
                     "kernel": f"k_sweep ({2 * launches} launches per step: reverse, forward; per-launch averages)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src, "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": dom_ms,
                     "reverse": {"ms": kernel_ms[3], "algorithmic_bytes": bytes_rev}, "forward": {"ms": kernel_ms[4], "algorithmic_bytes": bytes_fwd}},
    }
    if inp.on_device:
        # product against product (two routes into the same kernels); the comparison with the ORACLE's resampler is part of the
        # cpu_baseline leg (`inputs_identical_to_oracle_resampler`)
        out["batch_resampler_equals_one_path_route"] = inp.one_path_check
    if curve_gather is not None:
        out["curve_gather"] = curve_gather
    kept = None
    if keep:
        kept = dict(batch=batch, inp=inp, prob=prob, cap=cap, res=res, K=K, B=B, chunk0=chunk_sizes[0] if B else 0,
                    knot_digests=inp.device_knot_digests(64) if (rank == 0 and B) else [])
    elif batch is not None:
        batch.close()
    return out, kept


def recorded_traffic(workload, paths, layout):
    """HBM bytes per sweep launch from the PMC counters of a recorded rocprofv3 run (profiles/pmc_sweep_traffic.json) --
    only when that run had this workload, batch size and spline layout; otherwise null"""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "pmc_sweep_traffic.json")))
    except Exception:
        return None, None, None
    for e in rec.get("runs", []):
        if e.get("workload") == workload and e.get("paths") == paths and e.get("layout") == layout:
            return e.get("hbm_bytes_per_launch"), e.get("command"), e.get("issue")
    return None, None, None


# ---------------------------------------------------------------------------------------------------------------------
# side measurements of a run (rank 0)
# ---------------------------------------------------------------------------------------------------------------------
def measure_resampler(hip, workload, knots, n_paths):
    """SURVEY.md 8f-1 beside the hot path: taught points -> knots by batotp_hip_resample (taught points resident)"""
    seeds = [5000 + k for k in range(min(n_paths, 64))]
    taught, sres_in = taught_points_f32(workload, seeds, knots)
    xs = [widen(workload, t) for t in taught]
    host = make_knots(workload, seeds[0], knots)
    prm = resample_params(workload, host[2])
    tiled = [xs[p % len(xs)] for p in range(n_paths)]
    best, same, nk = None, True, 0
    for _ in range(2):   # the second call finds the context's workspaces allocated
        r = capi.Resampled(hip, prm, tiled, [sres_in] * n_paths)
        ms = r.ms()
        same = bool(np.array_equal(r.knots(0)[: host[0].shape[0]], host[0]) and r.sres[0] == host[1])
        nk = int(r.n_knots.sum())
        r.close()
        best = ms if best is None else min(best, ms)
    hip.trim()
    return {"paths": n_paths, "knots": nk, "ms": best, "knots_per_s": nk / (best * 1e-3), "identical_to_one_path_route": same,
            "what": "remClosePts + adjust_s x2 + interpSpecial + uniform re-evaluation on the device (taught points resident)"}


def load_cpu_checker():
    """TEST INFRASTRUCTURE, cpu_baseline leg only: the CPU oracle behind the product's C-ABI (never measured as the product)"""
    return capi.Library(os.path.join(ROOT, "oracle", "_build", "libbatotp_oracle_abi.so"))


def cpu_baseline(kept, budget_s, out_prm=None, want_output=False, max_distinct=16, single_path=False):
    """the oracle (oracle/_build, the bit-identical C restatement of the reference's path) on the host cores, on a bounded sample
    of the SAME inputs: (i) one path on one thread, (ii) one path per thread on all host threads.  Both with the
    regions of test/main.cpp:90-96: sweeps only ("Accel. constraint integ.") and spline / dynamics build + sweeps.  Every sampled
    path's traversal time and step counts are compared with the GPU's.  single_path: the configuration IS one trajectory
    (BASELINE configs 2 / 3): the one-thread figure is the baseline, no all-core run."""
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    octx = capi.Context(load_cpu_checker(), 0)
    cores = os.cpu_count() or 1
    inp, prob, cap, K = kept["inp"], kept["prob"], kept["cap"], kept["K"]
    D = max(1, min(K, max_distinct))                      # distinct paths of the sample
    # ... resampled by the ORACLE (oracle/batotp_oracle_resample.c through dump_knots), never by the product: the CPU leg is
    # independent of the device from the taught points on, and the knots the GPU swept are compared with these bit for bit
    with cf.ThreadPoolExecutor(max_workers=min(D, cores)) as ex:
        hosts = list(ex.map(inp.oracle_knots, range(D)))
    import hashlib
    dig = kept.get("knot_digests") or []
    n_cmp = min(D, len(dig))
    knots_same = [bool(hashlib.sha256(np.ascontiguousarray(hosts[k][0]).tobytes()).hexdigest() == dig[k][0] and hosts[k][0].shape[1] == dig[k][1]
                       and hosts[k][1] == dig[k][2]) for k in range(n_cmp)]

    def cpu_batch(n_paths, passes=1):
        nk = [hosts[i % D][0].shape[1] for i in range(n_paths)]
        pr = capi.Problem.from_buffer_copy(bytes(prob))
        pr.flags &= ~(capi.F_COMPACT_SPLINES | capi.F_CURVES_IN_PLACE | capi.F_MVC_IN_CURVES)
        b = capi.Batch(octx, pr, nk, cap)
        for i in range(n_paths):
            b.upload_knots(i, [hosts[i % D][0]], [hosts[i % D][1]])
        prepare_dynamics(b, pr, n_paths)
        dt, sw = 0.0, 0.0
        for _ in range(passes):
            t = time.perf_counter()
            b.precompute(0); b.pointwise_mvc(); b.sweep(-1); b.sweep(+1)
            dt = time.perf_counter() - t
            sw = 1e-3 * (b.kernel_ms(3) + b.kernel_ms(4))
        rr = b.results()
        th0 = None
        if want_output:
            oo = capi.Output(b, out_prm, 0, 1)
            th0 = oo.rows(0)
            oo.close()
        b.close()
        return dt, sw, sum(nk), rr, th0

    t_one, sw_one, n_one, rows, th0 = cpu_batch(1, passes=2)      # one path = one busy thread
    info = {"unit": "waypoints/s", "kind": "port",
            "single_thread": {"value": n_one / t_one, "sweeps_only_value": n_one / max(sw_one, 1e-9), "seconds": t_one, "sweep_seconds": sw_one,
                              "steps_per_s": float(rows["steps_rev"][0] + rows["steps_fwd"][0]) / max(sw_one, 1e-9)}}
    n_sample = 1
    if single_path:
        info.update({"value": n_one / t_one, "cores": 1,
                     "sample": f"the configuration's one trajectory (N = {n_one}) on one host thread, same regions (K1-K4; sweeps alone "
                               f"{1e3 * sw_one:.0f} ms), oracle/ C restatement at -O3 -ffp-contract=off"})
    else:
        try:
            import psutil
            avail = psutil.virtual_memory().available
        except Exception:
            avail = 64 << 30
        pr0 = capi.Problem.from_buffer_copy(bytes(prob)); pr0.flags &= ~(capi.F_COMPACT_SPLINES | capi.F_CURVES_IN_PLACE | capi.F_MVC_IN_CURVES)
        per_path = 1.5 * bytes_per_path(pr0, WORKLOADS[inp.workload]["C"], float(n_one), cap)     # host copy + the checker's arrays
        by_mem = max(1, int(0.25 * avail / per_path))
        n_sample = int(max(1, min(2 * cores, (budget_s / max(t_one, 1e-3)) * cores, by_mem)))
        if n_sample > cores:
            n_sample = (n_sample // cores) * cores
        wall, sw, n_wp, rows, _ = cpu_batch(n_sample, passes=2)
        used = min(cores, n_sample)
        info.update({"value": n_wp / wall, "sweeps_only_value": n_wp / max(sw, 1e-9), "cores": used,
                     "sample": f"{n_sample} paths of the same workload ({D} distinct, N~{n_one}), OpenMP one path per thread on {used} of {cores} "
                               f"host threads, same regions (K1-K4), oracle/ C restatement at -O3 -ffp-contract=off"})
    res = kept["res"]
    idx = np.arange(n_sample) % D                          # GPU path p of a chunk holds distinct path p % K; D <= K
    ok = idx < res.shape[0]
    err = float(np.max(np.abs(res["t_total"][idx[ok]] - rows["t_total"][: n_sample][ok]))) if np.any(ok) else 0.0
    mism = int(np.count_nonzero((res["steps_fwd"][idx[ok]] != rows["steps_fwd"][: n_sample][ok]) | (res["steps_rev"][idx[ok]] != rows["steps_rev"][: n_sample][ok])))
    info["paths_compared"] = int(np.count_nonzero(ok))
    # the knots the GPU swept (device resampler) against the oracle resampler's, bit for bit (sha256 of the rows, N, sres)
    info["inputs_identical_to_oracle_resampler"] = bool(n_cmp > 0 and all(knots_same))
    info["inputs_compared"] = n_cmp
    return info, err, mism, th0


# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as a CHILD process (this process never
    touches a GPU) and pass its output and exit code through"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def launch_check(args, rank, world):
    """--launch-check: launcher, sharding and gather without any GPU work (gloo; what the CPU test-suite drives)"""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo")
    c = CONFIGS[args.config]
    total = args.paths or c["paths"]
    lo, hi = bdist.shard_range(total, rank, world) if c["scaling"] == "strong" else (0, total)
    rows = np.zeros(hi - lo, dtype=capi.RESULT_DTYPE)
    rows["n_fwd"] = np.arange(lo, hi) if c["scaling"] == "strong" else rank
    allrows = bdist.gather_results(rows) if world > 1 else rows
    # the rank's chunk plan on a 288 GB GPU (same arithmetic as measure(): 93 % of the free memory less 8 GB) and the sizes of
    # the curve gather (all_gather of the per-path point counts, then one packed (s, sdot) buffer per rank to
    # rank 0) with nominal curve lengths -- the size exchange is real, the buffers are not allocated
    w = WORKLOADS[c["workload"]]
    cfgd = w["cfg"]
    pr = capi.Problem()
    pr.n_joints, pr.n_cart = cfgd["n_joints"], (cfgd["n_cart"] if (cfgd.get("cart_vel_on") or cfgd.get("is_parallel")) else 0)
    pr.flags = (capi.F_TRQ_ON if cfgd.get("trq_on") else 0) | (capi.F_PARALLEL if cfgd.get("is_parallel") else 0)
    if cfgd.get("par2ser"):
        pr.flags |= capi.F_PAR2SER
    # the layout measure() picks: (value, second derivative) pairs for velocity / acceleration-only problems and for the cable robot
    # in serial form (every channel as pairs)
    if not (pr.flags & capi.F_TRQ_ON) or ((pr.flags & capi.F_PARALLEL) and (pr.flags & capi.F_PAR2SER)):
        pr.flags |= capi.F_NO_SAMPLES | capi.F_COMPACT_SPLINES
    knots = args.knots or c["knots"]
    cap = int(knots * w["cap"]) + 1024
    fit = max(1, int(0.93 * (288e9 - 8 * 2 ** 30) / bytes_per_path(pr, w["C"], float(knots), cap)))
    B = hi - lo
    K = max(1, min(args.distinct or c["distinct"], max(B, 1)))
    chunks = plan_chunks(B, K, fit)
    pts = (0.4 * knots * (1.0 + 0.5 * ((np.arange(lo, hi) % 7) / 7.0))).astype(np.int64)      # nominal forward-curve lengths
    cnt = torch.zeros(max(1, -(-total // world) if c["scaling"] == "strong" else total), dtype=torch.int64)
    cnt[: pts.shape[0]] = torch.from_numpy(pts)
    allcnt = [torch.zeros_like(cnt) for _ in range(world)]
    if world > 1:
        dist.all_gather(allcnt, cnt)
    else:
        allcnt = [cnt]
    recv_points = [int(t.sum().item()) for t in allcnt]
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "n_gpus": world, "config": args.config, "scaling": c["scaling"],
                          "paths_total": total if c["scaling"] == "strong" else total * world, "gathered_rows": int(allrows.shape[0]),
                          "rows_in_rank_order": bool(np.all(np.diff(allrows["n_fwd"]) >= 0)), "max_over_ranks": float(t.item()),
                          "paths_per_rank": B, "chunks_per_rank": chunks, "paths_that_fit_one_gpu": fit,
                          "curve_gather": {"points_per_rank": recv_points, "GB_to_rank0": 16e-9 * sum(recv_points[1:]),
                                           "what": "nominal forward-curve lengths; the per-path point counts were exchanged by all_gather "
                                                   "exactly as batotp_amd.dist.gather_curves does before its grouped send/recv"}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="fill7", choices=sorted(CONFIGS))
    ap.add_argument("--paths", type=int, default=0, help="override the configuration's number of paths (per GPU for weak, total for strong scaling)")
    ap.add_argument("--knots", type=int, default=0, help="override the configuration's target knots per path")
    ap.add_argument("--distinct", type=int, default=0, help="distinct seeded paths per GPU (tiled to the batch)")
    ap.add_argument("--group", type=int, default=0, help="lanes per path in the sweep kernel (0 = automatic)")
    ap.add_argument("--ppw", type=int, default=0, help="paths per wavefront in the sweep kernel (0 = automatic)")
    ap.add_argument("--prefetch", type=int, nargs=2, default=None, help="sweep prefetch bits, reverse forward (batotp_hip_set_sweep_prefetch)")
    ap.add_argument("--hold", type=int, nargs=2, default=None, help="sweep loop form, reverse forward (batotp_hip_set_sweep_hold; -2 automatic)")
    ap.add_argument("--spline-tiles", type=int, default=None, help="A/B: K1 in tiles of knots, 1 always / 0 never / -1 automatic (batotp_hip_set_spline_tiles)")
    ap.add_argument("--no-fast-forward", action="store_true", help="A/B: run every bisection iteration's check (batotp_hip_set_fast_forward 0)")
    ap.add_argument("--lean", action="store_true", help="one curve buffer per path and the pointwise values in it, whatever the configuration says (experiments)")
    ap.add_argument("--k3-form", type=int, default=None, help="A/B: per-knot evaluation kernel, 1 k_pointwise_va / 0 the general kernel (batotp_hip_set_k3_form)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sides", action="store_true", help="skip the side measurements (resampler, output stage, nested-loop cross-check)")
    ap.add_argument("--no-as-worded", action="store_true", help="default configuration only: skip the BASELINE configs as worded")
    ap.add_argument("--coefficient-rows", action="store_true",
                    help="keep four coefficients per knot and channel instead of the compact (value, second derivative) form")
    ap.add_argument("--launch-check", action="store_true", help="exercise launcher, sharding and gather only (gloo, no GPU work)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))   # nothing above touched a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: start it as `python bench.py --gpus {args.gpus}` or "
                         f"through torch.distributed.run --nproc-per-node {args.gpus}")
    if args.launch_check:
        return launch_check(args, rank, world)

    import torch
    import torch.distributed as dist

    dist_ctx = None
    if world > 1 or os.environ.get("BATOTP_BENCH_FORCE_DIST") == "1":   # the latter: exercise RCCL on one GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist_ctx = {"dist": dist, "dev": torch.device("cuda", local_rank)}
    torch.cuda.set_device(local_rank)
    hip = capi.Context(capi.load_hip(), local_rank)  # raises if the HIP extension or the GPU is missing
    if args.prefetch:
        hip.set_sweep_prefetch(*args.prefetch)
    if args.hold:
        hip.set_sweep_hold(*args.hold)
    if args.no_fast_forward:
        hip.set_fast_forward(False)
    if args.k3_form is not None:
        hip.set_k3_form(args.k3_form)
    if args.spline_tiles is not None:
        hip.set_spline_tiles(args.spline_tiles)

    default_run = args.config == "fill7" and not args.paths and not args.knots
    cpu_job = (None, None)
    out, kept = measure(hip, args.config, rank, world, args.steps, args.warmup, dist_ctx, args.paths, args.knots, args.group, args.ppw,
                        args.coefficient_rows, args.distinct, keep=True, lean=args.lean)
    workload = CONFIGS[args.config]["workload"]
    prob, batch = kept["prob"], kept["batch"]
    vel_acc_only = not (prob.flags & (capi.F_TRQ_ON | capi.F_CART_VEL_ON | capi.F_CART_ACC_ON))
    # The headline's sweep kernel (k_sweep8, flat loop) is behind a gate: the toolchain the library was built with must be the one the
    # loop was validated with, and a canary on this device must agree with the nested loops.  A closed gate silently costs a factor
    # of 1.7 (the nested loops of the general kernel run instead, same results) -- so it is reported at the top level and on stderr.
    gate = hip.flat_loop_status() if vel_acc_only else None
    out["flat_loop_gate"] = {"status": gate, "meaning": {1: "open: k_sweep8 with the flat stage / bisection loop", -1: "CLOSED: library built by a toolchain "
                             "the flat loop was not validated with (nested loops run instead, same results, ~1.7x slower)", -2: "CLOSED: the "
                             "on-device canary disagreed with the nested loops", -3: "CLOSED: the canary could not run", None: "not applicable "
                             "(torque or Cartesian limits: one path per wavefront)"}.get(gate, "?"),
                             "toolchain_built_with_validated_with": list(hip.library.toolchain())}
    if gate is not None and gate != 1 and rank == 0:
        print(f"bench: WARNING: the gate of the flat sweep loop is CLOSED (batotp_hip_flat_loop_status = {gate}: {out['flat_loop_gate']['meaning']}); "
              f"toolchain {hip.library.toolchain()}", file=sys.stderr)

    will_check = 1 if (not args.no_sides and batch is not None and vel_acc_only and kept["chunk0"] > 6144 and args.group in (0, 8)) else 0
    if dist_ctx is not None:
        # the check below contains a collective: every rank must take the same decision (a rank with an empty or smaller share
        # would otherwise skip the all_reduce the others wait in)
        flag = torch.tensor([will_check], dtype=torch.int64, device=dist_ctx["dev"])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        will_check = int(flag.item())
    if will_check:
        # the automatic loop form of the reverse sweep (flat stage / bisection loop) against the nested loops on the same
        # batch: the result rows must be identical; untimed.  EVERY rank checks its own batch (nobody waits in a barrier
        # while rank 0 sweeps), the verdict is the AND over the ranks.  (Up to 6144 paths the reverse sweep runs k_sweep1.)
        ref_rows = batch.results().tobytes()
        launch_default = batch.last_sweep_launch(-1)
        launch_default_fwd = batch.last_sweep_launch(+1)
        hip.set_sweep_hold(-1, -1)
        batch.sweep(-1); batch.sweep(+1)
        nested_rev, nested_fwd = batch.kernel_ms(3), batch.kernel_ms(4)
        same = batch.results().tobytes() == ref_rows
        hip.set_sweep_hold(-2, -2)
        if dist_ctx is not None:
            flag = torch.tensor([1 if same else 0], dtype=torch.int64, device=dist_ctx["dev"])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            same = bool(flag.item())
        out["nested_loop_cross_check"] = {"sweep_rev_ms": nested_rev, "sweep_fwd_ms": nested_fwd, "result_rows_identical": bool(same),
                                          "default_reverse_launch_lanes_ppw_hold": list(launch_default),
                                          "default_forward_launch_lanes_ppw_hold": list(launch_default_fwd),
                                          "flat_loop_gate": hip.flat_loop_status(), "toolchain": list(hip.library.toolchain()),
                                          "what": "the same batch once more through the general kernel k_sweep with the nested stage / bisection loops "
                                                  "in both sweeps (the default runs k_sweep8 -- flat loop, hold 4 reverse / 8 forward -- when its gate, "
                                                  "validated toolchain + on-device canary, is open: flat_loop_gate 1); every rank checks its own batch; untimed"}
    if rank == 0 and not args.no_sides and batch is not None:
        # SURVEY.md 8f-2 beside the hot path: the output stage of the first paths of the batch on the device
        wcfg = WORKLOADS[workload]["cfg"]
        covered = (wcfg["robot"] == "GENJNT" and not (prob.flags & capi.F_TRQ_ON)) or \
                  (wcfg["robot"] == "CSPR3DOF" and (prob.flags & capi.F_TRQ_ON) and (prob.flags & capi.F_PARALLEL))
        out_prm, hip_out0 = None, None
        if covered:
            out_prm = capi.OutputParams(prob.n_joints, capi.PATH_JOINT if wcfg["path_type"] == "JOINT" else capi.PATH_CART, prob.integ_res, 0.008, 5.0)
            n_out_paths = min(kept["chunk0"], 512)
            best = None
            for _ in range(2):  # the second call finds the context's workspace allocated
                o = capi.Output(batch, out_prm, 0, n_out_paths)
                ms, pts = o.ms(), int(o.n_pts.sum())
                hip_out0 = o.rows(0)
                o.close()
                best = ms if best is None else min(best, ms)
            out["output_stage"] = {"paths": n_out_paths, "points": pts, "ms": best, "points_per_s": pts / (best * 1e-3),
                                   "out_res": out_prm.out_res, "out_smooth_fact": out_prm.out_smooth_fact,
                                   "what": "s(t) spline + re-sampling at constant time steps + joint spline evaluation + smoothing / "
                                           "down-sampling (+ re-interpolation when out_res < integ_res) on the device"}
        cpu_job = (out_prm, hip_out0)
    if batch is not None:
        batch.close()
    # what the CPU baseline needs later lives on the host (the device batch is released now)
    kept_host = {k: kept[k] for k in ("inp", "prob", "cap", "K", "res", "chunk0", "knot_digests")} if (rank == 0 and kept is not None) else None
    kept = None
    hip.trim()

    if rank == 0 and not args.no_sides and default_run and device_resamplable(workload):
        out["resample"] = measure_resampler(hip, workload, CONFIGS[args.config]["knots"], 1024)

    # the BASELINE configs as worded, beside the headline (every rank takes part: cfg4 / cfg5 shard their batch over the ranks)
    worded_host = {}
    if default_run and not args.no_as_worded:
        worded = {}
        for name in ("cfg2", "cfg3", "cfg4", "cfg5", "cfg5_distinct2048"):
            try:
                w, wk = measure(hip, name, rank, world, CONFIGS[name].get("steps", AS_WORDED_STEPS), 1, dist_ctx, keep=True)
                if wk is not None and wk.get("batch") is not None:
                    wk["batch"].close()
                if rank == 0 and wk is not None and wk.get("chunk0"):
                    worded_host[name] = {k: wk[k] for k in ("inp", "prob", "cap", "K", "res", "chunk0", "knot_digests")}
                wk = None
                worded[name] = {k: w.get(k) for k in ("value", "unit", "steps", "warmup", "ms_per_step", "ms_per_step_median", "ms_per_step_min_max",
                                                       "scaling", "data", "config", "kernel_ms", "steps_per_knot", "steps_per_path",
                                                       "us_per_integration_step", "gathered_rows", "curve_gather", "launch",
                                                       "paths_with_error_status", "swapped_seeds", "slowest_over_mean_path")}
                worded[name]["roofline_frac"] = w["roofline"]["frac"]
                worded[name]["roofline_bound"] = w["roofline"]["bound"]
            except Exception as e:  # the main line must not depend on a side measurement
                if world > 1:
                    raise          # ... but ranks must not drift apart in the collectives
                worded[name] = {"error": f"{type(e).__name__}: {str(e)[-300:]}"}
            hip.trim()
        out["as_worded"] = worded
    hip.close()
    if dist_ctx is not None:
        dist.destroy_process_group()
    # CPU baseline: the oracle (bit-identical port of the reference's path) on the host cores, bounded sample -- rank 0, after
    # the last collective (the other ranks are done: nobody is parked in a barrier meanwhile), whatever the world size
    if rank == 0 and not args.no_cpu_baseline and not args.no_sides and kept_host is not None and kept_host["chunk0"]:
        out_prm, hip_out0 = cpu_job
        info, err, mism, th0 = cpu_baseline(kept_host, args.cpu_seconds, out_prm, hip_out0 is not None, max_distinct=64)
        out["cpu_baseline"] = info
        out["vs_cpu_baseline"] = out["value"] / info["value"]
        out["traversal_time_err_s"] = err
        out["step_count_mismatches"] = mism
        if th0 is not None and "output_stage" in out:
            out["output_stage"]["identical_to_oracle"] = bool(th0.tobytes() == hip_out0.tobytes())
    # ... and beside every BASELINE configuration as worded (north_star: each number "next to the reference CPU BA timed on the
    # same box's host cores"): one trajectory on one thread for cfg 2 / 3, one path per thread on all host threads for cfg 4 / 5
    if rank == 0 and not args.no_cpu_baseline:
        for name, kh in worded_host.items():
            try:
                single = CONFIGS[name]["paths"] == 1
                info, err, mism, _ = cpu_baseline(kh, max(2.0, 0.5 * args.cpu_seconds), max_distinct=(1 if single else 32), single_path=single)
                w = out["as_worded"][name]
                w["cpu_baseline"] = info
                w["vs_cpu_baseline"] = w["value"] / info["value"]
                w["traversal_time_err_s"] = err
                w["step_count_mismatches"] = mism
            except Exception as e:
                out["as_worded"][name]["cpu_baseline"] = {"error": f"{type(e).__name__}: {str(e)[-300:]}"}
    if rank == 0:
        # RCCL writes a version banner through C stdio on rank 0: push it out first, so that the JSON line is the last line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        try:     # an untruncated copy for profiles/ (the driver keeps only the tail of long lines)
            if os.path.isdir(os.path.join(ROOT, "gpurun_out")) and default_run and not args.no_as_worded:   # (not the profiling passes)
                open(os.path.join(ROOT, "gpurun_out", "bench_line_full.json"), "w").write(json.dumps(out, indent=1))
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
