#!/usr/bin/env python3
"""Generate tests/golden/* -- runs ONLY in the build container (needs /root/reference).

For every case this script
  1. assembles a work directory: a config.dat plus its trajectory file (either the reference's own
     shipped example data, input/<robot>/*, or a synthetic input written by batotp_amd.pathgen);
  2. runs THE REFERENCE ITSELF on it -- the prebuilt /root/reference/bin/batest (real Eigen, GCC
     5.3.1; started through the dynamic loader because the mount is read-only / not executable) --
     and keeps its outputs s-sdot.dat (both integrated curves, float32) and traj_out.dat;
  3. runs oracle/_build/dump_knots (host resampling of this repo, no device) to record the fp64
     knot values and the problem description that feed the hot path.
The fixtures are DATA (inputs and expected outputs).  No reference source is copied.

Usage:  python oracle/make_golden.py [case ...]
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from batotp_amd import pathgen  # noqa: E402

REF = "/root/reference"
REF_BIN = os.path.join(REF, "bin", "batest")
LOADER = "/lib64/ld-linux-x86-64.so.2"
DUMP = os.path.join(ROOT, "oracle", "_build", "dump_knots")
GOLD = os.path.join(ROOT, "tests", "golden")


def edit_config(src: str, dst: str, subs: dict):
    """copy a config.dat replacing the value of the lines whose trailing comment names a key"""
    out = []
    for line in open(src):
        for key, val in subs.items():
            if re.search(r"//\s*" + re.escape(key) + r"\b", line):
                line = f"{val} // {key} (edited)\n"
        out.append(line)
    open(dst, "w").write("".join(out))


def shipped(robot_dir: str, subs=None):
    def build(work):
        src = os.path.join(REF, "input", robot_dir)
        for f in os.listdir(src):
            if f.endswith(".m") or f.startswith("trajKuka"):
                continue  # generators / raw logs are not inputs of batest
            shutil.copy(os.path.join(src, f), os.path.join(work, f))
            os.chmod(os.path.join(work, f), 0o644)
        if subs:
            edit_config(os.path.join(src, "config.dat"), os.path.join(work, "config.dat"), subs)
    return build


def shipped_ur5_positions_only():
    """the shipped UR5 example (joints and tool poses taught together, path type BOTH) with the three position columns only:
    nCart = 3, no orientations -- the reference then skips aa2qVect / q2aaVect and resamples both channel sets as taught
    (ba.cpp:184-192, 245-262, 1726-1736, 1922)"""
    def build(work):
        src = os.path.join(REF, "input", "UR5")
        rows = [", ".join(x.strip() for x in line.split(",")[:10]) for line in open(os.path.join(src, "trajUR.csv")).read().splitlines()]
        open(os.path.join(work, "trajUR.csv"), "w").write("\n".join(rows) + "\n")      # timestamp, j1..j6, x, y, z
        edit_config(os.path.join(src, "config.dat"), os.path.join(work, "config.dat"), {"nCart": 3})
    return build


def with_repeats(x, every):
    """repeat every `every`-th taught point and the last one (remClosePts has to drop them again; the repeated last
    point takes its tail rule)"""
    cols = []
    for i in range(x.shape[1]):
        cols.append(i)
        if i % every == every - 1:
            cols.append(i)
    cols.append(x.shape[1] - 1)
    return np.ascontiguousarray(x[:, cols])


def synth_gen7dof(seed, n_coarse, repeat_every=0, **cfg):
    def build(work):
        th = pathgen.gen7dof_fine(seed, n_coarse)
        if repeat_every:
            th = with_repeats(th, repeat_every)
        pathgen.write_traj_bin(os.path.join(work, "path.dat"), 0.01, th, None)
        kw = dict(robot="GENJNT", is_parallel=0, n_joints=7, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                  degrees=0, jnt_vel=[5] * 7, jnt_acc_on=1, jnt_acc=[10] * 7, integ_res=0.01, max_integ_time=200000.0,
                  theta_res=0.1, theta_res2=0.1, out_res=0.008, out_smooth=5)
        kw.update(cfg)
        pathgen.write_config(os.path.join(work, "config.dat"), **kw)
    return build


def synth_ur(seed, n_coarse, **cfg):
    def build(work):
        th = pathgen.ur_like_fine(seed, n_coarse)
        pathgen.write_traj_bin(os.path.join(work, "path.dat"), 0.01, th, None)
        kw = dict(robot="GENJNT", is_parallel=0, n_joints=6, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                  degrees=1, jnt_vel=[160] * 6, jnt_acc_on=1, jnt_acc=[573, 573, 573, 1146, 1146, 1146], integ_res=0.008,
                  max_integ_time=200000.0, theta_res=0.3, theta_res2=0.3, out_res=0.008, out_smooth=1)
        kw.update(cfg)
        pathgen.write_config(os.path.join(work, "config.dat"), **kw)
    return build


def synth_cspr(seed, n_coarse, repeat_every=0, **cfg):
    def build(work):
        ca = pathgen.cspr_fine(seed, n_coarse)
        if repeat_every:
            ca = with_repeats(ca, repeat_every)
        pathgen.write_traj_bin(os.path.join(work, "path.dat"), 0.005, None, ca)
        kw = dict(robot="CSPR3DOF", is_parallel=1, n_joints=3, n_cart=3, traj_file="path.dat", is_bin=1, path_type="CART",
                  degrees=0, jnt_vel=[4] * 3, jnt_acc_on=1, jnt_acc=[8] * 3, trq_on=1, trq_max=[12] * 3, trq_min=[1] * 3,
                  cart_vel_on=1, cart_vel=4.0, cart_acc_on=0, cart_acc=100.0, integ_res=0.01, max_integ_time=200000.0,
                  s_weights=(0, 0, 1), scale_type=2, theta_res=0.01, theta_res2=0.01, cart_res=0.01, cart_res2=0.01,
                  out_res=0.02, out_smooth=1, par2ser=1)
        kw.update(cfg)
        pathgen.write_config(os.path.join(work, "config.dat"), **kw)
    return build


# name -> (builder, keep_full_outputs)
CASES = {
    # the reference's own five examples (BASELINE config 1 is RR)
    "RR": (shipped("RR"), True),
    "UR5": (shipped("UR5"), True),
    "GEN7DOF": (shipped("GEN7DOF"), True),
    "CSPR3DOF": (shipped("CSPR3DOF"), True),
    "KUKA-LWR-IV": (shipped("KUKA-LWR-IV"), True),
    # same data, other branches of the constraint code
    "CSPR3DOF_par": (shipped("CSPR3DOF", {"isPar2Ser": 0}), True),          # parallel-mechanism torque branch (LU solves)
    "RR_acc": (shipped("RR", {"isJntAccConOn": 1, "JntAccLims": "900 900"}), True),  # torque + joint accel
    "UR5_nocartacc": (shipped("UR5", {"isCartAccConOn": 0}), True),
    "UR5_pos3": (shipped_ur5_positions_only(), True),                        # BOTH path with nCart = 3: positions without orientations
    "KUKA_cartacc": (shipped("KUKA-LWR-IV", {"isCartAccConOn": 1, "CartAccMax": 2.0}), True),
    # solveLinSys through Eigen's Jacobi SVD (util.cpp:421-438): per-knot conversion (isPar2Ser = 1) / every constraint check (0)
    "CSPR3DOF_svd": (shipped("CSPR3DOF", {"isSVD": 1}), True),
    "CSPR3DOF_par_svd": (shipped("CSPR3DOF", {"isSVD": 1, "isPar2Ser": 0}), True),
    # synthetic inputs in the shapes of BASELINE configs 2, 4, 5 (small, kept in full)
    "synth_gen7dof_s0": (synth_gen7dof(0, 60), True),
    "synth_gen7dof_s1_vel": (synth_gen7dof(1, 40, jnt_acc_on=0), True),
    "synth_ur_s2": (synth_ur(2, 80), True),
    "synth_cspr_s3": (synth_cspr(3, 20), True),
    "synth_cspr_s5": (synth_cspr(5, 60), True),
    # taught paths with repeated points: remClosePts really removes something (also on the device resampler's path kinds)
    "synth_gen7dof_s6_dup": (synth_gen7dof(6, 30, repeat_every=7), True),
    "synth_cspr_s9_dup": (synth_cspr(9, 12, repeat_every=5), True),
    # input decimation + smoothing (ba.cpp:195-242)
    "synth_gen7dof_s10_decim": (synth_gen7dof(10, 45, decim=3, smooth=3), True),
    "synth_cspr_s11_decim": (synth_cspr(11, 14, decim=2), True),
    # s = teach time (sWeights 1 0 0: adjust_s returns at once, ba.cpp:416): the knots are the taught points after close-point
    # removal, input smoothing / decimation -- the branch the host library keeps (BA::keepTaughtSpacing with the filters of
    # batotp_amd/host/util.cpp); repeated points so that remClosePts really thins the path, the last one repeated too
    "synth_gen7dof_s14_teachtime": (synth_gen7dof(14, 24, repeat_every=7, decim=3, smooth=3, s_weights=(1, 0, 0), scale_type=0), True),
    "synth_gen7dof_s15_teachtime_decim2": (synth_gen7dof(15, 18, repeat_every=5, decim=2, s_weights=(1, 0, 0), scale_type=0), True),
    # BASELINE-size single paths: digest only (sha256 of the float32 curves + 1-in-64 samples)
    "synth_gen7dof_s4_50k": (synth_gen7dof(4, 871), False),
    "synth_ur_s7_100k": (synth_ur(7, 500), False),
    "synth_cspr_s8_40k": (synth_cspr(8, 200), False),
}


def run_case(name):
    build, full = CASES[name]
    dst = os.path.join(GOLD, name)
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    with tempfile.TemporaryDirectory() as work:
        build(work)
        inputs = sorted(os.listdir(work))
        ref = subprocess.run([LOADER, REF_BIN, "config.dat"], cwd=work, capture_output=True, text=True)
        if not os.path.exists(os.path.join(work, "s-sdot.dat")):
            raise RuntimeError(f"{name}: reference produced no s-sdot.dat\n{ref.stdout[-2000:]}\n{ref.stderr[-2000:]}")
        log = ref.stdout
        m_rev = re.search(r"rev\. integ\.:\s*(\d+) steps", log)
        m_fwd = re.search(r"fwd\. integ\.:\s*(\d+) steps.*?traj time\. ([0-9.]+) sec", log)
        n_knots = re.search(r"after splineFact: (\d+)", log)
        fails = len(re.findall(r"applyAccelConstraintsBisectionPt\(\) error", log))
        dk = subprocess.run([DUMP, "config.dat"], cwd=work, capture_output=True, text=True)
        if dk.returncode != 0:
            raise RuntimeError(f"{name}: dump_knots failed\n{dk.stdout[-2000:]}")
        kb = open(os.path.join(work, "knots.bin"), "rb").read()
        N, nJ, nC = np.frombuffer(kb, "<i8", 3, 0)
        sres = float(np.frombuffer(kb, "<f8", 1, 24)[0])
        y = np.frombuffer(kb, "<f8", int((nJ + nC) * N), 32).reshape(int(nJ + nC), int(N))
        prob = np.frombuffer(open(os.path.join(work, "problem.bin"), "rb").read(), np.uint8)
        curves = pathgen.read_s_sdot(os.path.join(work, "s-sdot.dat"))
        expected = {
            "case": name, "n_knots": int(N), "n_knots_ref": int(n_knots.group(1)), "n_joints": int(nJ), "n_cart": int(nC),
            "sres": sres, "sres_hex": float(sres).hex(),
            "n_rev": int(m_rev.group(1)), "n_fwd": int(m_fwd.group(1)), "t_total_print": float(m_fwd.group(2)),
            "ref_bisect_fail_msgs": fails, "inputs": inputs,
            "sha256_rev": hashlib.sha256(curves[0][1].tobytes() + curves[0][2].tobytes()).hexdigest(),
            "sha256_fwd": hashlib.sha256(curves[1][1].tobytes() + curves[1][2].tobytes()).hexdigest(),
            "reference": "prebuilt /root/reference/bin/batest run in the build container",
        }
        assert expected["n_knots"] == expected["n_knots_ref"], (name, expected)
        for f in inputs:
            shutil.copy(os.path.join(work, f), os.path.join(dst, f))
        if full:
            np.savez_compressed(os.path.join(dst, "knots.npz"), y=y, sres=np.float64(sres), problem=prob)
            shutil.copy(os.path.join(work, "s-sdot.dat"), os.path.join(dst, "ref_s-sdot.dat"))
            shutil.copy(os.path.join(work, "traj_out.dat"), os.path.join(dst, "ref_traj_out.dat"))
        else:
            np.savez_compressed(os.path.join(dst, "ref_curves_sampled.npz"),
                                rev_s=curves[0][1][::64], rev_sd=curves[0][2][::64],
                                fwd_s=curves[1][1][::64], fwd_sd=curves[1][2][::64])
            expected["sha256_knots"] = hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest()
            expected["sha256_traj_out"] = hashlib.sha256(open(os.path.join(work, "traj_out.dat"), "rb").read()).hexdigest()
            for f in inputs:  # the big synthetic input is regenerated from its seed by the tests
                if f.endswith(".dat") and f != "config.dat":
                    os.remove(os.path.join(dst, f))
        json.dump(expected, open(os.path.join(dst, "expected.json"), "w"), indent=1)
        print(f"{name}: N={N} rev={expected['n_rev']} fwd={expected['n_fwd']} T={expected['t_total_print']} fails={fails}")


if __name__ == "__main__":
    names = sys.argv[1:] or list(CASES)
    for n in names:
        run_case(n)
