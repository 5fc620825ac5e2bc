#!/usr/bin/env python3
"""Generate tests/golden_self/* -- regression fixtures for configurations the REFERENCE CANNOT RUN.

BASELINE config 3 (KUKA LWR IV+ with torque limits) has no reference model: Robot::dynSerial
(/root/reference/batotp/robot.cpp:349-360) knows the two-link arm only and the reference binary segfaults on
KUKA + isTrqConOn.  So nothing here comes from the reference: the expected values are what THIS
repository's oracle produces today (host BA library linked against the oracle shim, oracle/_build/batest_oracle),
written down so that a later change of the chain model, the oracle or the kernels shows up as a diff.  They are
"parity unpinned" fixtures and the tests that use them say so.

For every case: a config.dat (edited copy of tests/golden/KUKA-LWR-IV/config.dat, or a synthetic one) + its path,
run through batest_oracle (log -> step counts, s-sdot.dat, traj_out.dat) and dump_knots (fp64 knots + problem).

Usage:  python oracle/make_selfgolden.py [case ...]
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

BUILD = os.path.join(ROOT, "oracle", "_build")
GOLD = os.path.join(ROOT, "tests", "golden")
SELF = os.path.join(ROOT, "tests", "golden_self")

KUKA_VEL = [110, 110, 128, 128, 204, 184, 184]           # tests/golden/KUKA-LWR-IV/config.dat
KUKA_ACC = [137.5, 157.1, 213.3, 213.3, 510.0, 460.0, 613.3]
KUKA_TRQ = [176, 176, 100, 100, 100, 38, 38]             # rated joint torques of the LWR IV+ [N m]
NAN7 = [float("nan")] * 7                                # "NAN" = -max (reference ba.cpp:2020-2028)


def edit_config(src, dst, subs):
    out = []
    for line in open(src):
        for key, val in subs.items():
            if re.search(r"//\s*" + re.escape(key) + r"\b", line):
                line = f"{val} // {key} (edited)\n"
        out.append(line)
    open(dst, "w").write("".join(out))


def shipped_kuka(subs):
    def build(work):
        src = os.path.join(GOLD, "KUKA-LWR-IV")
        shutil.copy(os.path.join(src, "KUKApath.dat"), os.path.join(work, "KUKApath.dat"))
        edit_config(os.path.join(src, "config.dat"), os.path.join(work, "config.dat"), subs)
    return build


def synth_kuka(seed, n_coarse, **cfg):
    def build(work):
        th = pathgen.kuka_like_fine(seed, n_coarse)
        pathgen.write_traj_bin(os.path.join(work, "path.dat"), 0.01, th, None)
        kw = dict(robot="KUKA", is_parallel=0, n_joints=7, n_cart=3, traj_file="path.dat", is_bin=1, path_type="JOINT",
                  degrees=1, jnt_vel=KUKA_VEL, jnt_acc_on=1, jnt_acc=KUKA_ACC, trq_on=1, trq_max=KUKA_TRQ, trq_min=NAN7,
                  cart_vel_on=0, cart_vel=0.6, integ_res=0.005, max_integ_time=2000000.0, s_weights=(0, 1, 0), scale_type=1,
                  theta_res=0.3, theta_res2=0.3, out_res=0.005, out_smooth=1)
        kw.update(cfg)
        pathgen.write_config(os.path.join(work, "config.dat"), **kw)
    return build


# name -> (builder, keep the input path file in the fixture)
CASES = {
    # the reference's shipped KUKA path, limits doubled / quadrupled so that the torque limits bind; no Cartesian limit
    "KUKA_trq": (shipped_kuka({"isTrqConOn": 1, "JntTrqMax": "20 90 20 45 8 8 1", "JntTrqMin": "NAN NAN NAN NAN NAN NAN NAN",
                               "isCartVelConOn": 0, "JntVelLims": "220 220 256 256 408 368 368",
                               "JntAccLims": "550 628 853 853 2040 1840 2453"}), False),
    # torque limits barely above the gravity load (the sweep crawls), Cartesian speed limit on (38 spline channels)
    "KUKA_trq_tight": (shipped_kuka({"isTrqConOn": 1, "JntTrqMax": "9 68 9 31 3 4.3 0.45", "JntTrqMin": "NAN NAN NAN NAN NAN NAN NAN",
                                     "JntVelLims": "220 220 256 256 408 368 368",
                                     "JntAccLims": "550 628 853 853 2040 1840 2453"}), False),
    # rated limits, shipped configuration otherwise
    "KUKA_trq_rated": (shipped_kuka({"isTrqConOn": 1, "JntTrqMax": "176 176 100 100 100 38 38",
                                     "JntTrqMin": "NAN NAN NAN NAN NAN NAN NAN"}), False),
    # synthetic BASELINE-config-3 shape, small
    "synth_kuka_s12_trq": (synth_kuka(12, 30), True),
    "synth_kuka_s13_trq_half": (synth_kuka(13, 24, trq_max=[88, 88, 50, 50, 50, 19, 19], jnt_vel=[2 * v for v in KUKA_VEL],
                                           jnt_acc=[4 * a for a in KUKA_ACC]), True),
}


def run_case(name):
    build, keep_path = CASES[name]
    dst = os.path.join(SELF, name)
    shutil.rmtree(dst, ignore_errors=True)
    os.makedirs(dst)
    with tempfile.TemporaryDirectory() as work:
        build(work)
        inputs = sorted(os.listdir(work))
        run = subprocess.run([os.path.join(BUILD, "batest_oracle"), "config.dat"], cwd=work, capture_output=True, text=True)
        if run.returncode != 0 or not os.path.exists(os.path.join(work, "s-sdot.dat")):
            raise RuntimeError(f"{name}: batest_oracle failed\n{run.stdout[-2000:]}")
        log = run.stdout
        m_rev = re.search(r"rev\. integ\.:\s*(\d+) steps", log)
        m_fwd = re.search(r"fwd\. integ\.:\s*(\d+) steps.*?traj time\. ([0-9.]+) sec", log)
        fails = sum(int(v) for v in re.findall(r"error: (\d+) point\(s\) did not respect", log))
        dk = subprocess.run([os.path.join(BUILD, "dump_knots"), "config.dat"], cwd=work, capture_output=True, text=True)
        if dk.returncode != 0:
            raise RuntimeError(f"{name}: dump_knots failed\n{dk.stdout[-2000:]}")
        kb = open(os.path.join(work, "knots.bin"), "rb").read()
        N, nJ, nC = np.frombuffer(kb, "<i8", 3, 0)
        sres = float(np.frombuffer(kb, "<f8", 1, 24)[0])
        y = np.frombuffer(kb, "<f8", int((nJ + nC) * N), 32).reshape(int(nJ + nC), int(N))
        prob = np.frombuffer(open(os.path.join(work, "problem.bin"), "rb").read(), np.uint8)
        curves = pathgen.read_s_sdot(os.path.join(work, "s-sdot.dat"))
        expected = {
            "case": name, "n_knots": int(N), "n_joints": int(nJ), "n_cart": int(nC), "sres": sres, "sres_hex": float(sres).hex(),
            "n_rev": int(m_rev.group(1)), "n_fwd": int(m_fwd.group(1)), "t_total_print": float(m_fwd.group(2)),
            "ref_bisect_fail_msgs": fails, "inputs": inputs,
            "sha256_rev": hashlib.sha256(curves[0][1].tobytes() + curves[0][2].tobytes()).hexdigest(),
            "sha256_fwd": hashlib.sha256(curves[1][1].tobytes() + curves[1][2].tobytes()).hexdigest(),
            "sha256_traj_out": hashlib.sha256(open(os.path.join(work, "traj_out.dat"), "rb").read()).hexdigest(),
            "reference": "NONE (the reference has no dynamics model for this robot): self-generated by this repository's oracle, "
                         "parity unpinned",
        }
        shutil.copy(os.path.join(work, "config.dat"), os.path.join(dst, "config.dat"))
        if keep_path:
            for f in inputs:
                if f != "config.dat":
                    shutil.copy(os.path.join(work, f), os.path.join(dst, f))
        np.savez_compressed(os.path.join(dst, "knots.npz"), y=y, sres=np.float64(sres), problem=prob)
        shutil.copy(os.path.join(work, "s-sdot.dat"), os.path.join(dst, "ref_s-sdot.dat"))
        json.dump(expected, open(os.path.join(dst, "expected.json"), "w"), indent=1)
        print(f"{name}: N={N} rev={expected['n_rev']} fwd={expected['n_fwd']} T={expected['t_total_print']} fails={fails}")


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(CASES)):
        run_case(n)
