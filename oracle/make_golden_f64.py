#!/usr/bin/env python3
"""fp64 golden vectors read out of THE REFERENCE BINARY while it runs -- build container only (needs /root/reference and
/opt/rocm/bin/rocgdb).

oracle/make_golden.py pins the oracle with the files the reference writes, which are float32 (ba.cpp:2745-2748 casts on write).
This script closes that gap (SURVEY.md 8c iii-v): it runs the reference's prebuilt, unstripped bin/batest under rocgdb in batch
mode on the same case directories and reads the reference's OWN fp64 memory:

  * at the return of each BATOTP::BA::sweep (ba.cpp:979-1195): traj.sMVC, traj.sdot (every point, raw doubles), traj.nPts,
    traj.tTotalTraj  ->  tests/golden/<case>/ref_curves_f64.npz (small cases) or their sha256 in expected_f64.json (the three
    BASELINE-size cases);
  * at ~128 sampled calls per sweep of BATOTP::BA::applyAccelConstraintsBisectionPt (ba.cpp:1248-1332) and of BATOTP::BA::sdotLim
    (ba.cpp:1204-1236): the cursor state on entry and everything the routine leaves behind on return
    ->  tests/golden/<case>/ref_point_kats.npz.

Member offsets inside BATOTP::Traj come from this repository's own layout-identical header (oracle/traj_offsets.cpp); before a
value is trusted the script checks them against what the reference prints itself (step counts, traversal time).  The fixtures are
DATA; no reference source is copied, the binary is copied to a scratch directory only because its mount is not executable.

Usage:  python oracle/make_golden_f64.py [case ...]
"""
import hashlib
import json
import os
import re
import shutil
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden as mg  # noqa: E402  (the case builders)

GDB = "/opt/rocm/bin/rocgdb"
OFFSETS_TOOL = os.path.join(ROOT, "oracle", "_build", "traj_offsets")
SAMPLES_PER_SWEEP = 128

GDB_SCRIPT = r'''
import gdb, json, struct
OFF = json.loads("""%(offsets)s""")
PLAN = json.loads("""%(plan)s""")
gdb.execute("set pagination off"); gdb.execute("set confirm off"); gdb.execute("set startup-with-shell off")
inf = gdb.selected_inferior()
def rd(addr, n): return bytes(inf.read_memory(addr, n))
def u64(addr): return struct.unpack("<Q", rd(addr, 8))[0]
def f64(addr): return struct.unpack("<d", rd(addr, 8))[0]
def i32(addr): return struct.unpack("<i", rd(addr, 4))[0]
def vec(addr):  # std::vector<double>: begin, end, end of storage
    b, e = u64(addr), u64(addr + 8)
    return rd(b, e - b) if e > b else b""
def reg(name): return int(gdb.parse_and_eval("$" + name)) & 0xFFFFFFFFFFFFFFFF
def addr_of(sym): return int(gdb.parse_and_eval("(unsigned long)&'%%s'" %% sym))
A_SWEEP = addr_of("BATOTP::BA::sweep(BATOTP::Traj&)")
A_ACCEL = addr_of("BATOTP::BA::applyAccelConstraintsBisectionPt(BATOTP::Traj&, double&, int&)")
A_SDLIM = [int(l.split()[0], 16) for l in gdb.execute("info functions BATOTP::BA::sdotLim", to_string=True).splitlines() if "sdotLim" in l and l.strip().startswith("0x")][0]
bp_sweep = gdb.Breakpoint("*%%d" %% A_SWEEP)
bp_accel = gdb.Breakpoint("*%%d" %% A_ACCEL) if PLAN["kats"] else None
bp_sdlim = gdb.Breakpoint("*%%d" %% A_SDLIM) if PLAN["kats"] else None
out = open("f64_dump.bin", "wb")
def emit(tag, payload):
    out.write(struct.pack("<4sQ", tag, len(payload))); out.write(payload)
sweep_no = 0      # 1 = reverse, 2 = forward
ret_bp = None
traj = 0
n_acc = n_sdl = 0
gdb.execute("run", to_string=True)
while True:
    try:
        pc = reg("pc")
    except gdb.error:
        break
    if pc == A_SWEEP:
        sweep_no += 1
        traj = reg("rsi")
        ret = u64(reg("rsp"))
        ret_bp = gdb.Breakpoint("*%%d" %% ret, temporary=True)
        ret_pc = ret
        n_acc = n_sdl = 0
        if bp_accel:
            bp_accel.ignore_count = 0; bp_sdlim.ignore_count = 0
    elif sweep_no and pc == ret_pc:
        emit(b"SWP%%d" %% sweep_no, struct.pack("<Id", i32(traj + OFF["nPts"]) & 0xFFFFFFFF, f64(traj + OFF["tTotalTraj"])))
        emit(b"SMV%%d" %% sweep_no, vec(traj + OFF["sMVC"]))
        emit(b"SDT%%d" %% sweep_no, vec(traj + OFF["sdot"]))
        if sweep_no == 2:
            break
    elif bp_accel and pc == A_ACCEL:
        t, p_sddot, p_niter = reg("rsi"), reg("rdx"), reg("rcx")
        ins = struct.pack("<ddqd", f64(t + OFF["sCur"]), f64(t + OFF["sdotCur"]), i32(t + OFF["curSegC"]), f64(p_sddot))
        gdb.execute("finish", to_string=True)
        rc = struct.unpack("<i", struct.pack("<I", reg("rax") & 0xFFFFFFFF))[0]
        outs = struct.pack("<ddddqqq", f64(t + OFF["sdotCur"]), f64(p_sddot), f64(t + OFF["sddotL"]), f64(t + OFF["sddotH"]),
                           i32(p_niter), rc, i32(t + OFF["curSegC"]))
        emit(b"ACC%%d" %% sweep_no, ins + outs)
        n_acc += 1
        bp_accel.ignore_count = PLAN["stride"][sweep_no - 1] if n_acc >= 2 else 0   # the two bootstrap calls, then every k-th
        continue
    elif bp_sdlim and pc == A_SDLIM:
        t, p_sdot = reg("rsi"), reg("rdx")
        ins = struct.pack("<ddq", f64(t + OFF["sCur"]), f64(p_sdot), i32(t + OFF["curSegMVC"])) + \
              (vec(t + OFF["thetaDpt"]) + b"\0" * 64)[:64] + rd(t + OFF["CartAccCoeffs"], 8)
        gdb.execute("finish", to_string=True)
        outs = struct.pack("<dq", f64(p_sdot), i32(t + OFF["curSegMVC"]))
        emit(b"SDL%%d" %% sweep_no, ins + outs)
        n_sdl += 1
        bp_sdlim.ignore_count = PLAN["stride"][sweep_no - 1] if n_sdl >= 2 else 0
        continue
    try:
        gdb.execute("continue", to_string=True)
    except gdb.error:
        break
out.close()
gdb.execute("kill")
'''


def offsets():
    src = os.path.join(ROOT, "oracle", "traj_offsets.cpp")
    if not os.path.exists(OFFSETS_TOOL) or os.path.getmtime(OFFSETS_TOOL) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-std=c++11", "-O0", "-Wno-invalid-offsetof", "-I" + os.path.join(ROOT, "batotp_amd", "host"),
                               "-I" + os.path.join(ROOT, "include"), "-o", OFFSETS_TOOL, src])
    return subprocess.check_output([OFFSETS_TOOL], text=True)


def parse_dump(path):
    recs = {}
    b = open(path, "rb").read()
    at = 0
    while at < len(b):
        tag, n = struct.unpack_from("<4sQ", b, at)
        at += 12
        recs.setdefault(tag.decode(), []).append(b[at:at + n])
        at += n
    return recs


def run_case(name):
    build, full = mg.CASES[name]
    dst = os.path.join(mg.GOLD, name)
    exp = json.load(open(os.path.join(dst, "expected.json")))
    stride = [max(1, (6 * exp["n_rev"]) // SAMPLES_PER_SWEEP), max(1, (6 * exp["n_fwd"]) // SAMPLES_PER_SWEEP)]
    with tempfile.TemporaryDirectory() as work:
        build(work)
        exe = os.path.join(work, "batest_ref")
        shutil.copy(mg.REF_BIN, exe)
        os.chmod(exe, 0o755)
        script = os.path.join(work, "dump.py")
        open(script, "w").write(GDB_SCRIPT % dict(offsets=offsets(), plan=json.dumps({"kats": bool(full), "stride": stride})))
        r = subprocess.run([GDB, "-batch", "-x", script, "--args", exe, "config.dat"], cwd=work, capture_output=True, text=True, timeout=3600)
        dump = os.path.join(work, "f64_dump.bin")
        if not os.path.exists(dump):
            raise RuntimeError(f"{name}: no dump\n{r.stdout[-3000:]}\n{r.stderr[-3000:]}")
        recs = parse_dump(dump)
    curves = {}
    for k, which in ((1, "rev"), (2, "fwd")):
        n_pts, t_total = struct.unpack("<Id", recs[f"SWP{k}"][0])
        s = np.frombuffer(recs[f"SMV{k}"][0], "<f8").copy()
        sd = np.frombuffer(recs[f"SDT{k}"][0], "<f8").copy()
        # the offsets are right if the reference's own printout agrees with what was read at them
        assert n_pts == s.size == sd.size, (name, which, n_pts, s.size, sd.size)
        assert n_pts == exp["n_" + which], (name, which, n_pts, exp)   # the point count the reference printed (ba.cpp:1138-1152)
        assert np.all(np.diff(s) > 0) and s[0] == 0.0, (name, which, "s must ascend from 0")
        curves[which] = (s, sd, t_total)
    assert abs(curves["fwd"][2] - exp["t_total_print"]) < 5e-4, (name, curves["fwd"][2], exp["t_total_print"])
    # ... and the float32 files the reference wrote must be these doubles rounded
    if full:
        ref32 = mg.pathgen.read_s_sdot(os.path.join(dst, "ref_s-sdot.dat"))
        for k, which in enumerate(("rev", "fwd")):
            assert np.array_equal(curves[which][0].astype(np.float32), ref32[k][1]) and np.array_equal(curves[which][1].astype(np.float32), ref32[k][2]), (name, which)
    out = {"case": name, "t_rev": curves["rev"][2], "t_total": curves["fwd"][2], "t_total_hex": float(curves["fwd"][2]).hex(),
           "n_rev_pts": int(curves["rev"][0].size), "n_fwd_pts": int(curves["fwd"][0].size),
           "sha256_rev_f64": hashlib.sha256(curves["rev"][0].tobytes() + curves["rev"][1].tobytes()).hexdigest(),
           "sha256_fwd_f64": hashlib.sha256(curves["fwd"][0].tobytes() + curves["fwd"][1].tobytes()).hexdigest(),
           "reference": "fp64 memory of the prebuilt /root/reference/bin/batest read under rocgdb at the return of BA::sweep"}
    json.dump(out, open(os.path.join(dst, "expected_f64.json"), "w"), indent=1)
    if full:
        np.savez_compressed(os.path.join(dst, "ref_curves_f64.npz"), rev_s=curves["rev"][0], rev_sd=curves["rev"][1],
                            fwd_s=curves["fwd"][0], fwd_sd=curves["fwd"][1], t_rev=np.float64(curves["rev"][2]), t_total=np.float64(curves["fwd"][2]))
        acc, sdl = [], []
        for k, d in ((1, -1), (2, 1)):
            for raw in recs.get(f"ACC{k}", []):
                s_cur, sdot_in, seg_in, sddot_in, sdot_out, sddot_out, l, h, n_iter, rc, seg_out = struct.unpack("<ddqdddddqqq", raw)
                acc.append((d, s_cur, sdot_in, seg_in, sddot_in, sdot_out, sddot_out, l, h, n_iter, rc, seg_out))
            first = None
            for raw in recs.get(f"SDL{k}", []):
                s_cur, sdot_in, seg_in = struct.unpack_from("<ddq", raw, 0)
                th = struct.unpack_from("<8d", raw, 24)
                cart0, sdot_out, seg_out = struct.unpack_from("<ddq", raw, 24 + 64)
                # BA::_sdotMin: seeded with the first call's INPUT (ba.cpp:1027-1030), re-seeded with its output (:1034-1035)
                sdot_min = sdot_in if first is None else first
                if first is None:
                    first = sdot_out
                sdl.append((d, s_cur, sdot_in, sdot_min, seg_in, cart0) + th + (sdot_out, seg_out))
        acc_dt = np.dtype([("dir", "<i4"), ("s_cur", "<f8"), ("sdot_in", "<f8"), ("seg_in", "<i8"), ("sddot_in", "<f8"), ("sdot_out", "<f8"),
                           ("sddot_out", "<f8"), ("sddot_l", "<f8"), ("sddot_h", "<f8"), ("n_iter", "<i8"), ("rc", "<i8"), ("seg_out", "<i8")])
        sdl_dt = np.dtype([("dir", "<i4"), ("s_cur", "<f8"), ("sdot_in", "<f8"), ("sdot_min", "<f8"), ("seg_mvc_in", "<i8"), ("cart0", "<f8"),
                           ("theta_d_pt", "<f8", (8,)), ("sdot_out", "<f8"), ("seg_mvc_out", "<i8")])
        np.savez_compressed(os.path.join(dst, "ref_point_kats.npz"), accel=np.array(acc, dtype=acc_dt),
                            sdot_lim=np.array([(r[0], r[1], r[2], r[3], r[4], r[5], r[6:14], r[14], r[15]) for r in sdl], dtype=sdl_dt))
        print(f"{name}: rev {curves['rev'][0].size} fwd {curves['fwd'][0].size} points, T = {curves['fwd'][2]!r}, {len(acc)} + {len(sdl)} point KATs")
    else:
        print(f"{name}: rev {curves['rev'][0].size} fwd {curves['fwd'][0].size} points, T = {curves['fwd'][2]!r} (digests only)")


if __name__ == "__main__":
    for n in (sys.argv[1:] or list(mg.CASES)):
        run_case(n)
