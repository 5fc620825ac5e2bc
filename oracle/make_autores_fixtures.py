#!/usr/bin/env python3
"""tests/golden/<case>/autores.npz -- what the automatic integration resolution (reference ba.cpp:493-556, the class default
ba.h:309) makes of a golden case: knots, integration step, rewritten s weights and scale type.

Provenance: the reference's prebuilt batest cannot produce them (test/main.cpp:53 switches the rule off), so these vectors
were written ONCE, in round 4, by oracle/dump_autores.cpp linked against the round-3 host resampler of this repository (the
statement-level restatement of ba.cpp:95-863, deleted in the same round in favour of the device resampler).  They pin the
rule as restated there; tests compare the oracle's bo_resample and the HIP resampler with them bit for bit.
To regenerate: check out the commit before the deletion, build oracle/dump_autores.cpp as its header says, run this script.
"""
import os, shutil, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.environ.get("DUMP_AUTORES", "/tmp/autores/dump_autores")
CASES = ["UR5", "RR", "KUKA-LWR-IV", "CSPR3DOF", "GEN7DOF", "synth_cspr_s3", "synth_cspr_s9_dup", "KUKA_cartacc", "UR5_nocartacc", "synth_cspr_s11_decim"]
for c in (sys.argv[1:] or CASES):
    src = os.path.join(ROOT, "tests", "golden", c)
    with tempfile.TemporaryDirectory() as w:
        for f in os.listdir(src):
            shutil.copy(os.path.join(src, f), w)
        r = subprocess.run([TOOL, "config.dat"], cwd=w, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        b = open(os.path.join(w, "autores.bin"), "rb").read()
    N, nJ, nC = (int(v) for v in np.frombuffer(b, "<i8", 3, 0))
    sres, integ = np.frombuffer(b, "<f8", 2, 24)
    sw = np.frombuffer(b, "<f8", 3, 40)
    scale = int(np.frombuffer(b, "<i8", 1, 64)[0])
    y = np.frombuffer(b, "<f8", (nJ + nC) * N, 72).reshape(nJ + nC, N)
    np.savez_compressed(os.path.join(src, "autores.npz"), y=y, sres=np.float64(sres), integ_res=np.float64(integ), s_weights=sw.copy(), scale_type=np.int64(scale))
    print(c, N, nJ, nC, float(sres), float(integ), sw, scale)
