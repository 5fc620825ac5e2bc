#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: adds the resampler fixture `resample.npz` to every golden case the device
resampler covers.

For each tests/golden/<case> holding knots.npz, the host half of the pipeline (oracle/_build/dump_knots,
the same tool that produced knots.npz) is run on the case's config and taught path; it reports whether
the configuration is in the device resampler's scope and dumps struct batotp_resample_params.  The
script checks that the knots it gets are byte-identical to the committed knots.npz, then writes
   resample.npz : params (raw struct bytes), traj_file (name of the taught-path file in the case
                  directory), n_in, sres_in
and, for the configurations the device output stage covers,
   output.npz   : params (raw struct batotp_output_params); expected output = the case's ref_traj_out.dat
The expected output of the resampler is knots.npz itself (y, sres) -- the knots behind the s-sdot /
trajectory outputs that are byte-identical to the reference binary's.

Usage: python oracle/make_resample_fixtures.py        (needs `make -C oracle`)"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TOOL = os.path.join(ROOT, "oracle", "_build", "dump_knots")


def main():
    made = []
    for case in sorted(os.listdir(GOLD)):
        d = os.path.join(GOLD, case)
        if not os.path.exists(os.path.join(d, "knots.npz")):
            continue
        with tempfile.TemporaryDirectory() as work:
            for f in os.listdir(d):
                if not f.startswith("ref_") and not f.endswith(".npz") and not f.endswith(".json"):
                    shutil.copy(os.path.join(d, f), work)
            r = subprocess.run([TOOL, "config.dat"], cwd=work, capture_output=True, text=True)
            assert r.returncode == 0, (case, r.stdout[-1500:])
            raw = open(os.path.join(work, "resample.bin"), "rb").read()
            supported = int(np.frombuffer(raw, "<i4", 1, 0)[0])
            kb = open(os.path.join(work, "knots.bin"), "rb").read()
            N, nJ, nC = (int(v) for v in np.frombuffer(kb, "<i8", 3, 0))
            sres = float(np.frombuffer(kb, "<f8", 1, 24)[0])
            y = np.frombuffer(kb, "<f8", (nJ + nC) * N, 32).reshape(nJ + nC, N)
            z = np.load(os.path.join(d, "knots.npz"))
            assert y.tobytes() == np.ascontiguousarray(z["y"]).tobytes() and sres == float(z["sres"]), case
            oraw = open(os.path.join(work, "output.bin"), "rb").read()
            if int(np.frombuffer(oraw, "<i4", 1, 0)[0]):
                np.savez(os.path.join(d, "output.npz"), params=np.frombuffer(oraw[4:], np.uint8))
                print(f"{case:28s} output stage fixture written")
            if not supported:
                print(f"{case:28s} host resampler only")
                continue
            tb = open(os.path.join(work, "taught.bin"), "rb").read()
            n_in = int(np.frombuffer(tb, "<i8", 1, 0)[0])
            sres_in = float(np.frombuffer(tb, "<f8", 1, 24)[0])
            traj_file = [f for f in os.listdir(d) if f.endswith(".dat") and f not in ("config.dat",) and not f.startswith("ref_")]
            if len(traj_file) == 1:
                np.savez(os.path.join(d, "resample.npz"), params=np.frombuffer(raw[4:], np.uint8), traj_file=traj_file[0],
                         n_in=n_in, sres_in=sres_in)
            else:
                # text (CSV) input: keep the few taught points themselves, as the host reader delivers them
                nJ, nC = (int(v) for v in np.frombuffer(tb, "<i8", 2, 8))
                x = np.frombuffer(tb, "<f8", (nJ + nC) * n_in, 32).reshape(nJ + nC, n_in)
                np.savez(os.path.join(d, "resample.npz"), params=np.frombuffer(raw[4:], np.uint8), traj_file="", n_in=n_in,
                         sres_in=sres_in, x=x)
            made.append(case)
            print(f"{case:28s} n_in={n_in} -> N={N}")
    print("fixtures:", ", ".join(made))


if __name__ == "__main__":
    sys.exit(main())
