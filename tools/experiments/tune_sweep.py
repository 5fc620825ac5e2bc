#!/usr/bin/env python3
"""Times the two sweep launches of one resident batch under several settings of the 8-lane sweep kernel
(batotp_hip_set_sweep_hold, batotp_hip_set_paths_per_wave) and checks that the results do not change.

    python tools/tune_sweep.py --workload ur6 --paths 16384 --configs=-1:-1:0,5:-1:0,5:5:0

A configuration is hold_reverse:hold_forward:paths_per_wave (-1 = nested loops, 0 = automatic paths per wavefront)."""
import argparse
import concurrent.futures as cf
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402
from batotp_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ur6")
    ap.add_argument("--paths", type=int, default=16384)
    ap.add_argument("--knots", type=int, default=100000)
    ap.add_argument("--distinct", type=int, default=32)
    ap.add_argument("--configs", default="-1:-1:0,5:-1:0,5:5:0")
    ap.add_argument("--reps", type=int, default=1)
    a = ap.parse_args()
    hip = capi.Context(capi.load_hip(), 0)
    hip.set_sweep_group(8)
    K = min(a.distinct, a.paths)
    with cf.ThreadPoolExecutor(max_workers=min(K, os.cpu_count() or 1)) as ex:
        base = list(ex.map(lambda s: bench.make_knots(a.workload, s, a.knots), [1000 + k for k in range(K)]))
    prob = base[0][2]
    if prob.flags & capi.F_NO_SAMPLES:
        prob.flags |= capi.F_COMPACT_SPLINES
    nk = [base[p % K][0].shape[1] for p in range(a.paths)]
    cap = int(max(nk) * {"ur6": 0.5, "gen7": 2.2, "cspr": 0.6}[a.workload]) + 1024
    b = capi.Batch(hip, prob, nk, cap)
    for p in range(a.paths):
        b.upload_knots(p, [base[p % K][0]], [base[p % K][1]])
    b.precompute(0)
    first = None
    for cfg in a.configs.split(","):
        hr, hf, ppw = (int(x) for x in cfg.split(":"))
        hip.set_sweep_hold(hr, hf)
        hip.set_paths_per_wave(ppw)
        best = None
        for _ in range(a.reps):
            b.sweep(-1)
            b.sweep(+1)
            ms = (b.kernel_ms(3), b.kernel_ms(4))
            if best is None or sum(ms) < sum(best):
                best = ms
        r = b.results()
        h = hashlib.sha256(r.tobytes())
        for p in (0, 1, a.paths // 2 + 3, a.paths - 1):
            for d in (-1, 1):
                s, sd = b.curve(p, d)
                h.update(s.tobytes()); h.update(sd.tobytes())
        digest = h.hexdigest()[:16]
        if first is None:
            first = digest
        print(json.dumps({"hold_rev": hr, "hold_fwd": hf, "ppw": ppw, "rev_ms": round(best[0], 1), "fwd_ms": round(best[1], 1),
                          "digest": digest, "same_as_first": digest == first}), flush=True)


if __name__ == "__main__":
    main()
