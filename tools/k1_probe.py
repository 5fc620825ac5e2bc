#!/usr/bin/env python3
"""K1 probe: precompute time of the tiled against the sequential spline build on a resident batch, and the number of series the
tiled build handed back to the sequential kernel (expected 0).  usage: tools/k1_probe.py [paths] [knots]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from batotp_amd import capi

paths = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
knots = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
hip = capi.Context(capi.load_hip(), 0)
inp = bench.Inputs(hip, "gen7", knots, [1000 + k for k in range(64)])
prob = capi.Problem.from_buffer_copy(bytes(inp.prob))
prob.flags |= capi.F_COMPACT_SPLINES | capi.F_CURVES_IN_PLACE
cap = 4096
for tiles in (True, False, True):
    hip.set_spline_tiles(tiles)
    b = capi.Batch(hip, prob, [int(inp.n_knots[p % inp.K]) for p in range(paths)], cap)
    inp.fill(b, paths)
    ms = []
    for _ in range(3):
        b.precompute(0)
        ms.append(b.kernel_ms(1))
    print("tiles" if tiles else "sequential", paths, "paths", "precompute ms", ["%.2f" % m for m in ms], "fallbacks", b.spline_tile_fallbacks() if tiles else "-")
    b.close()
