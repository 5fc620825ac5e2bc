#!/usr/bin/env python3
"""Parity of a build with -DBK_PARK_RELOAD_AGENT=1 (kernels.hip.h: the solve kernels reload their parked values with agent-scope loads)
against the shipped library, on the GPU: the resampler's golden cases (knots bit for bit, also against the fixture) and the Thomas solve of
long series through both of its kernels.  usage: check_reload_agent.py <variant .so>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from batotp_amd import capi  # noqa: E402
import helpers  # noqa: E402

base = capi.Context(capi.load_hip(), 0)
var = capi.Context(capi.Library(sys.argv[1]), 0)
n_cases = 0
for name in helpers.RESAMPLE_CASES:
    c = helpers.ResampleCase(name)
    got = []
    for ctx in (base, var):
        r = capi.Resampled(ctx, c.params, [c.x], [c.sres_in])
        assert int(r.status[0]) == 0, name
        got.append(r.knots(0).tobytes())
        r.close()
    assert got[0] == got[1] == c.y.tobytes(), name
    n_cases += 1
rng = np.random.default_rng(5)
n_series = 0
for n in (34000, 100003, 400000):
    y = np.cumsum(rng.standard_normal(n))
    a = capi.spline_lanes_kat(base, y)
    b = capi.spline_lanes_kat(var, y)
    assert a[0].tobytes() == b[0].tobytes() == a[1].tobytes() == b[1].tobytes() and a[2] == b[2] == 0, n
    n_series += 1
print(f"reload-agent build: {n_cases} resampler cases = shipped library = fixtures bit for bit; {n_series} long series through both solve kernels identical")
