#!/usr/bin/env python3
"""time batotp_hip_precompute alone for several batch shapes (K1 scaling study)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from batotp_amd import capi
hip = capi.Context(capi.load_hip(), 0)
prob = capi.make_problem(6, 0, flags=capi.F_JNT_ACC_ON | capi.F_NO_SAMPLES, jnt_vel_max=[160]*6, jnt_acc_max=[573]*6, integ_res=0.008)
rng = np.random.default_rng(0)
for B, N in ((64, 100000), (1024, 100000), (4096, 100000), (4096, 10000), (16384, 10000), (64, 1000)):
    y = np.cumsum(rng.standard_normal((6, N)) * 0.1, axis=1)
    b = capi.Batch(hip, prob, [N] * B, 16)
    for p in range(B):
        b.upload_knots(p, [y], [0.3])
    for _ in range(2):
        t = time.perf_counter(); b.precompute(1); dt = time.perf_counter() - t
    print(f"B={B} N={N}: precompute {dt*1e3:.2f} ms (event {b.kernel_ms(1):.2f} ms) -> {dt/N*1e9:.1f} ns per knot-row, {B*N/dt:.3e} knots/s")
    b.close()
