#!/bin/bash
# round 6: 64 concurrent one-path processes beside a resident batch, with the two-evaluations guard reporting (3 rounds x 1024 paths =
# 6144+ evaluations of the device resampler)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
timeout 1500 python tools/repro_concurrent_resample.py --rounds 3 > gpurun_out/r06_i_repro.log 2>&1
tail -8 gpurun_out/r06_i_repro.log
