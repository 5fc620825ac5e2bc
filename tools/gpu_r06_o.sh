#!/bin/bash
# round 6: the whole one-path resampler in a loop, traced, 64 processes at a time beside a sweeping batch (tools/stress_resample_one.py)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
timeout 1100 python tools/stress_resample_one.py --jobs 64 --seconds 560 --life 60 --calls-per-context 8 --resident 1024 --out gpurun_out/stress_rs > gpurun_out/r06_o_stress.log 2>&1
tail -40 gpurun_out/r06_o_stress.log | cut -c1-1800
du -sh gpurun_out/stress_rs
