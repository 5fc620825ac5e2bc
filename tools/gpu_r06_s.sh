#!/bin/bash
# round 6, the last four GPU minutes: the one-path resampler in a loop with TWO paths taking turns in every process (a stale read would show)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
timeout 285 python tools/stress_resample_one.py --alternate --jobs 64 --seconds 1 --life 205 --calls-per-context 16 --resident 1024 --out gpurun_out/stress_rs2 > gpurun_out/r06_s_stress.log 2>&1
tail -12 gpurun_out/r06_s_stress.log | cut -c1-1800
du -sh gpurun_out/stress_rs2
