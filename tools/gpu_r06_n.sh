#!/bin/bash
# round 6: focused stress of the Thomas-solve kernels (the stage the traced event named) under 64 concurrent processes beside a sweeping batch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
timeout 900 python tools/stress_spline_lanes.py --jobs 64 --seconds 150 --n 34000 --resident 1024 > gpurun_out/r06_n_stress.log 2>&1
tail -30 gpurun_out/r06_n_stress.log | cut -c1-400
