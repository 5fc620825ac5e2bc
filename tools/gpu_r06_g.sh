#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
( timeout 1500 python -m pytest tests/test_gpu_resample.py tests/test_gpu_output.py tests/test_gpu_parity.py -x -q -m gpu \
   -k "pos3 or UR5 or checksums or block_upload or pose or golden_knots or traj_out or product_batch_driver or concurrent" 2>&1 | tail -8 ) > gpurun_out/r06_g_tests.log 2>&1
tail -5 gpurun_out/r06_g_tests.log
