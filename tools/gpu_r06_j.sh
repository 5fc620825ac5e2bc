#!/bin/bash
# round 6: the trace test, then 64 concurrent one-path processes beside a resident batch with the guard naming the stage of a disagreement
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_resample.py -x -q -m gpu -k "intermediate_stage or checksums or concurrent" 2>&1 | tail -4
timeout 2400 python tools/repro_concurrent_resample.py --rounds 6 > gpurun_out/r06_j_repro.log 2>&1
tail -14 gpurun_out/r06_j_repro.log
