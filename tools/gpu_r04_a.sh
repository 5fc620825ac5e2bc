#!/bin/bash
# round 4: k_sweep8 on the GPU: parity subset, then A/B against the flat instantiation of k_sweep on one resident batch
#   usage: tools/gpu_r04_a.sh [variants] [pytest -k expression | none]
set -u
mkdir -p gpurun_out
K=${2:-flat or lane_groupings or compact or in_place or shared_reciprocal}
if [ "$K" != "none" ]; then
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "$K" 2>&1 | tail -15 ) > gpurun_out/r04_a_tests.log 2>&1
tail -5 gpurun_out/r04_a_tests.log
fi
( timeout 1500 python tools/run_hotpath.py --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 1 \
   --variants "${1:-0:4:-1:8:8,1:4:-1:8:8,1:4:8:8:8,1:3:8:8:8,1:5:8:8:8,0:4:-1:16:4,1:4:8:16:4,1:3:6:16:4,1:5:8:16:4,1:6:8:16:4}" 2>&1 | tail -30 ) > gpurun_out/r04_a_ab.log 2>&1
cat gpurun_out/r04_a_ab.log
