#!/bin/bash
# round 4: the certified fast-forward in k_sweep8: parity subset, then A/B of batotp_hip_set_fast_forward 0 / 1 (forward) / 3 (both) on the reduced batch
set -u
ulimit -c 0
mkdir -p gpurun_out
if [ "${1:-tests}" = "tests" ]; then
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "flat or lane_groupings or compact_splines or in_place or fuzz or ragged" 2>&1 | tail -6 )
fi
for ff in 0 1 3; do
  echo "== fast-forward $ff"
  timeout 900 python tools/run_hotpath.py --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 1 --fast-forward $ff --variants "1:4:8:8:8,1:6:8:8:8" 2>&1 | grep variant | cut -c1-125
done 2>&1 | tee gpurun_out/r04_ff.log
