#!/bin/bash
# round 6, second GPU pass: A/B on ONE box of (A) certificate phase + cursor quotients through cached reciprocals, (B) certificate phase
# only, (C) neither = the reverse kernel of rounds 4-5, (D) cached reciprocals only; reduced batch; then parity subset with (A)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
for lib in libbatotp_hip.so libbatotp_hip_expB.so libbatotp_hip_expC.so libbatotp_hip_expD.so; do
  V="1:4:8:8:8:3,1:5:8:8:8:3,1:4:8:8:8:2,1:4:8:8:8:4"
  case $lib in *expC*|*expD*) V="1:4:8:8:8:0,1:5:8:8:8:0";; esac
  echo "== $lib"
  timeout 900 python tools/run_hotpath.py --lib batotp_amd/csrc/$lib --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 2 --variants "$V" 2>&1 | grep -E "variant|step" | cut -c1-150
done | tee gpurun_out/r06_b_ab.log
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_output.py -x -q -m gpu \
    -k "flat or hard_problems or gated or segment_cursor or in_place or compact_splines or lane_groupings or ragged" 2>&1 | tail -8 ) > gpurun_out/r06_b_tests.log 2>&1
tail -4 gpurun_out/r06_b_tests.log
