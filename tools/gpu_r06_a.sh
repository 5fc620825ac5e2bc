#!/bin/bash
# round 6, first GPU pass: the certificate phase of the reverse k_sweep8 (batotp_hip_set_cert_hold): parity subset, A/B of the holds on the
# reduced batch, in-kernel sections, the headline batch with the best candidates
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_output.py -x -q -m gpu \
    -k "flat or hard_problems or gated or segment_cursor or in_place or compact_splines" 2>&1 | tail -8 ) > gpurun_out/r06_a_tests.log 2>&1
tail -4 gpurun_out/r06_a_tests.log
# reduced batch (16 384 paths of 2e4 knots, 64 distinct): form:holdRev:holdFwd:ppw:group:certHold
V=""
for c in 0 1 2 3 4 5 6; do for h in 4 5 6; do V="$V,1:$h:8:8:8:$c"; done; done
timeout 1200 python tools/run_hotpath.py --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 2 --variants "${V#,}" 2>&1 | grep -E "variant|rows|step" | cut -c1-150 | tee gpurun_out/r06_a_ab.log
for c in 0 1 3; do
  echo "== sections, cert hold $c (reduced batch)"
  timeout 600 python tools/sweep8_sections.py --lib batotp_amd/csrc/libbatotp_hip_s8prof.so --paths 16384 --knots 20000 --distinct 64 --cert-hold $c 2>&1 | grep -v "^forward" | head -24
done > gpurun_out/r06_a_sections.txt 2>&1
cat gpurun_out/r06_a_sections.txt
