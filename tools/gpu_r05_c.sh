#!/bin/bash
# round 5: A/B of k_sweep8 variants (alternative builds of the library) on one reduced and one full-size batch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for lib in libbatotp_hip.so $@; do
  for rep in 1 2; do
    python tools/run_hotpath.py --lib batotp_amd/csrc/$lib --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 1 --tag "$lib small" 2>&1 | tail -2
  done
done > gpurun_out/r05_c_ab.log 2>&1
for lib in libbatotp_hip.so $@; do
  python tools/run_hotpath.py --lib batotp_amd/csrc/$lib --workload gen7 --paths 13312 --knots 100000 --distinct 64 --group 8 --reps 1 --tag "$lib full" 2>&1 | tail -2
done >> gpurun_out/r05_c_ab.log 2>&1
cat gpurun_out/r05_c_ab.log
