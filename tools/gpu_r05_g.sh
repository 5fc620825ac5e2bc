#!/bin/bash
# round 5: velocity / acceleration-only batches between the one-path kernel and k_sweep8: one or two paths per wavefront of k_sweep1
# against the 8-lane kernel, GEN7DOF N = 5e4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for paths in 2048 3072 4096 6144 8192 12288; do
  for v in "64 1" "64 2" "8 0"; do
    set -- $v
    python tools/run_hotpath.py --workload gen7 --paths $paths --knots 50000 --distinct 64 --group $1 --ppw $2 --reps 1 --tag "paths $paths lanes $1 ppw $2" 2>&1 | tail -2
  done
done 2>&1 | tee gpurun_out/r05_g_midsize.log
