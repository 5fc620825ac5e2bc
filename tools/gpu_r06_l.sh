#!/bin/bash
# round 6: paths per wavefront below 8 (more wavefronts than two per SIMD: the reverse kernel's 165 VGPRs admit three), holds around the optimum
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
V="1:4:8:8:8:3,1:4:8:7:8:3,1:4:8:6:8:3,1:4:8:6:8:2,1:3:8:8:8:3,1:3:8:8:8:2,1:4:8:8:8:3"
timeout 900 python tools/run_hotpath.py --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 2 --variants "$V" 2>&1 | grep -E "variant" | cut -c1-150 | tee gpurun_out/r06_l_ab.log
