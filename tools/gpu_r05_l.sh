#!/bin/bash
# output stage: timing and per-kernel breakdown
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python tools/bench_output.py --workload gen7 --paths 512 2>&1 | tail -1 | tee gpurun_out/r05_l_output_timing.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_out -- python3 $GRAFT_REPO_ROOT/tools/bench_output.py --workload gen7 --paths 512 --calls 2 > /tmp/prof_out.log 2>&1
f=$(grep -l k_out $(find /tmp/prof_out -name '*kernel_stats.csv') | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r05_l_output_gen7_kernel_stats.csv
head -16 "$f" | cut -c1-160
cd $GRAFT_REPO_ROOT
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_output.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -5
