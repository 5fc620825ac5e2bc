#!/bin/bash
# LDS window of k_rs_special: resampler parity tests, the three resampler timings, per-kernel breakdown of the GEN7DOF call
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_resample.py tests/test_gpu_fuzz.py tests/test_gpu_output.py -k "resampl or output" -q -m gpu -p no:cacheprovider > gpurun_out/r05_k_resample_tests.log 2>&1
echo "resample tests rc=$?" >> gpurun_out/r05_k_resample_tests.log
tail -4 gpurun_out/r05_k_resample_tests.log
for w in ur6 gen7 cspr; do python tools/bench_resample.py --workload $w --paths 1024 --knots 100000 2>&1 | tail -1 | cut -c1-300; done | tee gpurun_out/r05_k_resample_timings.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -- python3 $GRAFT_REPO_ROOT/tools/bench_resample.py --workload gen7 --paths 1024 --knots 100000 > /tmp/prof_rs.log 2>&1
f=$(find /tmp/prof_rs -name '*kernel_stats.csv' | head -1)
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r05_k_resample_gen7_kernel_stats.csv
head -12 "$f" | cut -c1-200
