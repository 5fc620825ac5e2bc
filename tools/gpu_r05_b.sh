#!/bin/bash
# round 5: the whole GPU suite (fail-fast off: every test runs), then the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider --durations=40 > gpurun_out/r05_b_full_gpu_tests.log 2>&1
echo "full gpu suite rc=$?" >> gpurun_out/r05_b_full_gpu_tests.log
tail -25 gpurun_out/r05_b_full_gpu_tests.log
python bench.py > gpurun_out/r05_b_bench.json 2> gpurun_out/r05_b_bench.err
echo "bench rc=$?"
tail -c 1500 gpurun_out/r05_b_bench.json; tail -5 gpurun_out/r05_b_bench.err
cp gpurun_out/bench_line_full.json gpurun_out/r05_b_bench_line_full.json 2>/dev/null
