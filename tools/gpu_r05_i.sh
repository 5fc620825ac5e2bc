#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_resample.py tests/test_gpu_fuzz.py -x -q -m gpu -k "resampl or taught or golden_knots or ragged or chunked or kinematics or short or auto" > gpurun_out/r05_i_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_i_tests.log
tail -4 gpurun_out/r05_i_tests.log
for w in ur6 gen7 cspr; do python tools/bench_resample.py --workload $w --paths 1024 --knots 100000 2>&1 | tail -1 | cut -c1-400; done
