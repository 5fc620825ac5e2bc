#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
for k in 1 2 3; do
echo "== with poison, run $k"; BATOTP_TEST_POISON=1 timeout 600 python -m pytest tests/test_gpu_fuzz.py -q -m gpu -k "hard_problems" 2>&1 | tail -4
done
( BATOTP_TEST_POISON=1 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -40 ) > gpurun_out/r06_f_poison_suite.log 2>&1
tail -12 gpurun_out/r06_f_poison_suite.log
