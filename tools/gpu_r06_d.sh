#!/bin/bash
# round 6, fourth GPU pass: the certain-failure certificate of k_sweep1 (tests, cfg 5 with 2048 seeds as they come), then the suite's
# resampler / output / parity files under BATOTP_TEST_POISON=1
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
( timeout 1500 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_zz_as_worded.py -x -q -m gpu \
    -k "cable or cspr or CSPR or cfg5 or tension" 2>&1 | tail -8 ) > gpurun_out/r06_d_tests.log 2>&1
tail -4 gpurun_out/r06_d_tests.log
for cfg in cfg5_distinct2048 cfg5; do
python bench.py --config $cfg --steps 1 --warmup 1 --no-sides --no-cpu-baseline 2> gpurun_out/r06_d_$cfg.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg ms',round(d['ms_per_step'],1),{k:round(v,1) for k,v in d['kernel_ms'].items()},'bad',d['paths_with_error_status'],'swapped',d['swapped_seeds'],d['slowest_over_mean_path'],d['steps_per_path'])
"
done 2>&1 | tee gpurun_out/r06_d_cfg5.log
