#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_zz_as_worded.py -x -q -m gpu -k cfg5 --durations=3 > gpurun_out/r05_f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_f_tests.log
tail -8 gpurun_out/r05_f_tests.log
python bench.py --config cfg5 --steps 3 --warmup 1 --no-sides 2> gpurun_out/r05_f_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms',round(d['ms_per_step'],1),{k:round(v,1) for k,v in d['kernel_ms'].items()},'chunks',d['config']['chunks_per_step'],'err paths',d['paths_with_error_status'],'vs cpu',d.get('vs_cpu_baseline'),d.get('traversal_time_err_s'),d.get('step_count_mismatches'),(d.get('cpu_baseline') or {}).get('inputs_identical_to_oracle_resampler'))
"
tail -3 gpurun_out/r05_f_err.txt
