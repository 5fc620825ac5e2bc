#!/bin/bash
# round 5: two paths per wavefront in k_sweep1 (cable robot): parity, then cfg 5 on one GPU with one / two paths per wavefront
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_paths_per_wavefront or pairs_for_all" > gpurun_out/r05_d_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r05_d_tests.log
tail -5 gpurun_out/r05_d_tests.log
for v in "--group 0 --ppw 0" "--lean --group 64 --ppw 1" "--lean --group 64 --ppw 2"; do
  echo "== cfg5 $v"
  python bench.py --config cfg5 --steps 2 --warmup 1 --no-sides --no-cpu-baseline $v 2> gpurun_out/r05_d_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms',d['ms_per_step'],d['kernel_ms'],'chunks',d['config']['chunks_per_step'],'swapped',d['swapped_seeds'],'err paths',d['paths_with_error_status'],d['steps_per_path'])
"
  tail -2 gpurun_out/r05_d_err.txt
done
