#!/bin/bash
# round 5: one or two paths per wavefront of k_sweep1 for the shares of BASELINE config 5 a GPU gets at 2, 4 and 8 GPUs
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for paths in 2048 1024 512 3072; do
for v in "--ppw 1" "--ppw 2"; do
  echo "== cfg5 --paths $paths --lean --group 64 $v"
  python bench.py --config cfg5 --paths $paths --steps 2 --warmup 1 --no-sides --no-cpu-baseline --lean --group 64 $v 2> gpurun_out/r05_e_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms',round(d['ms_per_step'],1),{k:round(v,1) for k,v in d['kernel_ms'].items()},'chunks',d['config']['chunks_per_step'],'err paths',d['paths_with_error_status'])
"
done
done 2>&1 | tee gpurun_out/r05_e_ppw.log
