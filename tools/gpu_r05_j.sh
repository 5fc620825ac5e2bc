#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/r05_j_full_gpu_tests.log 2>&1
echo "full gpu suite rc=$?" >> gpurun_out/r05_j_full_gpu_tests.log
tail -14 gpurun_out/r05_j_full_gpu_tests.log
for cfg in cfg5 cfg3; do
python bench.py --config $cfg --steps 2 --warmup 1 --no-sides --no-cpu-baseline 2> gpurun_out/r05_j_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg','ms',round(d['ms_per_step'],1),{k:round(v,1) for k,v in d['kernel_ms'].items()})
"
done
for w in gen7 cspr; do python tools/bench_resample.py --workload $w --paths 1024 --knots 100000 2>&1 | tail -1 | cut -c1-300; done
