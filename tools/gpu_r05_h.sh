#!/bin/bash
# round 5: the RCCL path of bench.py with a world of one (BATOTP_BENCH_FORCE_DIST=1): rows all_gather + curve gather on one GPU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for cfg in cfg4 cfg5; do
BATOTP_BENCH_FORCE_DIST=1 python bench.py --config $cfg --steps 2 --warmup 1 --no-sides --no-cpu-baseline 2> gpurun_out/r05_h_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$cfg','ms',round(d['ms_per_step'],1),'gathered rows',d['gathered_rows'],'curve gather',d.get('curve_gather'))
"
tail -2 gpurun_out/r05_h_err.txt
done 2>&1 | tee gpurun_out/r05_h_force_dist.log
