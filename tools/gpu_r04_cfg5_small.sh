#!/bin/bash
# cfg 5 at reduced sizes, pairs for all channels (default) and coefficient rows: usage tools/gpu_r04_cfg5_small.sh "paths:knots ..." [modes]
set -u
ulimit -c 0
mkdir -p gpurun_out
for pk in ${1:-256:20000}; do
for mode in ${2:-pairs rows}; do
  flag=""; [ "$mode" = "rows" ] && flag="--coefficient-rows"
  timeout 600 python bench.py --config cfg5 --paths ${pk%%:*} --knots ${pk##*:} --distinct 32 --steps 1 --warmup 1 --no-cpu-baseline --no-sides $flag 2>gpurun_out/cfg5s_err.log | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; c=d['config']; print('$pk', '$mode', 'layout', c['spline_layout'], 'chunks', c['chunks_per_step'], 'ms_per_step %.1f' % d['ms_per_step'], 'pre %.1f k3 %.1f rev %.1f fwd %.1f' % (k['precompute'],k['pointwise_mvc'],k['sweep_rev'],k['sweep_fwd']), 'err', d['paths_with_error_status'], 'wp/s %.3e' % d['value'])" 2>&1 | tail -1
  grep -v amdgpu.ids gpurun_out/cfg5s_err.log | tail -3
done; done 2>&1 | tee gpurun_out/r04_cfg5s.log
