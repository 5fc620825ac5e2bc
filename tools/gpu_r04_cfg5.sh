#!/bin/bash
# round 4: BASELINE config 5 as worded on one GPU: every channel as pairs (the default) against coefficient rows
set -u
ulimit -c 0
mkdir -p gpurun_out
for mode in "" "--coefficient-rows"; do
  timeout 900 python bench.py --config cfg5 --steps 2 --warmup 1 --no-cpu-baseline --no-sides $mode 2>gpurun_out/cfg5_err.log | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; c=d['config']; print('cfg5', '$mode', 'layout', c['spline_layout'], 'chunks', c['chunks_per_step'], 'ms_per_step %.1f' % d['ms_per_step'], 'pre %.1f k3 %.1f rev %.1f fwd %.1f' % (k['precompute'],k['pointwise_mvc'],k['sweep_rev'],k['sweep_fwd']), 'err', d['paths_with_error_status'], 'wp/s %.3e' % d['value'])"
  tail -3 gpurun_out/cfg5_err.log
done 2>&1 | tee gpurun_out/r04_cfg5.log
