#!/bin/bash
# A/B of builds of the library on the as-worded configurations (k_sweep1): the in-tree library against
# batotp_amd/csrc/libbatotp_hip_<variant>.so (built with other -D switches); kernel times from bench.py's JSON line.
# usage: tools/ab_sweep1.sh "<variant> ..." [configs...]      (run on the GPU box from the repository root)
set -u
variants=$1; shift
cfgs=${@:-cfg2 cfg3 cfg4}
mkdir -p gpurun_out
run() {
  for c in $cfgs; do
    python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-sides 2>/dev/null | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_ms']; print('$1', '$c', 'ms_per_step %.1f' % d['ms_per_step'], 'pre %.2f k3 %.2f rev %.1f fwd %.1f' % (k['precompute'],k['pointwise_mvc'],k['sweep_rev'],k['sweep_fwd']), 'us/step', d.get('us_per_integration_step'), 'err', d['paths_with_error_status'])"
  done
}
run base
cp batotp_amd/csrc/libbatotp_hip.so /tmp/base.so
for v in $variants; do
  cp batotp_amd/csrc/libbatotp_hip_${v}.so batotp_amd/csrc/libbatotp_hip.so
  run $v
done
cp /tmp/base.so batotp_amd/csrc/libbatotp_hip.so
