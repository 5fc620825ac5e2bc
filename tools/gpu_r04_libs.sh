#!/bin/bash
# A/B of library builds (batotp_amd/csrc/libbatotp_hip_<name>.so, other -D switches) on the reduced headline batch
#   usage: tools/gpu_r04_libs.sh "<name> ..." [variants]
set -u
mkdir -p gpurun_out
V=${2:-1:4:8:8:8}
for n in $1; do
  lib=batotp_amd/csrc/libbatotp_hip_$n.so
  [ "$n" = "base" ] && lib=batotp_amd/csrc/libbatotp_hip.so
  echo "== $n"
  timeout 900 python tools/run_hotpath.py --lib $lib --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 1 --variants "$V" 2>&1 | grep variant
done 2>&1 | tee gpurun_out/r04_libs.log
