#!/bin/bash
# round 6: A/B of the certificate's bisection replay with both next candidates formed beside the test (-DS8_SPEC_MID=1), reduced batch, same box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
: > gpurun_out/r06_p_ab.log
for lib in batotp_amd/csrc/libbatotp_hip.so batotp_amd/csrc/libbatotp_hip_specmid.so batotp_amd/csrc/libbatotp_hip.so batotp_amd/csrc/libbatotp_hip_specmid.so; do
  echo "== $lib" >> gpurun_out/r06_p_ab.log
  timeout 400 python tools/run_hotpath.py --lib $lib --workload gen7 --paths 16384 --knots 20000 --distinct 64 --group 8 --reps 3 --variants "1:4:8:8:8:3,1:4:8:8:8:3" 2>&1 | grep -E "variant" | cut -c1-150 >> gpurun_out/r06_p_ab.log
done
cat gpurun_out/r06_p_ab.log
