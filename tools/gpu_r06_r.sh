#!/bin/bash
# round 6, last GPU minutes: the soak once more with the guard's array dump armed (no oracle pass: the guard's report is the detector)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
timeout 1400 python tools/repro_concurrent_resample.py --rounds 8 --no-oracle --out gpurun_out/repro3 > gpurun_out/r06_r_repro.log 2>&1
tail -14 gpurun_out/r06_r_repro.log | cut -c1-1500
ls -la gpurun_out/repro3 2>/dev/null | head
