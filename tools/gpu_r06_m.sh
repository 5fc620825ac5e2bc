#!/bin/bash
# round 6: a longer soak of 64 concurrent one-path processes beside a sweeping resident batch, every evaluation tracing its stages
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
timeout 3300 python tools/repro_concurrent_resample.py --rounds 12 > gpurun_out/r06_m_repro.log 2>&1
tail -20 gpurun_out/r06_m_repro.log
