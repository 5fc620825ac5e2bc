#!/bin/bash
# round 6: the whole GPU suite with every context poisoning its workspaces and batch arrays before use (batotp_hip_set_poison)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
( BATOTP_TEST_POISON=1 timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -60 ) > gpurun_out/r06_e_poison_suite.log 2>&1
tail -60 gpurun_out/r06_e_poison_suite.log
