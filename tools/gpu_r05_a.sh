#!/bin/bash
# round 5, first GPU pass: the new per-knot kernel, the ADVICE changes, the resampler regression, K3 timing, k_sweep8 sections
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "per_knot or hard_problems" > gpurun_out/r05_a_k3tests.log 2>&1
echo "k3 tests rc=$?" >> gpurun_out/r05_a_k3tests.log
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gated or golden or compact or pointwise or matches" > gpurun_out/r05_a_parity.log 2>&1
echo "parity subset rc=$?" >> gpurun_out/r05_a_parity.log
python -m pytest tests/test_gpu_resample.py tests/test_gpu_output.py -x -q -m gpu > gpurun_out/r05_a_resample.log 2>&1
echo "resample+output rc=$?" >> gpurun_out/r05_a_resample.log
for form in 1 0; do
  python bench.py --k3-form $form --steps 2 --warmup 1 --no-sides --no-as-worded --no-cpu-baseline > gpurun_out/r05_a_bench_k3form$form.json 2> gpurun_out/r05_a_bench_k3form$form.err
done
python tools/sweep8_sections.py --lib batotp_amd/csrc/libbatotp_hip_s8prof.so --paths 16384 > gpurun_out/r05_a_sweep8_sections.txt 2>&1
tail -3 gpurun_out/r05_a_k3tests.log gpurun_out/r05_a_parity.log gpurun_out/r05_a_resample.log
python - <<'PY'
import json
for f in (1,0):
    try:
        d=json.loads(open(f"gpurun_out/r05_a_bench_k3form{f}.json").read().strip().splitlines()[-1])
        print("k3 form",f,"value",d["value"],"ms",d["ms_per_step"],d["kernel_ms"])
    except Exception as e: print("bench",f,"failed",e)
PY
cat gpurun_out/r05_a_sweep8_sections.txt
