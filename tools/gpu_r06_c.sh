#!/bin/bash
# round 6, third GPU pass: the default bench line of the new library, in-kernel sections of the headline batch, the new tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
( time python bench.py > gpurun_out/r06_c_bench.json 2> gpurun_out/r06_c_bench.err ) 2> gpurun_out/r06_c_bench.time
tail -3 gpurun_out/r06_c_bench.err; cat gpurun_out/r06_c_bench.time
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_c_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["kernel_ms"], "gate", d.get("flat_loop_gate",{}).get("status"))
print("cross", d.get("nested_loop_cross_check"))
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","cores","paths_compared","inputs_compared","inputs_identical_to_oracle_resampler")}, d.get("traversal_time_err_s"), d.get("step_count_mismatches"))
for k,v in d.get("as_worded",{}).items():
    print(k, v.get("ms_per_step"), v.get("kernel_ms"), "vs_cpu", v.get("vs_cpu_baseline"), "err", v.get("error"), "bad", v.get("paths_with_error_status"), "swapped", v.get("swapped_seeds"), v.get("slowest_over_mean_path"))
print("resample", d.get("resample")); print("output", d.get("output_stage"))
PY
timeout 900 python tools/sweep8_sections.py --lib batotp_amd/csrc/libbatotp_hip_s8prof.so --paths 16384 > gpurun_out/r06_c_sections.txt 2>&1
cat gpurun_out/r06_c_sections.txt | head -60
( timeout 1500 python -m pytest tests/test_gpu_resample.py tests/test_gpu_output.py tests/test_gpu_zz_as_worded.py -x -q -m gpu -k "block_upload or segment_cursor or as_worded or golden_knots" 2>&1 | tail -8 ) > gpurun_out/r06_c_tests.log 2>&1
tail -5 gpurun_out/r06_c_tests.log
