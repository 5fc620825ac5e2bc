#!/bin/bash
# round 6, final tree: the driver's own sequence on a fresh box -- pytest -x -q -m gpu, smoke(), then the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ulimit -c 0
( time timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 ) > gpurun_out/r06_q_suite.log 2>&1
tail -9 gpurun_out/r06_q_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
( time python bench.py > gpurun_out/r06_q_bench.json 2> gpurun_out/r06_q_bench.err ) 2> gpurun_out/r06_q_bench.time
cat gpurun_out/r06_q_bench.time | tail -3; tail -2 gpurun_out/r06_q_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06_q_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], d["kernel_ms"], "gate", d.get("flat_loop_gate",{}).get("status"), "roofline frac", d["roofline"]["frac"], d["roofline"].get("issue",{}).get("frac"))
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","cores","paths_compared","inputs_compared","inputs_identical_to_oracle_resampler")}, d.get("traversal_time_err_s"), d.get("step_count_mismatches"), "vs", d.get("vs_cpu_baseline"))
for k,v in d.get("as_worded",{}).items():
    print(k, v.get("ms_per_step"), v.get("kernel_ms"), "vs_cpu", v.get("vs_cpu_baseline"), "err", v.get("error"), "bad", v.get("paths_with_error_status"), "swapped", v.get("swapped_seeds"), "Terr", v.get("traversal_time_err_s"), v.get("step_count_mismatches"))
print("resample", d.get("resample",{}).get("ms"), "output", d.get("output_stage",{}).get("ms"), d.get("output_stage",{}).get("identical_to_oracle"))
PY
