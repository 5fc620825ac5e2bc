#!/bin/bash
# end of round 5: the full GPU suite, then the default bench line, on one fresh box
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 > gpurun_out/r05_m_full_gpu_tests.log 2>&1
echo "full gpu suite rc=$?" >> gpurun_out/r05_m_full_gpu_tests.log
tail -14 gpurun_out/r05_m_full_gpu_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_m_bench.json 2> gpurun_out/r05_m_bench.err
echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_m_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['kernel_ms'], d['roofline'])
print({k:v for k,v in d.get('output_stage',{}).items() if k!='what'})
print({k:v for k,v in d.get('resample',{}).items() if k!='what'})
for k,v in d.get('as_worded',{}).items(): print(k, v.get('ms_per_step'))
print(d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('inputs_identical_to_oracle_resampler'))
PY
