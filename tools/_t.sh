#!/bin/bash
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_r04_a.json 2> gpurun_out/bench_r04_a.err; tail -3 gpurun_out/bench_r04_a.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_r04_a.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'sustained', d.get('sustained',{}).get('windows_per_s'))
print(json.dumps(d['configs'], indent=1)); print(d['seam_b1_host']); print(d['roofline']['step_traffic']); print(d['stage_ms']); print(d['cpu_baseline'])
PY
python -m pytest tests/test_bench_gpu.py -x -q 2>&1 | tail -3
