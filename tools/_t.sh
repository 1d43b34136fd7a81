#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
echo "== default"; python bench.py --no-cpu-baseline --sustained-s 1 --no-config-extras --no-exact-f32 --no-fp8-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.0f ms %.4f sustained %.0f' % (d['value'], d['ms_per_step'], d['sustained']['windows_per_s']))"
echo "== rim on side stream"; BP_RIM_STREAM=1 python bench.py --no-cpu-baseline --sustained-s 1 --no-config-extras --no-exact-f32 --no-fp8-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.0f ms %.4f sustained %.0f' % (d['value'], d['ms_per_step'], d['sustained']['windows_per_s']))"
done
BP_RIM_STREAM=1 python -m pytest tests/test_gpu_parity.py -x -q -k "batch_invariance or end_to_end or track_path" 2>&1 | tail -2
