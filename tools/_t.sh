#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "extended or fused_branch or batch_invariance or bf16" 2>&1 | tail -40
for i in 1 2; do
echo "== ext rim GEMM"; python bench.py --ext-cqt-44k --batch 512 --steps 10 --no-cpu-baseline --sustained-s 0 --no-config-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.0f ms %.4f' % (d['value'], d['ms_per_step'])); print({k: round(v,4) for k,v in d['stage_ms'].items() if v})"
echo "== ext rim exact"; BP_RIM=exact python bench.py --ext-cqt-44k --batch 512 --steps 10 --no-cpu-baseline --sustained-s 0 --no-config-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.0f ms %.4f' % (d['value'], d['ms_per_step'])); print({k: round(v,4) for k,v in d['stage_ms'].items() if v})"
done
