#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "stage or batch_invariance or onset_march or bf16 or fp8" 2>&1 | tail -5
for i in 1 2 3; do
echo "== default"; tools/ab_run.sh
echo "== march32"; BP_ONSET=march32 tools/ab_run.sh
done
