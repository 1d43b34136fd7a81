#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "fused_branch or batch_invariance or onset_march or bf16 or end_to_end" 2>&1 | tail -3
for i in 1 2 3; do
echo "== default"; tools/ab_run.sh
done
