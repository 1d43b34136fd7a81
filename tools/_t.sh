#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "stage_pyramid or batch_invariance or ext or end_to_end or track_path" 2>&1 | tail -3
for i in 1 2; do
echo "== default (one launch, pf1)"; tools/ab_run.sh
echo "== pf0"; BASIC_PITCH_AMD_LIB=$PWD/basic_pitch_amd/lib/var_pf0.so tools/ab_run.sh
echo "== wide"; BP_PYR=wide tools/ab_run.sh
done
tools/kstats.sh | grep -i "decimate\|filterbank\|bench"
