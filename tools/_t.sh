#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "fused_branch or batch_invariance or end_to_end or edge_cases or track_path" 2>&1 | tail -3
for i in 1 2; do
echo "== default (dma, 5 wps)"; tools/ab_run.sh
echo "== classic"; BP_CONV2=classic tools/ab_run.sh
echo "== dma 6 wps"; BP_CONV2_WPS=6 tools/ab_run.sh
echo "== dma 4 wps"; BP_CONV2_WPS=4 tools/ab_run.sh
echo "== dma 3 wps"; BP_CONV2_WPS=3 tools/ab_run.sh
done
