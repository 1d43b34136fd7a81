#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "fused_branch or batch_invariance or end_to_end or edge_cases or track_path or multi_track or ort_shim" 2>&1 | tail -3
python - <<'PY'
import json, sys, time
sys.path.insert(0, '.')
import torch, numpy as np
import bench
print(json.dumps({k: round(v['windows_per_s']) for k, v in bench.config_extras(torch, 0).items()}))
print(bench.seam_b1_host())
PY
