#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/profile_gpu.sh r04_b 300 5 > gpurun_out/prof_r04_b.log 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
