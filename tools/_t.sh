#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/profile_gpu.sh r04_d 20 5 > gpurun_out/prof_r04_d.log 2>&1
python bench.py > gpurun_out/r04_d_bench.json 2>/dev/null
cut -c1-200 gpurun_out/r04_d_bench.json
