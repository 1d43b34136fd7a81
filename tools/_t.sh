#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/profile_gpu.sh r04_a 20 5 > gpurun_out/prof_r04_a.log 2>&1
python bench.py > gpurun_out/r04_a_bench.json 2> gpurun_out/r04_a_bench.err
tail -2 gpurun_out/r04_a_bench.err
cut -c1-300 gpurun_out/r04_a_bench.json
