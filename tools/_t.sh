#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04e
python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --workload files --native --files 1024 --lanes 3 2>/dev/null | tail -1 > gpurun_out/r04e/bench_files_native.json
python bench.py --workload files --native --files 1024 --lanes 3 2>/dev/null | tail -1 > gpurun_out/r04e/bench_files_native_2.json
python bench.py --workload files --files 128 2>/dev/null | tail -1 > gpurun_out/r04e/bench_files_python.json
python bench.py 2>/dev/null | tail -1 > gpurun_out/r04e/bench.json
python __graft_entry__.py smoke 2>&1 | tail -2
for f in gpurun_out/r04e/*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print({k:d[k] for k in ('value','unit','ms_per_step','worker_ms_per_file') if k in d})"; done
