#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_file_pipeline.py tests/test_gpu_parity.py -x -q -k "ingest or pipeline or raw_pcm or transcribe or golden or track or resample" 2>&1 | tail -4
for cfg in "2 0" "3 0" "4 0" "4 14"; do
set -- $cfg
echo "== lanes $1 threads $2"; python bench.py --workload files --native --files 1024 --lanes $1 --native-threads $2 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('files/s %.1f' % d['value'], {k: round(v,2) for k,v in d.get('worker_ms_per_file').items()})"
done
