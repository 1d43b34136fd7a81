#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_file_pipeline.py tests/test_note_decode.py -x -q 2>&1 | tail -3
for l in 2 3 4; do for t in 0 12 24; do
python bench.py --workload files --files 256 --native --lanes $l --native-threads $t 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('native lanes $l threads $t: files/s %.1f  events %d' % (d['value'], d['config']['note_events']))"
done; done
python bench.py --workload files --files 256 --native > gpurun_out/bench_files_native.json 2>/dev/null; cat gpurun_out/bench_files_native.json | cut -c1-400
