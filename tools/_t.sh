#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 1 4 16 64; do echo "== batch $b"; python bench.py --batch $b --steps 50 --no-cpu-baseline --sustained-s 0 --no-config-extras --no-exact-f32 --no-fp8-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('value %.0f ms %.4f' % (d['value'], d['ms_per_step'])); print({k: round(v,4) for k,v in d['stage_ms'].items() if v})"; done
