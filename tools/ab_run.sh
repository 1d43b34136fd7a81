#!/bin/bash
# quick A/B on the GPU box: stage table of a short bench run; args are passed to bench.py
python bench.py --no-cpu-baseline --sustained-s 0 --no-config-extras --steps 30 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f  ms %.4f' % (d['value'], d['ms_per_step']))
print({k: round(v,4) for k,v in d['stage_ms'].items() if v})"
