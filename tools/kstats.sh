#!/bin/bash
# per-kernel average durations of the default bench command (rocprofv3 kernel trace); extra bench args / env via $1 / $2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$$
env $2 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$$ -o b -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-s 0 --no-exact-f32 --no-config-extras --no-fp8-extra $1 > /tmp/ks_$$.log 2>&1
grep '^{' /tmp/ks_$$.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('bench', round(d['value']), 'windows/s', round(d['ms_per_step'],4), 'ms/step')"
python - /tmp/ks_$$ <<'PY'
import csv, sys, glob
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    tot = 0.0
    for r in csv.DictReader(open(f)):
        if int(r["Calls"]) >= 20:
            print("%-62s calls %5s avg %9.1f us  %5s %%" % (r["Name"][:62], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
