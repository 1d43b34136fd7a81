#!/bin/bash
# kernel times of the device FLAC decoder: tools/experiments/flac_time2.py (synthetic 3-minute stereo file) or, with "real",
# tools/experiments/flac_time_real.py (tests/golden/vocadito_14.flac) under rocprofv3 --kernel-trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
S=flac_time2.py; [ "$1" = real ] && S=flac_time_real.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktf; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktf -o t -- python $ROOT/tools/experiments/$S > /dev/null 2>&1
python - <<'PY'
import csv,glob
for f in glob.glob('/tmp/ktf/**/t_kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'flac' in r['Name']: print('  ',r['Name'][:50], r['Calls'], round(float(r['AverageNs'])/1e3,1),'us')
PY
