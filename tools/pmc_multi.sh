#!/bin/bash
# pmc_multi.sh KERNEL_PATTERN "ENV=.." COUNTER...: mean per dispatch of arbitrary counters (several rocprofv3 passes of
# at most 7 counters each) for the kernels matching the pattern, on the default bench command (3 steps)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; envs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
while [ $# -gt 0 ]; do
  grp=""; n=0
  while [ $# -gt 0 ] && [ $n -lt 7 ]; do grp="$grp $1"; shift; n=$((n+1)); done
  env $envs rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcm_$$_$i -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustained-s 0 --no-exact-f32 --no-config-extras > /dev/null 2>&1
  i=$((i+1))
done
python - "$pat" /tmp/pmcm_$$_ <<'PY'
import csv, sys, glob, collections
pat, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k)
    for c, x in sorted(v.items()):
        print("   %-32s %.4g" % (c, sum(x) / len(x)))
PY
