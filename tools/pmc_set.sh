#!/bin/bash
# tools/pmc_set.sh KERNEL_PATTERN "COUNTER1 COUNTER2 ..." ["ENV=..."]: one --pmc pass of a short bench run, per-kernel means
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcs_$$
env ${3:-X=1} rocprofv3 --pmc $2 --output-format csv -d /tmp/pmcs_$$ -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustained-s 0 --no-exact-f32 --no-config-extras --no-fp8-extra > /tmp/pmcs_$$.log 2>&1
python - "$1" /tmp/pmcs_$$ <<'PY'
import csv, sys, glob, collections
pat, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
n = 0
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n += 1
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not n: print("no counter rows:", open(d + ".log").read()[-600:])
for k, v in acc.items():
    print(k, {c: "%.3g" % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
