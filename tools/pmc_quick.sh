#!/bin/bash
# quick per-kernel SQ counters for one kernel-name pattern ($1); extra env via $2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
env $2 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pmc_$$ -o b -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustained-s 0 --no-exact-f32 --no-config-extras > /dev/null 2>&1
python - "$1" /tmp/pmc_$$ <<'PY'
import csv, sys, glob, collections
pat, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: "%.3g" % (sum(x) / len(x)) for c, x in sorted(v.items())})
PY
