#!/bin/bash
# SQ counters of the device FLAC decoder's kernels (tools/experiments/flac_time2.py under rocprofv3 --pmc, two passes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcf_*
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_IFETCH SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_FLAT"; do
  i=$((i+1))
  env $1 rocprofv3 --pmc $set --output-format csv -d /tmp/pmcf_$i -o b -- python $ROOT/tools/experiments/flac_time2.py > /tmp/pmcf_$i.log 2>&1 || tail -3 /tmp/pmcf_$i.log
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/pmcf_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "flac_decode" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
wc = sum(acc["SQ_WAVE_CYCLES"]) / max(1, len(acc["SQ_WAVE_CYCLES"]))
for c, x in sorted(acc.items()):
    m = sum(x) / len(x)
    print("%-24s %12.4g  (%.3f of WAVE_CYCLES)" % (c, m, m / wc if wc else 0))
PY
