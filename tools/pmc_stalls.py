#!/usr/bin/env python
"""Where the waves' cycles go: two extra rocprofv3 --pmc passes (tools/_stalls.sh) into a markdown table.

    python tools/pmc_stalls.py gpurun_out/prof_X/st1/bench_counter_collection.csv gpurun_out/prof_X/st2/bench_counter_collection.csv

Per kernel, means per dispatch.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves
(MI355X_MICROARCH.md); the table gives them as fractions of SQ_WAVE_CYCLES.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles of the
matrix pipe (summed over SIMDs), SQ_BUSY_CU_CYCLES cycles of busy CUs: their ratio / 4 SIMDs is the matrix pipe's duty."""
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_report import counters  # noqa: E402

acc = {}
for p in sys.argv[1:]:
    for k, d in counters(p).items():
        acc.setdefault(k, {}).update(d)
cols = ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
        "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"]
print("| kernel | WAVE_CYCLES | " + " | ".join(c.replace("SQ_", "") for c in cols) + " | MFMA_BUSY / (4 x BUSY_CU) | MFMA_COEXEC / MFMA_BUSY |")
print("|---|---:|" + "---:|" * (len(cols) + 2))
for k, d in sorted(acc.items()):
    wc = d.get("SQ_WAVE_CYCLES", 0.0)
    if wc <= 0:
        continue
    row = [f"{d.get(c, float('nan')) / wc:.2f}" for c in cols]
    busy = d.get("SQ_BUSY_CU_CYCLES", 0.0)
    mf = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    co = d.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0.0)
    row.append(f"{mf / (4 * busy):.2f}" if busy else "-")
    row.append(f"{co / mf:.2f}" if mf else "-")
    print(f"| `{k}` | {wc:.3g} | " + " | ".join(row) + " |")
