#!/usr/bin/env python
"""Summarise tools/profile_gpu.sh output (rocprofv3 CSVs) into profiles/<tag>_kernel_stats.md, <tag>_pmc.md, <tag>_pmc.json.

    python tools/pmc_report.py gpurun_out/prof_e profiles/r01_e "<command>"

HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a
wide coalesced read stream, so read = 2 x FETCH_SIZE x 1024, written = WRITE_SIZE x 1024 (per dispatch means).
"""
from __future__ import annotations

import csv
import json
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    return name if len(name) <= 80 else name[:77] + "..."


def ours(name: str) -> bool:
    return name.startswith("bp::") or name.startswith("void bp::")


def counters(path):
    acc = defaultdict(lambda: defaultdict(list))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if ours(row["Kernel_Name"]):
                acc[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main() -> None:
    src, dst = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    stats = []
    with open(f"{src}/trace/bench_kernel_stats.csv", newline="") as f:
        for row in csv.DictReader(f):
            stats.append((short(row["Name"]), int(row["Calls"]), float(row["TotalDurationNs"]) / 1e3,
                          float(row["AverageNs"]) / 1e3, float(row["Percentage"])))
    lines = ["# rocprofv3 kernel trace", "", f"command: `rocprofv3 --kernel-trace --stats -- {cmd}`", "",
             "| kernel | calls | total us | mean us | % |", "|---|---:|---:|---:|---:|"]
    for n, c, tot, avg, pct in stats:
        if pct >= 0.05:
            lines.append(f"| `{n}` | {c} | {tot:.1f} | {avg:.2f} | {pct:.2f} |")
    open(dst + "_kernel_stats.md", "w").write("\n".join(lines) + "\n")

    fetch = counters(f"{src}/fetch/bench_counter_collection.csv")
    write = counters(f"{src}/write/bench_counter_collection.csv")
    sq = counters(f"{src}/sq/bench_counter_collection.csv")
    mean_us = {n: avg for n, c, tot, avg, pct in stats}
    calls = {n: c for n, c, tot, avg, pct in stats}
    steps = max(1, min(c for n, c in calls.items() if n.startswith("bp::")))  # launches of a once-per-step kernel
    out = {}
    md = ["# rocprofv3 PMC summary", "",
          f"command: `rocprofv3 --pmc <counters> -- {cmd}`; separate passes for FETCH_SIZE, WRITE_SIZE and the SQ "
          "counters (tools/profile_gpu.sh).  Values are means per dispatch.", "",
          "read = 2 x FETCH_SIZE x 1024 (gfx950 correction, MI355X_MICROARCH.md §HBM), written = WRITE_SIZE x 1024.", "",
          "| kernel | launches/step | mean us | FETCH_SIZE KiB | WRITE_SIZE KiB | HBM read MB (x2) | HBM written MB | GB/s |",
          "|---|---:|---:|---:|---:|---:|---:|---:|"]
    for n in sorted(fetch):
        fs = fetch[n].get("FETCH_SIZE", 0.0)
        ws = write.get(n, {}).get("WRITE_SIZE", 0.0)
        rd, wr = 2 * fs * 1024, ws * 1024
        us = mean_us.get(n, float("nan"))
        out[n] = {"launches_per_step": calls.get(n, 0) // steps, "mean_us": us, "hbm_read_bytes": rd, "hbm_write_bytes": wr}
        md.append(f"| `{n}` | {calls.get(n, 0) // steps} | {us:.1f} | {fs:.0f} | {ws:.0f} | {rd / 1e6:.1f} | {wr / 1e6:.1f} | "
                  f"{(rd + wr) / us / 1e3:.0f} |")
    names = ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VALU_MFMA_MOPS_F16",
             "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"]
    md += ["", "## SQ counters (mean per dispatch)", "",
           "| kernel | " + " | ".join(x.replace("SQ_", "") for x in names) + " |", "|---|" + "---:|" * len(names)]
    for n in sorted(sq):
        md.append(f"| `{n}` | " + " | ".join(f"{sq[n].get(c, 0):.3g}" for c in names) + " |")
    open(dst + "_pmc.md", "w").write("\n".join(md) + "\n")
    json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
    print("\n".join(lines[4:16]))
    print("\n".join(md[6:]))


if __name__ == "__main__":
    main()
