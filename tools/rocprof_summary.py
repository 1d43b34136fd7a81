#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2, rocpd SQLite) result into the small text summaries kept in profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/bench_results.db profiles/r01_kernel_stats.md "<command>"

Kernel-trace runs: per-kernel calls / total / mean / share (the `top_kernels` view = `--stats`).
PMC runs: per-kernel mean of every collected counter.
"""
from __future__ import annotations

import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)  # drop the argument list
    name = name.replace("void ", "")
    return name if len(name) <= 90 else name[:87] + "..."


def main() -> None:
    db_path, out_path = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    lines = [f"# rocprofv3 summary", "", f"source: `{db_path}`", ""]
    if cmd:
        lines += [f"command: `{cmd}`", ""]
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    if rows:
        lines += ["## kernel trace (`--kernel-trace --stats`), durations in microseconds", ""]
        lines += ["| kernel | calls | total us | mean us | % |", "|---|---:|---:|---:|---:|"]
        for name, calls, total, avg, pct in rows:
            if pct < 0.05:
                continue
            lines.append(f"| `{short(name)}` | {calls} | {total:.1f} | {avg:.2f} | {pct:.2f} |")
        lines.append("")
    try:
        pmc = list(
            cur.execute(
                "select k.name, p.counter_name, avg(p.value), count(*) from pmc_events p "
                "join kernels k on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name"
            )
        )
    except sqlite3.Error:
        try:
            pmc = list(
                cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name")
            )
        except sqlite3.Error:
            pmc = []
    if pmc:
        lines += ["## counters (mean per dispatch)", "", "| kernel | counter | mean | dispatches |", "|---|---|---:|---:|"]
        for name, ctr, val, cnt in pmc:
            if name.startswith("void at::") or name.startswith("__amd"):
                continue
            lines.append(f"| `{short(name)}` | {ctr} | {val:.6g} | {cnt} |")
        lines.append("")
    with open(out_path, "w") as f:
        f.write("\n".join(lines))
    print("\n".join(lines[:40]))


if __name__ == "__main__":
    main()
