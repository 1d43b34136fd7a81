#!/usr/bin/env python
"""Overlap report of a traced file job (tools): python tools/files_trace_report.py <dir with rocprofv3 CSVs>

Reads *kernel_trace.csv and *memory_copy_trace.csv (rocprofv3 --kernel-trace --memory-copy-trace) and prints, over the busiest
80 % of the job's span: wall time, the union and the sum of kernel intervals, of host-to-device and of device-to-host
copy intervals, and how much of each copy direction ran concurrently with kernels / with the other direction."""
import csv
import glob
import sys


def load(path, kind):
    rows = []
    for f in glob.glob(path + "/**/*" + kind + ".csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append(r)
    return rows


def union(iv):
    iv = sorted(iv)
    out, tot = [], 0
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out, sum(b - a for a, b in out)


def inter(u1, u2):
    i = j = tot = 0
    while i < len(u1) and j < len(u2):
        a, b = max(u1[i][0], u2[j][0]), min(u1[i][1], u2[j][1])
        if b > a:
            tot += b - a
        if u1[i][1] < u2[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    d = sys.argv[1]
    k = load(d, "kernel_trace")
    c = load(d, "memory_copy_trace")
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in k]
    cs = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", "")) for r in c]
    t0, t1 = min(a for a, _, _ in ks), max(b for _, b, _ in ks)
    lo, hi = t0 + (t1 - t0) // 10, t1 - (t1 - t0) // 10
    clip = lambda iv: [(max(a, lo), min(b, hi)) for a, b in iv if b > lo and a < hi]
    ku, ksum = union(clip([(a, b) for a, b, _ in ks]))
    h2d = clip([(a, b) for a, b, dr in cs if "HOST_TO_DEVICE" in dr.upper() or dr.upper().startswith("H2D")])
    d2h = clip([(a, b) for a, b, dr in cs if "DEVICE_TO_HOST" in dr.upper() or dr.upper().startswith("D2H")])
    hu, hsum = union(h2d)
    du, dsum = union(d2h)
    span = hi - lo
    ms = lambda x: f"{x / 1e6:9.2f} ms"
    print(f"span (middle 80 %)            {ms(span)}")
    print(f"kernels: union {ms(ksum)} ({100 * ksum / span:.0f} % of span), sum of durations {ms(sum(b - a for a, b in clip([(a, b) for a, b, _ in ks])))}")
    print(f"H2D:     union {ms(hsum)} ({100 * hsum / span:.0f} %), {len(h2d)} copies, sum {ms(sum(b - a for a, b in h2d))}")
    print(f"D2H:     union {ms(dsum)} ({100 * dsum / span:.0f} %), {len(d2h)} copies, sum {ms(sum(b - a for a, b in d2h))}")
    print(f"H2D under kernels {ms(inter(hu, ku))}   D2H under kernels {ms(inter(du, ku))}   H2D with D2H {ms(inter(hu, du))}")
    allu, alls = union([tuple(x) for x in ku] + [tuple(x) for x in hu] + [tuple(x) for x in du])
    print(f"anything busy {ms(alls)} ({100 * alls / span:.0f} % of span): idle {ms(span - alls)}")
    by = {}
    for a, b, n in ks:
        if b > lo and a < hi:
            key = n.split("(")[0][-48:]
            by.setdefault(key, [0, 0])
            by[key][0] += 1
            by[key][1] += b - a
    for key, (n, tot) in sorted(by.items(), key=lambda kv: -kv[1][1])[:24]:
        print(f"  {key:48s} calls {n:6d}  avg {tot / n / 1e3:8.1f} us  total {tot / 1e6:8.2f} ms")
    dirs = {}
    for a, b, dr in cs:
        dirs.setdefault(dr, [0, 0])
        dirs[dr][0] += 1
        dirs[dr][1] += b - a
    print("copy directions:", {k_: (v[0], round(v[1] / 1e6, 2)) for k_, v in dirs.items()})


if __name__ == "__main__":
    main()
