"""Medians of the RIM_PROF phase clocks (tools/experiments/rim_prof.py output, built with tools/build_variant.sh rimprof
conv_contour_rim.hip -DRIM_PROF): workgroups of the first round vs later ones, and when the sampled workgroups started."""
import re, statistics as st, sys

rows = []
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rim_prof.txt"):
    m = re.match(r"RIMQ wg (\d+) wave (\d+) real (\d+) stage (\d+) bar (\d+) k1 (\d+) epi (\d+) k2 (\d+) red (\d+)", l)
    if m:
        rows.append(tuple(int(x) for x in m.groups()))
rows.sort(key=lambda r: r[2])
n = len(rows) // 3  # the tool runs three steps; the last one is steady state
L = rows[-n:]
t0 = min(r[2] for r in L)
names = ["stage", "bar", "k1", "epi", "k2", "red"]
for nm, S in (("first round", [r for r in L if r[2] - t0 < 300]), ("later", [r for r in L if r[2] - t0 >= 300])):
    if S:
        print(nm, len(S), {k: int(st.median([r[3 + i] for r in S])) for i, k in enumerate(names)},
              "total", int(st.median([sum(r[3:]) for r in S])))
print("start of the sampled workgroups (us):", sorted(set((r[0], round((r[2] - t0) / 100, 1)) for r in L)))
