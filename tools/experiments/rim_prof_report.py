"""Medians of the RIM_PROF phase clocks (tools/experiments/rim_prof.py output): first round of workgroups vs later."""
import re, statistics as st, sys
rows = []
for l in open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/rim_prof.txt"):
    m = re.match(r"RIM wg (\d+) wave (\d+) hw (\w+) real (\d+) stage (\d+) bar (\d+) k (\d+) epi (\d+)", l)
    if m:
        rows.append(tuple(int(x, 16) if i == 2 else int(x) for i, x in enumerate(m.groups())))
rows.sort(key=lambda r: r[3])
n = len(rows) // 3
L = rows[-n:]
t0 = min(r[3] for r in L)
first = [r for r in L if (r[3] - t0) < 300]
later = [r for r in L if (r[3] - t0) >= 300]
for nm, S in (("first round", first), ("later", later)):
    if not S:
        continue
    print(nm, len(S), {k: int(st.median([r[i] for r in S])) for k, i in (("stage", 4), ("bar", 5), ("k", 6), ("epi", 7))},
          "total", int(st.median([sum(r[4:8]) for r in S])))
starts = sorted(set(round((r[3] - t0) / 100.0, 1) for r in L))
print("start times (us):", starts[:40])
