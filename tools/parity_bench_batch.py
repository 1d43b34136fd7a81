# The benchmark's own batch against the fp64 oracle, window by window (tools/): bench.py's 256 uniform[-1, 1) windows (torch
# generator, seed 1234, drawn on the device) + 256 normal(0, 0.01) windows through the default path, the opt-in fp8 mode and
# the exact-f32 A/B path; the fp32 numpy/torch oracle and the C fp32 oracle beside them as two more fp32-class evaluations
# of the same graph.   python tools/parity_bench_batch.py [n]  ->  markdown table on stdout
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_windows
from oracle import bp_oracle as O
from basic_pitch_amd import Model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = O.load_weights()
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1234)
xb = (torch.rand((256, 43844), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()[:n].cpu().numpy()
fam = {"bench uniform (torch seed 1234)": xb, "normal sigma 0.01 (numpy seed 1)": make_windows("normal", n, seed=1)}
KEYS = ("note", "onset", "contour")

def sliced(f, x):
    out = {k: [] for k in KEYS}
    for i in range(0, len(x), 32):
        r = f(x[i:i + 32])
        for k in KEYS: out[k].append(r[k])
    return {k: np.concatenate(v) for k, v in out.items()}

def per_window(p, ref):
    return np.max([np.abs(p[k] - ref[k]).reshape(len(ref[k]), -1).max(1) for k in KEYS], axis=0)

paths = {"default (all-f16 split)": {}, "exact-f32 A/B path": {"exact_f32_mfma": True}}
if "_ab" in os.environ.get("BASIC_PITCH_AMD_LIB", ""):  # the fp8-corrections mode lives in the A/B library since round 6
    paths["fp8 corrections (A/B library)"] = {"fp8_corrections": True}
try:
    O.c_library(); have_c = True
except OSError:
    have_c = False
for name, x in fam.items():
    t0 = time.time()
    r64 = sliced(lambda a: O.forward(a, W, np.float64), x)
    evals = {"fp32 oracle (torch CPU)": sliced(lambda a: O.forward(a, W, np.float32), x)}
    if have_c:
        evals["fp32 oracle (C / OpenMP)"] = sliced(lambda a: O.forward_c(a), x)
    for pn, kw in paths.items():
        m = Model(max_windows=256, **kw); evals[pn] = m.predict(x); m.close()
    e = {k: per_window(v, r64) for k, v in evals.items()}
    o = e["fp32 oracle (torch CPU)"]
    print(f"\n### {name}, {len(x)} windows (oracles {time.time() - t0:.0f} s)\n")
    print("| evaluation | max | p99 | median | within 1e-4 | within max(1e-4, 2 x torch fp32 oracle) | worst window (index: value / oracle's) |")
    print("|---|---|---|---|---|---|---|")
    for k, v in e.items():
        w = int(np.argmax(v / np.maximum(1e-4, 2 * o)))
        print(f"| {k} | {v.max():.2e} | {np.quantile(v, 0.99):.2e} | {np.median(v):.2e} | {(v <= 1e-4).sum()}/{len(v)} | "
              f"{(v <= np.maximum(1e-4, 2 * o)).sum()}/{len(v)} | {w}: {v[w]:.2e} / {o[w]:.2e} |")
    d = e["default (all-f16 split)"]
    bad = np.argsort(-d)[:6]
    print("\nsix worst windows of the default path: " + ", ".join(f"#{i}: hip {d[i]:.2e}, torch-fp32 {o[i]:.2e}"
          + (f", C-fp32 {e['fp32 oracle (C / OpenMP)'][i]:.2e}" if have_c else "") + f", exact-f32 {e['exact-f32 A/B path'][i]:.2e}" for i in bad))
