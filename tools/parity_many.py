# Whole-path deviation from the fp64 oracle over MANY noise-like windows (tools/): BP_FLAG_FP8_CORRECTIONS (opt-in since round 3) and
# the default (all-f16 split products), max per map.  python tools/parity_many.py [n_windows_per_family]
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_windows
from oracle import bp_oracle as O
from basic_pitch_amd import Model
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = O.load_weights()
a, b = Model(max_windows=256, fp8_corrections=True), Model(max_windows=256)  # fp8 opt-in mode, default (all-f16)
for kind in ("uniform", "normal"):
    x = make_windows(kind, n, 321)
    t0 = time.time()
    r64 = {k: [] for k in ("note", "onset", "contour")}
    for i in range(0, n, 8):
        r = O.forward(x[i : i + 8], W, np.float64)
        for k in r64: r64[k].append(r[k])
    r64 = {k: np.concatenate(v) for k, v in r64.items()}
    r32 = {k: [] for k in r64}
    for i in range(0, n, 8):
        r = O.forward(x[i : i + 8], W, np.float32)
        for k in r32: r32[k].append(r[k])
    r32 = {k: np.concatenate(v) for k, v in r32.items()}
    # per window: max over the three maps
    def per_window(p):
        return np.max([np.abs(p[k] - r64[k]).reshape(n, -1).max(1) for k in r64], axis=0)
    e32 = per_window(r32)
    pa, pb = a.predict(x), b.predict(x)
    print(kind, n, "windows, oracle %.0f s" % (time.time() - t0))
    for name, p in (("fp8 corrections", pa), ("f16 corrections", pb)):
        print("   %-16s" % name, {k: "max %.2e  p99.99 %.2e" % (np.abs(p[k] - r64[k]).max(), np.quantile(np.abs(p[k] - r64[k]), 0.9999)) for k in r64})
        e = per_window(p)
        print("   %-16s windows within 1e-4: %d/%d, within max(1e-4, 2 x fp32 oracle): %d/%d, worst ratio to that bound %.2f; fp32 oracle itself within 1e-4: %d/%d (max %.2e)"
              % ("", (e <= 1e-4).sum(), n, (e <= np.maximum(1e-4, 2 * e32)).sum(), n, (e / np.maximum(1e-4, 2 * e32)).max(), (e32 <= 1e-4).sum(), n, e32.max()))
