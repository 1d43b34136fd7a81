# How far the fp8 correction products move the three posteriorgrams (tools/): BP_FLAG_FP8_CORRECTIONS (opt-in) vs the default path (all-f16) on
# the same windows, max and 99.9th percentile of |difference| per map, for noise-like and tonal windows.
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_windows
from basic_pitch_amd import Model
a, b = Model(max_windows=256, fp8_corrections=True), Model(max_windows=256)  # fp8 opt-in mode, default (all-f16)
for kind, n in (("uniform", 512), ("normal", 512), ("tones", 256)):
    x = make_windows(kind, n, 123)
    pa, pb = a.predict(x), b.predict(x)
    print(kind, n, {k: ("max %.2e" % np.abs(pa[k] - pb[k]).max(), "p99.9 %.2e" % np.quantile(np.abs(pa[k] - pb[k]), 0.999)) for k in ("note", "onset", "contour")})
