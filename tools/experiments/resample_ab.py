"""The device resampler on fixed random signals; saves the outputs (A/B of its three kernels: run once per BP_RESAMPLE
setting against the A/B library — BASIC_PITCH_AMD_LIB=basic_pitch_amd/lib/libbasicpitch_amd_ab.so — and compare)."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from basic_pitch_amd import Model  # noqa: E402

CASES = [(44100, 2, 150001), (44100, 1, 2 * 1024 * 7), (44100, 1, 300), (88200, 1, 40000), (48000, 2, 9600), (16000, 1, 4000)]
rng = np.random.default_rng(8)
sig = [rng.uniform(-1, 1, (n, ch)).astype(np.float32) for _, ch, n in CASES]
m = Model(max_windows=8)
out = {f"c{i}": m.resample(x, sr) for i, (x, (sr, _, _)) in enumerate(zip(sig, CASES))}
m.close()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], os.environ.get("BP_RESAMPLE"), {k: v.shape for k, v in out.items()})
