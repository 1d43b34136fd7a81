"""Ten 2 : 1 resamplings of a 3-minute stereo signal (run under rocprofv3 --kernel-trace --stats: the kernel's own time)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from basic_pitch_amd import Model  # noqa: E402

m = Model(max_windows=8)
pcm = np.random.default_rng(1).uniform(-1, 1, (44100 * 180, 2)).astype(np.float32)
for _ in range(10):
    y = m.resample(pcm, 44100)
print(y.shape, float(np.abs(y).max()))
m.close()
