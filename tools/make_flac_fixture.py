#!/usr/bin/env python
"""Dev-time fixture builder: tests/golden/vocadito_10_excerpt.flac = seconds 2.0-3.0 of the reference's test clip
(tests/golden/vocadito_10.wav, 16-bit mono 44.1 kHz) coded with the test-side encoder tests/flac_writer.py.  The
decoder test compares the decoded FLAC with the WAV excerpt sample for sample (a WAV <-> FLAC pair)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import flac_writer  # noqa: E402
from basic_pitch_amd import audio  # noqa: E402

x, sr = audio.read_wav(os.path.join(ROOT, "tests", "golden", "vocadito_10.wav"))
pcm = np.round(x[2 * sr : 3 * sr] * 32768.0).astype(np.int64)
data = flac_writer.encode(pcm, sr, 16, blocksize=4096)
path = os.path.join(ROOT, "tests", "golden", "vocadito_10_excerpt.flac")
with open(path, "wb") as f:
    f.write(data)
print(path, len(data), "bytes")
