"""One case of the FLAC round-trip matrix against a debug build of the device decoder (FD_DEBUG prints per frame)."""
import os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flac_writer as FW
from basic_pitch_amd import Model
bits, ch, sr, n, bs = [int(v) for v in sys.argv[1:6]]
rng = np.random.default_rng(bits * 100 + ch)
t = np.arange(n) / sr
x = np.stack([0.4 * np.sin(2 * np.pi * 220 * (c + 1) * t) + 0.05 * rng.standard_normal(n) for c in range(ch)], 1)
full = 1 << (bits - 1)
pcm = np.clip(np.round(x * full), -full, full - 1).astype(np.int64)
pcm[100:400] = 0
pcm[1200:1500] = (pcm[1200:1500] >> 3) << 3
pcm[50] = -full
data = FW.encode(pcm, sr, bits, blocksize=bs, id3=(ch == 3))
m = Model(max_windows=16)
try:
    got, sr2 = m.flac_decode_device(data)
    print("decoded", got.shape, "equal", np.array_equal(got, pcm))
    if not np.array_equal(got, pcm):
        bad = np.argwhere(got != pcm)
        print("first differences", bad[:10].tolist())
except ValueError as e:
    print("ERROR", str(e)[:200])
