"""Decode-kernel time of the device FLAC decoder for tool variants (no correctness check: variants skip work)."""
import os, subprocess, sys, tempfile, wave, ctypes as C
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from basic_pitch_amd import Model
d = tempfile.mkdtemp()
exe = os.path.join(d, "flac_synth")
subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "flac_synth.c"), "-lm"], check=True)
rng = np.random.default_rng(7)
n = 180 * 44100
t = np.arange(n) / 44100.0
x = 0.3 * np.sin(2 * np.pi * 220.0 * t) * (np.sin(2 * np.pi * 1.5 * t) > 0) + 0.01 * rng.standard_normal(n)
pcm = (np.clip(np.stack([x, x[::-1]], 1), -1, 1) * 32767).astype("<i2")
with wave.open(os.path.join(d, "a.wav"), "wb") as w:
    w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
subprocess.run([exe, os.path.join(d, "a.wav"), os.path.join(d, "a.flac")], check=True, stderr=subprocess.DEVNULL)
data = open(os.path.join(d, "a.flac"), "rb").read()
m = Model(max_windows=128)
nf = C.c_int64()
for _ in range(4):
    m._lib.bp_flac_decode_device(m._handle, data, len(data), None, 0, C.byref(nf))
