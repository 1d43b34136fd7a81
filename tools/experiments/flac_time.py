"""Kernel times of the device FLAC decoder on a three-minute stereo file (tools/flac_synth.c makes it): run under
rocprofv3 --kernel-trace --stats, or alone for the wall time of bp_flac_decode_device / bp_infer_flac."""
import os, subprocess, sys, tempfile, time, wave
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
subprocess.run([exe, os.path.join(d, "a.wav"), os.path.join(d, "a.flac")], check=True)
data = open(os.path.join(d, "a.flac"), "rb").read()
m = Model(max_windows=128)
got, _ = m.flac_decode_device(data)
assert np.array_equal(got, pcm.astype(np.int32))
n_frames = C = None
import ctypes as C
nf = C.c_int64()
for _ in range(3):
    t0 = time.perf_counter()
    rc = m._lib.bp_flac_decode_device(m._handle, data, len(data), None, 0, C.byref(nf))
    t1 = time.perf_counter()
    assert rc == 0
    print("bp_flac_decode_device (H2D of %.1f MB + scan + chain + decode + finalize): %.2f ms" % (len(data) / 1e6, (t1 - t0) * 1e3))
for _ in range(3):
    t0 = time.perf_counter()
    m.predict_flac(data)
    print("predict_flac: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
