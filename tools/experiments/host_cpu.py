"""Host CPU time per call of the whole-track entry points, spinning vs blocking waits (one thread, one handle)."""
import os, resource, subprocess, sys, tempfile, time, wave, ctypes as C
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
nf = C.c_int64()

def cpu():
    r = resource.getrusage(resource.RUSAGE_SELF)
    return r.ru_utime, r.ru_stime

for blocking in (False, True):
    m = Model(max_windows=128, blocking_wait=blocking)
    for name, fn in (("bp_flac_decode_device", lambda: m._lib.bp_flac_decode_device(m._handle, data, len(data), None, 0, C.byref(nf))),
                     ("predict_flac", lambda: m.predict_flac(data)),
                     ("predict_pcm_raw", lambda: m.predict_pcm_raw(pcm, 1, n, 2, 44100))):
        for _ in range(3):
            fn()
        u0, s0 = cpu(); t0 = time.perf_counter()
        for _ in range(20):
            fn()
        u1, s1 = cpu(); t1 = time.perf_counter()
        print(f"blocking={blocking} {name}: wall {(t1-t0)*50:.2f} ms  user {(u1-u0)*50:.2f} ms  sys {(s1-s0)*50:.2f} ms per call", flush=True)
    del m
