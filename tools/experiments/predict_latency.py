"""Wall time of the calls a user of the reference makes: Model(), predict(path) on the reference's 10-second clip and on a
3-minute 44.1 kHz stereo file, with a fresh and with a reused Model."""
import os, sys, time, wave
import numpy as np
sys.path.insert(0, ".")
t0 = time.perf_counter()
from basic_pitch_amd import inference as inf
print("import %.3f s" % (time.perf_counter() - t0))
clip = os.path.join("tests", "golden", "vocadito_10.wav")
rng = np.random.default_rng(7)
n = 180 * 44100
t = np.arange(n) / 44100.0
x = 0.3 * np.sin(2 * np.pi * 220.0 * t) * (np.sin(2 * np.pi * 1.5 * t) > 0) + 0.01 * rng.standard_normal(n)
pcm = (np.clip(np.stack([x, x[::-1]], 1), -1, 1) * 32767).astype("<i2")
long = "/tmp/bp_long.wav"
with wave.open(long, "wb") as w:
    w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
for k in range(3):
    t0 = time.perf_counter(); m = inf.Model(); t1 = time.perf_counter()
    print("Model() %.1f ms" % ((t1 - t0) * 1e3))
    for path in (clip, long):
        for rep in range(2):
            t0 = time.perf_counter(); out, midi, ev = inf.predict(path, m); t1 = time.perf_counter()
            print("  predict(%s, model) %.1f ms  (%d events)" % (os.path.basename(path), (t1 - t0) * 1e3, len(ev)))
    m.close()
t0 = time.perf_counter(); inf.predict(clip); print("predict(clip) with the default model path %.1f ms" % ((time.perf_counter() - t0) * 1e3))
os.remove(long)
