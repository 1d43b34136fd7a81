import os, sys, time, wave, tempfile, cProfile, pstats
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from basic_pitch_amd.inference import Model, predict_many
rng = np.random.default_rng(7)
n = int(180 * 44100); t = np.arange(n) / 44100.0
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i in range(32):
        f0 = 110.0 * 2 ** (rng.integers(0, 36) / 12.0)
        x = 0.3 * np.sin(2 * np.pi * f0 * t) * (np.sin(2 * np.pi * 1.5 * t) > 0) + 0.01 * rng.standard_normal(n)
        pcm = (np.clip(np.stack([x, x[::-1]], 1), -1, 1) * 32767).astype("<i2")
        p = os.path.join(d, f"f{i}.wav")
        with wave.open(p, "wb") as w:
            w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100); w.writeframes(pcm.tobytes())
        paths.append(p)
    model = Model(max_windows=256)
    predict_many(paths[:4], model)
    for th in (8, 16, 32, 64):
        t0 = time.perf_counter(); predict_many(paths, model, group=32, decode_threads=th); print("threads", th, "files/s", 32 / (time.perf_counter() - t0))
    for g in (8, 16):
        t0 = time.perf_counter(); predict_many(paths, model, group=g, decode_threads=32); print("group", g, "files/s", 32 / (time.perf_counter() - t0))
    pr = cProfile.Profile(); pr.enable(); predict_many(paths, model, group=32); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(6)
