# Host-side cost of a file job (tools/): per-stage seconds per 3-minute stereo 44.1 kHz WAV file, each stage alone, then
# predict_many over 32 files.  On the GPU box: python tools/profile_files_host.py
import os, sys, time, wave, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from basic_pitch_amd import audio as A
from basic_pitch_amd import note_creation as infer
from basic_pitch_amd.inference import Model, predict_many

def timed(f, n=5):
    f(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    return (time.perf_counter() - t0) / n, r

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
    dt, (pcm, sr) = timed(lambda: A.read_audio(paths[0])); print("read_audio          %.1f ms" % (dt * 1e3), pcm.dtype, pcm.shape)
    dt, y = timed(lambda: model.resample(pcm, sr)); print("resample (H2D + k)  %.1f ms" % (dt * 1e3))
    dt, outs = timed(lambda: model.predict_tracks([y])); print("predict_tracks x1   %.1f ms" % (dt * 1e3))
    ys = [y] * 32
    dt, _ = timed(lambda: model.predict_tracks(ys), 2); print("predict_tracks x32  %.1f ms per file" % (dt * 1e3 / 32))
    out = outs[0]
    def dec():
        o = {k: v.copy() for k, v in out.items()}
        return infer.model_output_to_notes(o, onset_thresh=0.5, frame_thresh=0.3, min_note_len=11)
    dt, (midi, ev) = timed(dec); print("model_output_to_notes %.1f ms (%d events)" % (dt * 1e3, len(ev)))
    dt, _ = timed(midi.to_bytes); print("midi.to_bytes       %.1f ms" % (dt * 1e3))
    for th in (4, 16):
        t0 = time.perf_counter(); predict_many(paths, model, group=32, decode_threads=th)
        print("predict_many, %2d host threads: %.1f files/s" % (th, 32 / (time.perf_counter() - t0)))
