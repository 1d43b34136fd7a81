#!/usr/bin/env python
"""Parity table for the whole path (run on the GPU box): for every input family of the test suite, per window,
    |hip - fp64 oracle|,  |hip - fp32 oracle|,  |fp32 oracle - fp64 oracle|,  |C fp32 oracle - fp64 oracle|
as max-abs over the three posteriorgrams, plus where the per-window minimum of the log-power sits.  Inputs: the
uniform / normal windows the benchmark and the tests use, 32 tonal windows (tests/conftest.py make_windows("tones")),
a pure 440 Hz sine, a quiet sine, and the six windows of the reference's clip.  Writes markdown to the path given
(default gpurun_out/r02_parity.md); copy it to profiles/."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import GOLDEN, make_windows  # noqa: E402
from oracle import bp_oracle as O  # noqa: E402
from oracle import soxr_oracle as S  # noqa: E402


def main() -> None:
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r02_parity.md")
    flags = {a for a in sys.argv[2:]}
    from basic_pitch_amd import Model, audio

    W = O.load_weights()
    t = np.arange(O.AUDIO_N_SAMPLES) / 22050.0
    pcm, sr = audio.read_wav(os.path.join(GOLDEN, "vocadito_10.wav"))
    clip_windows, _ = O.window_track(S.resample(pcm[:, 0], sr))
    families = [
        ("uniform[-1,1)", make_windows("uniform", 4, 0)),
        ("normal 0.01", make_windows("normal", 4, 1)),
        ("tones (3 harmonics + 1e-3 noise)", make_windows("tones", 32, 2)),
        ("sine 440 Hz, amplitude 0.5", (0.5 * np.sin(2 * np.pi * 440.0 * t))[None].astype(np.float32)),
        ("sine 440 Hz, amplitude 0.01", (0.01 * np.sin(2 * np.pi * 440.0 * t))[None].astype(np.float32)),
        ("reference clip windows", clip_windows.astype(np.float32)),
    ]
    m = Model(max_windows=64, exact_f32_mfma="exact" in flags)
    lines = ["# Whole-path parity table (MI355X, default split-f16 path)" if "exact" not in flags else "# Whole-path parity table (exact-f32 A/B path)", "",
             "max-abs over note / onset / contour per window; `min bin` = (frame, CQT bin) of the window's log-power minimum in the fp64 oracle.", "",
             "| family | window | hip-fp64 | hip-fp32 | fp32-fp64 | C fp32-fp64 | hip<=max(1e-4, 2x fp32) | min bin |", "|---|---|---|---|---|---|---|---|"]
    summary = []
    for name, x in families:
        got = m.predict(x)
        r32 = O.forward(x, W, np.float32)
        r64 = O.forward(x, W, np.float64, intermediates=True)
        try:
            rc = O.forward_c(x, 0)
        except OSError:
            rc = None
        fam = []
        for i in range(x.shape[0]):
            d = lambda a, b: max(float(np.abs(a[k][i] - b[k][i]).max()) for k in ("note", "onset", "contour"))  # noqa: E731
            h64, h32, o = d(got, r64), d(got, r32), d(r32, r64)
            c = d(rc, r64) if rc is not None else float("nan")
            lp = r64["lp"][i]
            fr, b = np.unravel_index(int(np.argmin(lp)), lp.shape)
            ok = h64 <= max(1e-4, 2 * o)
            fam.append((h64, h32, o, c, ok))
            lines.append(f"| {name} | {i} | {h64:.2e} | {h32:.2e} | {o:.2e} | {c:.2e} | {'yes' if ok else 'NO'} | ({fr}, {b}) |")
        a = np.asarray([f[:4] for f in fam])
        summary.append((name, len(fam), a[:, 0].max(), np.median(a[:, 0]), a[:, 1].max(), a[:, 2].max(), np.median(a[:, 2]), a[:, 3].max(),
                        sum(f[4] for f in fam), int((a[:, 0] <= 1e-4).sum())))
    lines += ["", "## Summary", "", "| family | windows | hip-fp64 max | median | hip-fp32 max | fp32-fp64 max | median | C fp32-fp64 max | within max(1e-4, 2x) | within 1e-4 |",
              "|---|---|---|---|---|---|---|---|---|---|"]
    for s in summary:
        lines.append(f"| {s[0]} | {s[1]} | {s[2]:.2e} | {s[3]:.2e} | {s[4]:.2e} | {s[5]:.2e} | {s[6]:.2e} | {s[7]:.2e} | {s[8]}/{s[1]} | {s[9]}/{s[1]} |")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines[-len(summary) - 3:]))


if __name__ == "__main__":
    main()
