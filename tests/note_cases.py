"""Inputs and parameter sets of the note-decoding fixtures (shared by tools/make_note_fixtures.py, which runs the
UNMODIFIED reference on them, and tests/test_note_decode.py, which runs the C++ decoder and the numpy restatement)."""
import hashlib
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def synthetic(T, seed, density=0.02):
    """Smooth random posteriorgrams with note-like ridges (so onsets, long notes and melodia leftovers exist)."""
    rng = np.random.default_rng(seed)
    note = rng.uniform(0, 0.25, (T, 88)).astype(np.float32)
    onset = rng.uniform(0, 0.3, (T, 88)).astype(np.float32)
    contour = rng.uniform(0, 0.2, (T, 264)).astype(np.float32)
    for _ in range(max(1, int(T * density))):
        f = int(rng.integers(0, 88))
        t0 = int(rng.integers(0, max(1, T - 5)))
        ln = int(rng.integers(3, 60))
        t1 = min(T, t0 + ln)
        amp = rng.uniform(0.31, 0.95)
        note[t0:t1, f] = (amp + rng.normal(0, 0.03, t1 - t0)).clip(0, 1).astype(np.float32)
        if rng.random() < 0.7:
            onset[t0, f] = np.float32(rng.uniform(0.45, 0.99))
        c = 3 * f + int(rng.integers(-1, 2))
        contour[t0:t1, max(0, c - 1) : min(264, c + 2)] += np.float32(0.6)
    return {"note": note, "onset": onset, "contour": contour.clip(0, 1)}


def golden_output():
    mo = np.load(os.path.join(GOLDEN, "vocadito_10_model_output.npz"))
    return {k: np.ascontiguousarray(mo[k]).copy() for k in ("note", "onset", "contour")}


def digest(out):
    h = hashlib.sha256()
    for k in ("note", "onset", "contour"):
        h.update(np.ascontiguousarray(out[k], dtype=np.float32).tobytes())
    return h.hexdigest()


_BASE = dict(onset_thresh=0.5, frame_thresh=0.3, min_note_len=11)

# name -> (input builder, keyword arguments of model_output_to_notes beyond _BASE)
CASES = {
    "syn_default": (lambda: synthetic(787, 1), {}),
    "syn_no_melodia": (lambda: synthetic(1500, 2), {"melodia_trick": False}),
    "syn_freq_limits": (lambda: synthetic(1200, 3), {"min_freq": 80.0, "max_freq": 1500.0}),
    "syn_no_infer_onsets": (lambda: synthetic(900, 4), {"infer_onsets": False, "min_note_len": 5}),
    "syn_no_bends": (lambda: synthetic(600, 5), {"include_pitch_bends": False}),
    "syn_thresholds": (lambda: synthetic(3000, 6), {"onset_thresh": 0.6, "frame_thresh": 0.25}),
    "syn_multi_bends": (lambda: synthetic(700, 7, density=0.06), {"multiple_pitch_bends": True}),
    "syn_dense_overlaps": (lambda: synthetic(500, 8, density=0.12), {}),
    "clip_default": (golden_output, {}),
    "clip_freq_limits": (golden_output, {"min_freq": 120.0, "max_freq": 400.0}),
    "clip_min_freq_only": (golden_output, {"min_freq": 200.0}),
    "clip_max_freq_only": (golden_output, {"max_freq": 300.0}),
    "clip_no_melodia": (golden_output, {"melodia_trick": False}),
    "clip_no_infer_onsets": (golden_output, {"infer_onsets": False}),
    "clip_multi_bends": (golden_output, {"multiple_pitch_bends": True}),
    "clip_tempo_90": (golden_output, {"midi_tempo": 90}),
}


def case_args(name):
    build, kw = CASES[name]
    args = dict(_BASE)
    args.update(kw)
    return build(), args
