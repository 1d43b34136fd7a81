#!/usr/bin/env python
"""Dev-time fixture builder for a second clip ("add a multi-clip note-event parity test").

The reference ships a second recording, `tests/resources/vocadito_14.wav` (Vocadito, CC-BY-4.0; 12.2 s, 16-bit mono
44.1 kHz), without golden outputs.  This script stores it losslessly as FLAC (tests/flac_writer.py; also gives the
native FLAC decoder a real recording to chew on) and computes what the reference would say about it with everything
reference-derived that runs here: WAV decode -> oracle/soxr_oracle.py -> oracle/bp_oracle.py in fp64 -> the UNMODIFIED
reference note_creation.model_output_to_notes (tools/ref_stubs.py).  It also checks that the fp32 oracle decodes to the
same events (so the fixture does not sit on a threshold) and stores both, plus coarse posteriorgram statistics.
Outputs: tests/golden/vocadito_14.flac, tests/golden/vocadito_14_expected.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import flac_writer  # noqa: E402
import ref_stubs  # noqa: E402
from basic_pitch_amd import audio  # noqa: E402
from oracle import bp_oracle as O  # noqa: E402
from oracle import soxr_oracle as S  # noqa: E402

SRC = "/root/reference/tests/resources/vocadito_14.wav"


def main():
    pcm, sr = audio.read_wav(SRC)
    ints = np.round(pcm * 32768.0).astype(np.int64)
    assert np.array_equal((ints / 32768.0).astype(np.float32), pcm)
    data = flac_writer.encode(ints, sr, 16, blocksize=4096, plan=lambda fi: {"kind": "lpc" if fi % 2 else "fixed2", "escape": False})
    flac_path = os.path.join(ROOT, "tests", "golden", "vocadito_14.flac")
    with open(flac_path, "wb") as f:
        f.write(data)
    back, sr2 = audio.read_audio(flac_path)
    assert sr2 == sr and np.array_equal(back, pcm)
    y = S.resample(pcm[:, 0], sr)
    W = O.load_weights()
    nc = ref_stubs.load_reference_note_creation()
    store = {}
    events = {}
    for name, dt in (("fp64", np.float64), ("fp32", np.float32)):
        r = O.run_track(y, W, dt, batch=8)
        out = {k: np.ascontiguousarray(r[k], dtype=np.float32) for k in ("note", "onset", "contour")}
        if name == "fp64":
            for k in out:
                store[f"{k}_colmean"] = out[k].mean(axis=0)
                store[f"{k}_rowmax"] = out[k].max(axis=1)
        _, ev = nc.model_output_to_notes({k: v.copy() for k, v in out.items()}, 0.5, 0.3, min_note_len=11)
        events[name] = ev
    a, b = events["fp64"], events["fp32"]
    same = len(a) == len(b) and all(x[0] == y_[0] and x[1] == y_[1] and x[2] == y_[2] and list(x[4]) == list(y_[4]) for x, y_ in zip(a, b))
    print(len(a), "events (fp64 oracle);", len(b), "(fp32 oracle); identical discrete fields:", same)
    ev = a
    bends = [np.asarray(e[4], dtype=np.int64) for e in ev]
    store.update(
        start_s=np.asarray([e[0] for e in ev], np.float64), end_s=np.asarray([e[1] for e in ev], np.float64),
        pitch=np.asarray([e[2] for e in ev], np.int64), amplitude=np.asarray([e[3] for e in ev], np.float32),
        bend_offsets=np.cumsum([0] + [len(x) for x in bends]).astype(np.int64), bend_values=np.concatenate(bends).astype(np.int64),
        fp32_oracle_agrees=np.asarray([same]), n_samples_22k=np.asarray([len(y)]),
    )
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "vocadito_14_expected.npz"), **store)
    print("flac", len(data), "bytes")


if __name__ == "__main__":
    main()
