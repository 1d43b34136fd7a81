#!/usr/bin/env python
"""Dev-time fixture builder: the reference's own known-answer fixtures -> tests/golden/.

The only numeric pin the reference has for the hot path is its end-to-end test
(`tests/test_inference.py:43-76`): `predict(tests/resources/vocadito_10.wav)` must match
`tests/resources/vocadito_10/model_output.npz` (posteriorgrams) and `note_events.npz` (28 events) at
atol=1e-4.  Those two files are pickled object arrays; this script re-saves them as plain numeric
arrays (no pickle needed to load) and copies the audio clip (Vocadito dataset, CC-BY-4.0, see
tests/golden/ATTRIBUTION.md).  Run here (needs /root/reference); outputs are committed.
"""
from __future__ import annotations

import os
import shutil

import numpy as np

REF = "/root/reference/tests/resources"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    shutil.copyfile(os.path.join(REF, "vocadito_10.wav"), os.path.join(OUT, "vocadito_10.wav"))
    os.chmod(os.path.join(OUT, "vocadito_10.wav"), 0o644)

    mo = np.load(os.path.join(REF, "vocadito_10", "model_output.npz"), allow_pickle=True)["arr_0"].item()
    np.savez_compressed(
        os.path.join(OUT, "vocadito_10_model_output.npz"),
        note=mo["note"].astype(np.float32),
        onset=mo["onset"].astype(np.float32),
        contour=mo["contour"].astype(np.float32),
    )

    ev = np.load(os.path.join(REF, "vocadito_10", "note_events.npz"), allow_pickle=True)["arr_0"]
    bends = [np.asarray(e[4], dtype=np.int64) for e in ev]
    np.savez_compressed(
        os.path.join(OUT, "vocadito_10_note_events.npz"),
        start_s=np.asarray([e[0] for e in ev], dtype=np.float64),
        end_s=np.asarray([e[1] for e in ev], dtype=np.float64),
        pitch=np.asarray([e[2] for e in ev], dtype=np.int64),
        amplitude=np.asarray([e[3] for e in ev], dtype=np.float32),
        bend_offsets=np.cumsum([0] + [len(b) for b in bends]).astype(np.int64),
        bend_values=np.concatenate(bends).astype(np.int64),
    )
    print("golden fixtures written to", os.path.abspath(OUT))


if __name__ == "__main__":
    main()
