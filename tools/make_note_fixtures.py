#!/usr/bin/env python
"""Dev-time fixture builder: note events and MIDI object contents computed by the UNMODIFIED reference
`basic_pitch/note_creation.py` (loaded by tools/ref_stubs.py) for every case of tests/note_cases.py.

Covers the branches the reference's own known-answer test leaves out: frequency constraints
(`note_creation.py:314-343`), melodia trick off / on (452-509), `infer_onsets=False` (289-311, 395-396),
`include_pitch_bends=False` (104-107), `multiple_pitch_bends=True` and the overlap rule (222-286), `midi_tempo`.
Writes tests/golden/note_fixtures.npz (numeric arrays only, no pickle); run here (needs /root/reference), output
committed.  Each case also stores the SHA-256 of its input posteriorgrams (so the test knows it regenerated the same
input) and of the note / onset arrays after the call (`constrain_frequency` mutates them in place)."""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import note_cases  # noqa: E402
import ref_stubs  # noqa: E402


def main() -> None:
    nc = ref_stubs.load_reference_note_creation()
    store = {}
    for name in note_cases.CASES:
        out, args = note_cases.case_args(name)
        store[f"{name}/input_sha256"] = np.frombuffer(bytes.fromhex(note_cases.digest(out)), dtype=np.uint8)
        midi, events = nc.model_output_to_notes(out, **args)
        h = hashlib.sha256(out["note"].tobytes() + out["onset"].tobytes()).digest()
        store[f"{name}/mutated_sha256"] = np.frombuffer(h, dtype=np.uint8)
        store[f"{name}/start_s"] = np.asarray([e[0] for e in events], dtype=np.float64)
        store[f"{name}/end_s"] = np.asarray([e[1] for e in events], dtype=np.float64)
        store[f"{name}/pitch"] = np.asarray([e[2] for e in events], dtype=np.int64)
        store[f"{name}/amplitude"] = np.asarray([e[3] for e in events], dtype=np.float32)
        bends = [np.asarray(e[4] if e[4] is not None else [], dtype=np.int64) for e in events]
        store[f"{name}/has_bends"] = np.asarray([e[4] is not None for e in events], dtype=np.uint8)
        store[f"{name}/bend_offsets"] = np.cumsum([0] + [len(b) for b in bends]).astype(np.int64)
        store[f"{name}/bend_values"] = np.concatenate(bends + [np.zeros(0, np.int64)]).astype(np.int64)
        # the PrettyMIDI object (note_creation.py:222-267): instruments in insertion order
        store[f"{name}/midi_tempo_resolution"] = np.asarray([midi.initial_tempo, midi.resolution], dtype=np.float64)
        store[f"{name}/inst_program"] = np.asarray([i.program for i in midi.instruments], dtype=np.int64)
        store[f"{name}/inst_n_notes"] = np.asarray([len(i.notes) for i in midi.instruments], dtype=np.int64)
        store[f"{name}/inst_n_bends"] = np.asarray([len(i.pitch_bends) for i in midi.instruments], dtype=np.int64)
        notes = [n for i in midi.instruments for n in i.notes]
        pbs = [b for i in midi.instruments for b in i.pitch_bends]
        store[f"{name}/note_velocity"] = np.asarray([n.velocity for n in notes], dtype=np.int64)
        store[f"{name}/note_pitch"] = np.asarray([n.pitch for n in notes], dtype=np.int64)
        store[f"{name}/note_start"] = np.asarray([n.start for n in notes], dtype=np.float64)
        store[f"{name}/note_end"] = np.asarray([n.end for n in notes], dtype=np.float64)
        store[f"{name}/pb_pitch"] = np.asarray([b.pitch for b in pbs], dtype=np.int64)
        store[f"{name}/pb_time"] = np.asarray([b.time for b in pbs], dtype=np.float64)
        print(f"{name}: {len(events)} events, {len(midi.instruments)} instruments, {len(pbs)} pitch bends")
    path = os.path.join(ROOT, "tests", "golden", "note_fixtures.npz")
    np.savez_compressed(path, **store)
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
