"""Dev-time loader for the UNMODIFIED reference `basic_pitch/note_creation.py` (needs /root/reference).

`import basic_pitch` raises NameError here (no TF / CoreML / TFLite / onnxruntime: `basic_pitch/__init__.py:81-95`) and
`note_creation.py` imports four third-party modules that are not installable (`mir_eval`, `librosa`, `resampy`,
`pretty_midi`).  This module puts five tiny stand-ins on `sys.modules` and loads the reference's
`constants.py` and `note_creation.py` *from their files, unchanged*, under an empty `basic_pitch` package object.
The stand-ins cover exactly the calls `note_creation.py` makes, with the closed forms those libraries document:

  librosa.midi_to_hz(m)               440 * 2 ** ((m - 69) / 12)
  librosa.hz_to_midi(f)               12 * (log2(f) - log2(440)) + 69
  librosa.core.frames_to_time(f, sr, hop_length)   asanyarray(f) * hop_length (as int) / float(sr)
  librosa.core.cqt_frequencies        only used by sonification (never called here)
  pretty_midi.{PrettyMIDI, Instrument, Note, PitchBend, instrument_name_to_program}   attribute containers
  mir_eval.sonify, resampy            empty (sonification only)

Used by tools/make_note_fixtures.py; nothing under basic_pitch_amd/ or tests/ imports it.
"""
from __future__ import annotations

import importlib.util
import sys
import types

import numpy as np

REF_PKG = "/root/reference/basic_pitch"


def _mod(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install_stubs() -> None:
    librosa = _mod("librosa")
    librosa.midi_to_hz = lambda notes: 440.0 * (2.0 ** ((np.asanyarray(notes) - 69.0) / 12.0))
    librosa.hz_to_midi = lambda f: 12 * (np.log2(np.asanyarray(f)) - np.log2(440.0)) + 69
    core = _mod("librosa.core")
    core.frames_to_time = lambda frames, sr=22050, hop_length=512: (
        (np.asanyarray(frames) * hop_length).astype(int) / float(sr)
    )
    core.cqt_frequencies = None
    librosa.core = core

    pm = _mod("pretty_midi")

    class Note:
        def __init__(self, velocity, pitch, start, end):
            self.velocity, self.pitch, self.start, self.end = velocity, pitch, start, end

    class PitchBend:
        def __init__(self, pitch, time):
            self.pitch, self.time = pitch, time

    class Instrument:
        def __init__(self, program, is_drum=False, name=""):
            self.program, self.is_drum, self.name = program, is_drum, name
            self.notes, self.pitch_bends, self.control_changes = [], [], []

    class PrettyMIDI:
        def __init__(self, midi_file=None, resolution=220, initial_tempo=120.0):
            self.resolution, self.initial_tempo, self.instruments = resolution, initial_tempo, []

    pm.Note, pm.PitchBend, pm.Instrument, pm.PrettyMIDI = Note, PitchBend, Instrument, PrettyMIDI
    pm.instrument_name_to_program = lambda name: {"Electric Piano 1": 4}[name]  # General MIDI program list, 0-based

    me = _mod("mir_eval")
    me.sonify = _mod("mir_eval.sonify")
    _mod("resampy")


def load_reference_note_creation() -> types.ModuleType:
    """The reference's note_creation module object, code unmodified."""
    install_stubs()
    pkg = _mod("basic_pitch")
    pkg.__path__ = [REF_PKG]
    for name in ("constants", "note_creation"):
        spec = importlib.util.spec_from_file_location(f"basic_pitch.{name}", f"{REF_PKG}/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"basic_pitch.{name}"] = mod
        spec.loader.exec_module(mod)
        setattr(pkg, name, mod)
    return sys.modules["basic_pitch.note_creation"]
