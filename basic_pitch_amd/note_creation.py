"""Host-side mirror of `basic_pitch/note_creation.py` (spotify/basic-pitch v0.4.0) for the consumer of the
hot path: posteriorgrams -> note events -> MIDI.  Same names, arguments and return values; the loops run in
libbasicpitch_amd.so (`bp_notes_decode`, csrc/note_decode.cpp) instead of Python — 2 s of single-core Python
per 3-minute track in the reference.

The PrettyMIDI object is the stand-in of `basic_pitch_amd.midi` (pretty_midi is not installable here).
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict
from typing import Any, DefaultDict, Dict, List, Optional, Tuple

import numpy as np

from . import _native
from . import midi as pretty_midi
from .constants import (
    ANNOT_N_FRAMES,
    AUDIO_N_SAMPLES,
    AUDIO_SAMPLE_RATE,
    CONTOURS_BINS_PER_SEMITONE,
    FFT_HOP,
    N_FREQ_BINS_CONTOURS,
    N_FREQ_BINS_NOTES,
)

MIDI_OFFSET = 21
N_PITCH_BEND_TICKS = 8192
MAX_FREQ_IDX = 87
DEFAULT_MIN_NOTE_LEN = 11
ENERGY_TOLERANCE = 11
MAGIC_ALIGNMENT_OFFSET = 0.0018
MIDI_VELOCITY_SCALE = 127
PITCH_BEND_SCALE = 4096

NoteEvent = Tuple[float, float, int, float, Optional[List[int]]]


def sonify_midi(midi: "pretty_midi.PrettyMIDI", save_path, sr: Optional[int] = 44100) -> None:
    """note_creation.py:119-128: render the MIDI object with sine oscillators and save it as a WAV file."""
    from scipy.io import wavfile

    y = midi.synthesize(sr)
    wavfile.write(save_path, sr, y)


def _decode(
    frames: np.ndarray, onsets: np.ndarray, contours: np.ndarray, onset_thresh: float, frame_thresh: float,
    min_note_len: int, infer_onsets: bool, max_freq: Optional[float], min_freq: Optional[float],
    melodia_trick: bool, energy_tol: int, include_pitch_bends: bool,
):
    """One call into bp_notes_decode; returns the filled event / bend arrays."""
    lib = _native.load_library()
    # The reference takes any array-like (float64 arrays, sliced views, lists): coerce to what the C ABI needs.  When
    # that makes a copy of note / onset, the columns `constrain_frequency` zeroed are zeroed in the caller's array as
    # well, so the reference's in-place behaviour (note_creation.py:338-341) is kept — and nothing else is touched (a
    # float64 input keeps its float64 values); a read-only input is decoded from a copy and left alone.
    given = {"note": frames, "onset": onsets}
    frames = np.require(frames, np.float32, ["C", "W"])
    onsets = np.require(onsets, np.float32, ["C", "W"])
    contours = np.require(contours, np.float32, ["C"])
    for name, a, w in (("note", frames, N_FREQ_BINS_NOTES), ("onset", onsets, N_FREQ_BINS_NOTES), ("contour", contours, N_FREQ_BINS_CONTOURS)):
        if a.ndim != 2 or a.shape[1] != w:
            raise ValueError(f"{name}: expected an array of shape (T, {w}), got {a.shape}")
    T = frames.shape[0]
    if onsets.shape[0] != T or contours.shape[0] != T:
        raise ValueError("note, onset and contour must have the same number of frames")
    prm = _native.bp_note_params()
    lib.bp_note_params_default(C.byref(prm))
    prm.onset_threshold, prm.frame_threshold = float(onset_thresh), float(frame_thresh)
    prm.min_note_len, prm.energy_tol = int(min_note_len), int(energy_tol)
    prm.infer_onsets, prm.melodia_trick = int(bool(infer_onsets)), int(bool(melodia_trick))
    prm.include_pitch_bends = int(bool(include_pitch_bends))
    prm.min_freq_hz = float(min_freq) if min_freq is not None else 0.0
    prm.max_freq_hz = float(max_freq) if max_freq is not None else 0.0
    n_ev, n_b = C.c_int64(0), C.c_int64(0)
    cap_ev, cap_b = max(256, T // 4), max(4096, 4 * T)
    while True:
        events = (_native.bp_note_event * cap_ev)()
        bends = np.empty(cap_b, dtype=np.int32)
        rc = lib.bp_notes_decode(
            frames.ctypes.data, onsets.ctypes.data, contours.ctypes.data, T, C.byref(prm), C.addressof(events),
            cap_ev, bends.ctypes.data, cap_b, C.byref(n_ev), C.byref(n_b),
        )
        if rc == _native.BP_OK:
            for name, used in (("note", frames), ("onset", onsets)):
                orig = given[name]
                if used is not orig and isinstance(orig, np.ndarray) and orig.flags.writeable:
                    orig[:, ~used.any(axis=0) & orig.any(axis=0)] = 0
            return events, bends, n_ev.value
        if n_ev.value > cap_ev or n_b.value > cap_b:  # buffers too small: sizes were returned
            cap_ev, cap_b = max(cap_ev, n_ev.value), max(cap_b, n_b.value)
            continue
        raise ValueError(f"bp_notes_decode: {lib.bp_notes_last_error().decode(errors='replace')}")


def _note_params(onset_thresh, frame_thresh, min_note_len, infer_onsets, max_freq, min_freq, melodia_trick, energy_tol,
                 include_pitch_bends):
    prm = _native.bp_note_params()
    _native.load_library().bp_note_params_default(C.byref(prm))
    prm.onset_threshold, prm.frame_threshold = float(onset_thresh), float(frame_thresh)
    prm.min_note_len, prm.energy_tol = int(min_note_len), int(energy_tol)
    prm.infer_onsets, prm.melodia_trick = int(bool(infer_onsets)), int(bool(melodia_trick))
    prm.include_pitch_bends = int(bool(include_pitch_bends))
    prm.min_freq_hz = float(min_freq) if min_freq is not None else 0.0
    prm.max_freq_hz = float(max_freq) if max_freq is not None else 0.0
    return prm


def decode_candidates(note: np.ndarray, cand_bits: np.ndarray, bend_map: Optional[np.ndarray], prm) -> List[NoteEvent]:
    """The sequential half of note decoding from what the device extracted (`bp_note_candidates` /
    `bp_infer_pcm_raw_candidates`: the frequency-constrained note map, the bitmap of onset peaks, the pitch-bend map):
    note_creation.py:404-509 + the bends and frame times of 182-219, 346-357 -> [(start_s, end_s, pitch, amplitude, bends)]."""
    lib = _native.load_library()
    note = np.require(note, np.float32, ["C"])
    cand_bits = np.require(cand_bits, np.uint8, ["C"])
    T = note.shape[0]
    if note.ndim != 2 or note.shape[1] != N_FREQ_BINS_NOTES or cand_bits.shape != (T, 12):
        raise ValueError("expected note (T, 88) float32 and cand_bits (T, 12) uint8")
    if bend_map is not None:
        bend_map = np.require(bend_map, np.int8, ["C"])
        if bend_map.shape != (T, N_FREQ_BINS_NOTES):
            raise ValueError("expected bend_map (T, 88) int8")
    n_ev, n_b = C.c_int64(0), C.c_int64(0)
    cap_ev, cap_b = max(256, T // 4), max(4096, 4 * T)
    while True:
        events = (_native.bp_note_event * cap_ev)()
        bends = np.empty(cap_b, dtype=np.int32)
        rc = lib.bp_notes_decode_candidates(
            note.ctypes.data, cand_bits.ctypes.data, bend_map.ctypes.data if bend_map is not None else None, T,
            C.byref(prm), C.addressof(events), cap_ev, bends.ctypes.data, cap_b, C.byref(n_ev), C.byref(n_b),
        )
        if rc == _native.BP_OK:
            with_bends = bool(prm.include_pitch_bends)
            return [(float(e.start_s), float(e.end_s), int(e.pitch_midi), np.float32(e.amplitude),
                     bends[e.bend_offset : e.bend_offset + e.n_bends].tolist() if with_bends else None)
                    for e in events[: n_ev.value]]
        if n_ev.value > cap_ev or n_b.value > cap_b:
            cap_ev, cap_b = max(cap_ev, n_ev.value), max(cap_b, n_b.value)
            continue
        raise ValueError(f"bp_notes_decode_candidates: {lib.bp_notes_last_error().decode(errors='replace')}")


def output_to_notes_polyphonic(
    frames: np.ndarray, onsets: np.ndarray, onset_thresh: float, frame_thresh: float, min_note_len: int,
    infer_onsets: bool, max_freq: Optional[float], min_freq: Optional[float], melodia_trick: bool = True,
    energy_tol: int = ENERGY_TOLERANCE,
) -> List[Tuple[int, int, int, float]]:
    """note_creation.py:360-511 -> [(start_frame, end_frame, pitch_midi, amplitude)]."""
    dummy = np.zeros((frames.shape[0], N_FREQ_BINS_CONTOURS), dtype=np.float32)
    ev, _, n = _decode(frames, onsets, dummy, onset_thresh, frame_thresh, min_note_len, infer_onsets, max_freq,
                       min_freq, melodia_trick, energy_tol, include_pitch_bends=False)
    return [(ev[i].start_frame, ev[i].end_frame, ev[i].pitch_midi, np.float32(ev[i].amplitude)) for i in range(n)]


def model_frames_to_time(n_frames: int) -> np.ndarray:
    """note_creation.py:346-357."""
    original_times = (np.arange(n_frames) * FFT_HOP).astype(int) / float(AUDIO_SAMPLE_RATE)
    window_numbers = np.floor(np.arange(n_frames) / ANNOT_N_FRAMES)
    window_offset = (FFT_HOP / AUDIO_SAMPLE_RATE) * (ANNOT_N_FRAMES - (AUDIO_N_SAMPLES / FFT_HOP)) + MAGIC_ALIGNMENT_OFFSET
    return original_times - (window_offset * window_numbers)


def drop_overlapping_pitch_bends(note_events_with_pitch_bends: List[NoteEvent]) -> List[NoteEvent]:
    """note_creation.py:270-286: drop pitch bends from any notes that overlap in time with another note."""
    note_events = sorted(note_events_with_pitch_bends)
    for i in range(len(note_events) - 1):
        for j in range(i + 1, len(note_events)):
            if note_events[j][0] >= note_events[i][1]:
                break
            note_events[i] = note_events[i][:-1] + (None,)
            note_events[j] = note_events[j][:-1] + (None,)
    return note_events


def note_events_to_midi(
    note_events_with_pitch_bends: List[NoteEvent], multiple_pitch_bends: bool = False, midi_tempo: float = 120
) -> pretty_midi.PrettyMIDI:
    """note_creation.py:222-267.

    Same objects in the same order as the reference's per-note loop; the pitch bends of ALL notes go through numpy in one
    pass (a 3-minute track holds ~10 k of them, and this function runs under the GIL while the GPU waits for the next
    group of files): `np.linspace(start, end, n)` is restated as k * step + start with the last element set to `end` —
    numpy's own evaluation order (function_base.py), so the times are bit-identical."""
    mid = pretty_midi.PrettyMIDI(initial_tempo=midi_tempo)
    if not multiple_pitch_bends:
        note_events_with_pitch_bends = drop_overlapping_pitch_bends(note_events_with_pitch_bends)
    piano_program = pretty_midi.instrument_name_to_program("Electric Piano 1")
    events = note_events_with_pitch_bends
    if not events:
        return mid
    counts = [len(e[4]) if e[4] else 0 for e in events]
    total = sum(counts)
    ticks = np.zeros(0, dtype=np.int64)
    times = np.zeros(0, dtype=np.float64)
    if total:
        flat = np.fromiter((b for e, c in zip(events, counts) if c for b in e[4]), dtype=np.int64, count=total)
        ticks = np.round(flat * PITCH_BEND_SCALE / CONTOURS_BINS_PER_SEMITONE).astype(int)
        ticks[ticks > N_PITCH_BEND_TICKS - 1] = N_PITCH_BEND_TICKS - 1
        ticks[ticks < -N_PITCH_BEND_TICKS] = -N_PITCH_BEND_TICKS
        cnt = np.array([c for c in counts if c], dtype=np.int64)
        start = np.array([e[0] for e, c in zip(events, counts) if c], dtype=np.float64)
        stop = np.array([e[1] for e, c in zip(events, counts) if c], dtype=np.float64)
        first = np.cumsum(cnt) - cnt
        k = (np.arange(total) - np.repeat(first, cnt)).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            step = (stop - start) / (cnt - 1)  # n = 1: never used below (k = 0 -> start)
        step[cnt == 1] = 0.0
        times = k * np.repeat(step, cnt) + np.repeat(start, cnt)
        multi = cnt > 1
        times[(first + cnt - 1)[multi]] = stop[multi]
    n_ev = len(events)
    ev_start = np.array([e[0] for e in events], dtype=np.float64)
    ev_end = np.array([e[1] for e in events], dtype=np.float64)
    ev_pitch = np.array([e[2] for e in events], dtype=np.int64)
    ev_vel = np.array([int(np.round(MIDI_VELOCITY_SCALE * e[3])) for e in events], dtype=np.int64)
    ev_cnt = np.array(counts, dtype=np.int64)
    ev_first = np.cumsum(ev_cnt) - ev_cnt
    # one instrument, or (multiple_pitch_bends) one per note number in order of first appearance; inside an instrument
    # notes and pitch bends keep the event order
    if multiple_pitch_bends:
        groups: Dict[int, List[int]] = {}
        for i, e in enumerate(events):
            groups.setdefault(e[2], []).append(i)
        members = [np.array(ix, dtype=np.int64) for ix in groups.values()]
    else:
        members = [np.arange(n_ev)]
    for ix in members:
        c = ev_cnt[ix]
        sel = np.repeat(ev_first[ix], c) + (np.arange(int(c.sum())) - np.repeat(np.cumsum(c) - c, c))
        mid.instruments.append(pretty_midi.Instrument.from_arrays(piano_program, ev_pitch[ix], ev_vel[ix], ev_start[ix],
                                                                  ev_end[ix], ticks[sel], times[sel]))
    return mid


def model_output_to_notes(
    output: Dict[str, np.ndarray], onset_thresh: float, frame_thresh: float, infer_onsets: bool = True,
    min_note_len: int = DEFAULT_MIN_NOTE_LEN, min_freq: Optional[float] = None, max_freq: Optional[float] = None,
    include_pitch_bends: bool = True, multiple_pitch_bends: bool = False, melodia_trick: bool = True,
    midi_tempo: float = 120,
) -> Tuple[pretty_midi.PrettyMIDI, List[NoteEvent]]:
    """note_creation.py:52-116: model output -> (midi, [(start_s, end_s, pitch_midi, amplitude, bends)])."""
    frames, onsets, contours = output["note"], output["onset"], output["contour"]
    ev, bends, n = _decode(frames, onsets, contours, onset_thresh, frame_thresh, min_note_len, infer_onsets,
                           max_freq, min_freq, melodia_trick, ENERGY_TOLERANCE, include_pitch_bends)
    events: List[NoteEvent] = []
    flat_bends = bends.tolist() if (include_pitch_bends and hasattr(bends, "tolist")) else None
    for i in range(n):
        e = ev[i]
        if not include_pitch_bends:
            b = None
        elif flat_bends is not None:
            b = flat_bends[e.bend_offset : e.bend_offset + e.n_bends]
        else:
            b = [int(v) for v in bends[e.bend_offset : e.bend_offset + e.n_bends]]
        events.append((float(e.start_s), float(e.end_s), int(e.pitch_midi), np.float32(e.amplitude), b))
    return note_events_to_midi(events, multiple_pitch_bends, midi_tempo), events
