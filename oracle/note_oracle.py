"""ORACLE — CPU restatement of the reference's note decoding (posteriorgrams -> note events).
TEST INFRASTRUCTURE ONLY: imported by tests/ (and nothing under basic_pitch_amd/).

Restates, in plain numpy (scipy is not needed: `argrelmax` and `gaussian` are two-liners), the functions of
`basic_pitch/note_creation.py` (spotify/basic-pitch v0.4.0) that turn the model output into note events:

  * `constrain_frequency`           note_creation.py:314-343  (+ librosa.hz_to_midi: 12*(log2(f) - log2(440)) + 69)
  * `get_infered_onsets`            note_creation.py:289-311
  * `output_to_notes_polyphonic`    note_creation.py:360-511  (incl. the "melodia trick" 452-509)
  * `midi_pitch_to_contour_bin`     note_creation.py:168-179  (librosa.midi_to_hz: 440 * 2^((p - 69) / 12))
  * `get_pitch_bends`               note_creation.py:182-219  (scipy.signal.windows.gaussian(51, std=5))
  * `model_frames_to_time`          note_creation.py:346-357  (librosa.frames_to_time: frames * hop / sr)
  * `model_output_to_notes`         note_creation.py:52-116   (without the pretty_midi object)
  * `drop_overlapping_pitch_bends`  note_creation.py:270-286

Third-party pieces the reference calls and how they are restated (none is installed here):
  scipy.signal.argrelmax(x, axis=0)  -> x[t] > x[t-1] and x[t] > x[t+1], edges never peaks (order=1, mode="clip")
  scipy.signal.windows.gaussian(M, std) -> exp(-0.5 * ((arange(M) - (M-1)/2) / std)^2)
  librosa.hz_to_midi / midi_to_hz / core.frames_to_time -> the closed forms above

Pinned by the reference's own known-answer vectors: `tests/golden/vocadito_10_model_output.npz` (the
reference's posteriorgrams) -> `tests/golden/vocadito_10_note_events.npz` (its 28 note events) — every field of
every event reproduced exactly (tests/test_note_decode.py).  dtype behaviour follows numpy >= 2 (NEP 50):
the reference's float32 / float64 mixing is restated explicitly below.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np

AUDIO_SAMPLE_RATE = 22050
FFT_HOP = 256
ANNOT_N_FRAMES = 172
AUDIO_N_SAMPLES = 43844
N_FREQ_BINS_CONTOURS = 264
CONTOURS_BINS_PER_SEMITONE = 3
ANNOTATIONS_BASE_FREQUENCY = 27.5
MIDI_OFFSET = 21
MAX_FREQ_IDX = 87
MAGIC_ALIGNMENT_OFFSET = 0.0018


def hz_to_midi(f: float) -> float:
    return 12.0 * (np.log2(f) - np.log2(440.0)) + 69.0


def midi_to_hz(p: float) -> float:
    return 440.0 * (2.0 ** ((np.asarray(p, dtype=np.float64) - 69.0) / 12.0))


def constrain_frequency(onsets, frames, max_freq, min_freq):
    """note_creation.py:314-343 (in place, like the reference)."""
    n_freqs = onsets.shape[1]
    min_freq_idx, max_freq_idx = 0, n_freqs
    if min_freq is not None:
        min_freq_idx = int(np.round(hz_to_midi(min_freq) - MIDI_OFFSET))
    if max_freq is not None:
        max_freq_idx = int(np.round(hz_to_midi(max_freq) - MIDI_OFFSET))
    onsets[:, :min_freq_idx] = 0
    frames[:, :min_freq_idx] = 0
    onsets[:, max_freq_idx:] = 0
    frames[:, max_freq_idx:] = 0
    return onsets, frames


def get_infered_onsets(onsets: np.ndarray, frames: np.ndarray, n_diff: int = 2) -> np.ndarray:
    """note_creation.py:289-311.  float64 from the zero padding onwards."""
    diffs = []
    for n in range(1, n_diff + 1):
        fa = np.concatenate([np.zeros((n, frames.shape[1])), frames])  # float64
        diffs.append(fa[n:, :] - fa[:-n, :])
    frame_diff = np.min(diffs, axis=0)
    frame_diff[frame_diff < 0] = 0
    frame_diff[:n_diff, :] = 0
    with np.errstate(invalid="ignore", divide="ignore"):
        frame_diff = np.max(onsets) * frame_diff / np.max(frame_diff)
    return np.max([onsets, frame_diff], axis=0)


def argrelmax_axis0(x: np.ndarray):
    """scipy.signal.argrelmax(x, axis=0) for order=1, mode='clip'."""
    prev = np.concatenate([x[:1], x[:-1]])
    nxt = np.concatenate([x[1:], x[-1:]])
    return np.nonzero((x > prev) & (x > nxt))


def output_to_notes_polyphonic(
    frames, onsets, onset_thresh, frame_thresh, min_note_len, infer_onsets, max_freq, min_freq,
    melodia_trick=True, energy_tol=11,
) -> List[Tuple[int, int, int, float]]:
    """note_creation.py:360-511."""
    n_frames = frames.shape[0]
    onsets, frames = constrain_frequency(onsets, frames, max_freq, min_freq)
    if infer_onsets:
        onsets = get_infered_onsets(onsets, frames)
    peak_thresh_mat = np.zeros(onsets.shape)
    peaks = argrelmax_axis0(onsets)
    peak_thresh_mat[peaks] = onsets[peaks]
    onset_idx = np.where(peak_thresh_mat >= onset_thresh)
    onset_time_idx = onset_idx[0][::-1]
    onset_freq_idx = onset_idx[1][::-1]

    remaining_energy = np.zeros(frames.shape)
    remaining_energy[:, :] = frames[:, :]
    note_events = []
    for note_start_idx, freq_idx in zip(onset_time_idx, onset_freq_idx):
        if note_start_idx >= n_frames - 1:
            continue
        i = note_start_idx + 1
        k = 0
        while i < n_frames - 1 and k < energy_tol:
            if remaining_energy[i, freq_idx] < frame_thresh:
                k += 1
            else:
                k = 0
            i += 1
        i -= k
        if i - note_start_idx <= min_note_len:
            continue
        remaining_energy[note_start_idx:i, freq_idx] = 0
        if freq_idx < MAX_FREQ_IDX:
            remaining_energy[note_start_idx:i, freq_idx + 1] = 0
        if freq_idx > 0:
            remaining_energy[note_start_idx:i, freq_idx - 1] = 0
        amplitude = np.mean(frames[note_start_idx:i, freq_idx])
        note_events.append((int(note_start_idx), int(i), int(freq_idx) + MIDI_OFFSET, amplitude))

    if melodia_trick:
        energy_shape = remaining_energy.shape
        while np.max(remaining_energy) > frame_thresh:
            i_mid, freq_idx = np.unravel_index(np.argmax(remaining_energy), energy_shape)
            remaining_energy[i_mid, freq_idx] = 0
            i = i_mid + 1
            k = 0
            while i < n_frames - 1 and k < energy_tol:
                if remaining_energy[i, freq_idx] < frame_thresh:
                    k += 1
                else:
                    k = 0
                remaining_energy[i, freq_idx] = 0
                if freq_idx < MAX_FREQ_IDX:
                    remaining_energy[i, freq_idx + 1] = 0
                if freq_idx > 0:
                    remaining_energy[i, freq_idx - 1] = 0
                i += 1
            i_end = i - 1 - k
            i = i_mid - 1
            k = 0
            while i > 0 and k < energy_tol:
                if remaining_energy[i, freq_idx] < frame_thresh:
                    k += 1
                else:
                    k = 0
                remaining_energy[i, freq_idx] = 0
                if freq_idx < MAX_FREQ_IDX:
                    remaining_energy[i, freq_idx + 1] = 0
                if freq_idx > 0:
                    remaining_energy[i, freq_idx - 1] = 0
                i -= 1
            i_start = i + 1 + k
            assert i_start >= 0 and i_end < n_frames
            if i_end - i_start <= min_note_len:
                continue
            amplitude = np.mean(frames[i_start:i_end, freq_idx])
            note_events.append((int(i_start), int(i_end), int(freq_idx) + MIDI_OFFSET, amplitude))
    return note_events


def gaussian_window(m: int, std: float) -> np.ndarray:
    n = np.arange(0, m) - (m - 1.0) / 2.0
    return np.exp(-(n**2) / (2 * std * std))


def midi_pitch_to_contour_bin(pitch_midi: int) -> float:
    pitch_hz = midi_to_hz(pitch_midi)
    return 12.0 * CONTOURS_BINS_PER_SEMITONE * np.log2(pitch_hz / ANNOTATIONS_BASE_FREQUENCY)


def get_pitch_bends(contours: np.ndarray, note_events, n_bins_tolerance: int = 25):
    """note_creation.py:182-219."""
    window_length = n_bins_tolerance * 2 + 1
    freq_gaussian = gaussian_window(window_length, std=5)
    out = []
    for start_idx, end_idx, pitch_midi, amplitude in note_events:
        freq_idx = int(np.round(midi_pitch_to_contour_bin(pitch_midi)))
        freq_start_idx = np.max([freq_idx - n_bins_tolerance, 0])
        freq_end_idx = np.min([N_FREQ_BINS_CONTOURS, freq_idx + n_bins_tolerance + 1])
        sub = contours[start_idx:end_idx, freq_start_idx:freq_end_idx] * freq_gaussian[
            np.max([0, n_bins_tolerance - freq_idx]) : window_length
            - np.max([0, freq_idx - (N_FREQ_BINS_CONTOURS - n_bins_tolerance - 1)])
        ]
        pb_shift = n_bins_tolerance - np.max([0, n_bins_tolerance - freq_idx])
        bends = list(np.argmax(sub, axis=1) - pb_shift)
        out.append((start_idx, end_idx, pitch_midi, amplitude, bends))
    return out


def model_frames_to_time(n_frames: int) -> np.ndarray:
    """note_creation.py:346-357."""
    original_times = (np.arange(n_frames) * FFT_HOP).astype(int) / float(AUDIO_SAMPLE_RATE)
    window_numbers = np.floor(np.arange(n_frames) / ANNOT_N_FRAMES)
    window_offset = (FFT_HOP / AUDIO_SAMPLE_RATE) * (ANNOT_N_FRAMES - (AUDIO_N_SAMPLES / FFT_HOP)) + MAGIC_ALIGNMENT_OFFSET
    return original_times - (window_offset * window_numbers)


def model_output_to_notes(
    output, onset_thresh, frame_thresh, infer_onsets=True, min_note_len=11, min_freq=None, max_freq=None,
    include_pitch_bends=True, melodia_trick=True,
):
    """note_creation.py:52-116 minus the PrettyMIDI object: list of (start_s, end_s, pitch, amplitude, bends)."""
    frames, onsets, contours = output["note"], output["onset"], output["contour"]
    notes = output_to_notes_polyphonic(
        frames, onsets, onset_thresh=onset_thresh, frame_thresh=frame_thresh, infer_onsets=infer_onsets,
        min_note_len=min_note_len, min_freq=min_freq, max_freq=max_freq, melodia_trick=melodia_trick,
    )
    if include_pitch_bends:
        with_bends = get_pitch_bends(contours, notes)
    else:
        with_bends = [(n[0], n[1], n[2], n[3], None) for n in notes]
    times_s = model_frames_to_time(contours.shape[0])
    return [(times_s[n[0]], times_s[n[1]], n[2], n[3], n[4]) for n in with_bends], notes


def drop_overlapping_pitch_bends(events):
    """note_creation.py:270-286."""
    note_events = sorted(events)
    for i in range(len(note_events) - 1):
        for j in range(i + 1, len(note_events)):
            if note_events[j][0] >= note_events[i][1]:
                break
            note_events[i] = note_events[i][:-1] + (None,)
            note_events[j] = note_events[j][:-1] + (None,)
    return note_events


def note_candidates(output, onset_thresh, infer_onsets=True, min_freq=None, max_freq=None, include_pitch_bends=True):
    """What the device extracts for the host's note tracker (csrc/note_device.hip, bp_note_candidates), restated with the
    functions above: (note map after constrain_frequency, bitmap [T][12] uint8 of the onset peaks that reach the
    threshold — bit f & 7 of byte f >> 3, the twelfth byte zero —, pitch-bend map [T][88] int8 or None).  note_creation.py:289-311, 314-343,
    394-402, 182-219."""
    frames = np.array(output["note"], dtype=np.float32, copy=True)
    onsets = np.array(output["onset"], dtype=np.float32, copy=True)
    contours = np.asarray(output["contour"], dtype=np.float32)
    T = frames.shape[0]
    onsets, frames = constrain_frequency(onsets, frames, max_freq, min_freq)
    on = get_infered_onsets(onsets, frames) if infer_onsets else onsets
    peak_thresh_mat = np.zeros(on.shape)
    peaks = argrelmax_axis0(on)
    peak_thresh_mat[peaks] = on[peaks]
    cand = peak_thresh_mat >= onset_thresh
    cand[-1:] = False  # a note cannot start in the last frame (note_creation.py:405-406); the device never marks it
    bits = np.packbits(np.concatenate([cand, np.zeros((T, 8), bool)], axis=1), axis=1, bitorder="little")
    assert bits.shape == (T, 12)
    bend = None
    if include_pitch_bends:
        bend = np.zeros((T, 88), np.int8)
        for b in range(88):
            ev = get_pitch_bends(contours, [(0, T, b + MIDI_OFFSET, 0.0)])
            bend[:, b] = np.asarray(ev[0][4], dtype=np.int64)
    return frames, np.ascontiguousarray(bits), bend
