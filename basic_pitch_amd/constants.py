"""Geometry of the hot path, under the names the reference's callers import (basic_pitch/constants.py:25-71).

Everything follows from five numbers — 22,050 Hz, a hop of 256 samples, 2-second windows, 88 semitones from 27.5 Hz,
1 (notes) or 3 (contours) bins per semitone — which are also baked into the kernels (csrc/bp_common.h) and the C ABI
(include/basic_pitch_amd.h: BP_AUDIO_N_SAMPLES, BP_N_FRAMES, BP_N_NOTE_BINS, BP_N_CONTOUR_BINS).
"""
import numpy as np

# ---- audio
AUDIO_SAMPLE_RATE, AUDIO_N_CHANNELS, AUDIO_WINDOW_LENGTH = 22050, 1, 2  # Hz, mono, seconds per window
FFT_HOP = 256
AUDIO_N_SAMPLES = AUDIO_WINDOW_LENGTH * AUDIO_SAMPLE_RATE - FFT_HOP  # a window is one hop short of 2 s
assert AUDIO_N_SAMPLES == 43844

# ---- time axis of the posteriorgrams
ANNOTATIONS_FPS = AUDIO_SAMPLE_RATE // FFT_HOP
ANNOT_N_FRAMES = AUDIO_WINDOW_LENGTH * ANNOTATIONS_FPS
ANNOTATION_HOP = 1.0 / ANNOTATIONS_FPS
assert (ANNOTATIONS_FPS, ANNOT_N_FRAMES) == (86, 172)

# ---- frequency axis: the piano range, geometric bins
SEMITONES_PER_OCTAVE = 12
ANNOTATIONS_N_SEMITONES, ANNOTATIONS_BASE_FREQUENCY = 88, 27.5
NOTES_BINS_PER_SEMITONE, CONTOURS_BINS_PER_SEMITONE = 1, 3
N_FREQ_BINS_NOTES = NOTES_BINS_PER_SEMITONE * ANNOTATIONS_N_SEMITONES
N_FREQ_BINS_CONTOURS = CONTOURS_BINS_PER_SEMITONE * ANNOTATIONS_N_SEMITONES
assert (N_FREQ_BINS_NOTES, N_FREQ_BINS_CONTOURS) == (88, 264)


def _geometric_bins(per_semitone: int) -> np.ndarray:
    # ratio first, then its powers: the note decoder compares against these values, so the arithmetic (not just the
    # mathematics) is the reference's (constants.py:60-63)
    ratio = 2.0 ** (1.0 / (SEMITONES_PER_OCTAVE * per_semitone))
    return ANNOTATIONS_BASE_FREQUENCY * ratio ** np.arange(per_semitone * ANNOTATIONS_N_SEMITONES)


FREQ_BINS_NOTES = _geometric_bins(NOTES_BINS_PER_SEMITONE)
FREQ_BINS_CONTOURS = _geometric_bins(CONTOURS_BINS_PER_SEMITONE)
