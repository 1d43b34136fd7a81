"""Audio ingest for the hot path: decode -> mono -> 22.05 kHz float32.

Stands in for `librosa.load(path, sr=22050, mono=True)` at basic_pitch/inference.py:239 (librosa is
a third-party dependency of the reference and is not available here).  Decode covers PCM / float WAV
(the reference's test clips); downmix is the channel mean like librosa's `to_mono`; resampling is a
polyphase FIR (scipy.signal.resample_poly).  librosa >= 0.10 resamples with soxr_hq, which cannot be
reproduced bit-for-bit without libsoxr: posteriorgrams computed from a 44.1 kHz file therefore agree
with the reference's golden file to ~4e-3 instead of 1e-4 (note events agree exactly) — see DESIGN.md.
"""
from __future__ import annotations

import pathlib
import struct
from fractions import Fraction
from typing import Tuple, Union

import numpy as np

AUDIO_SAMPLE_RATE = 22050


def read_wav(path: Union[str, pathlib.Path]) -> Tuple[np.ndarray, int]:
    """Return (float32 samples [n, channels] in [-1, 1), sample_rate) for PCM8/16/24/32 or float WAV."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    pcm = None
    while pos + 8 <= len(data):
        cid = data[pos : pos + 4]
        size = struct.unpack_from("<I", data, pos + 4)[0]
        body = data[pos + 8 : pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _br, _ba, bits = struct.unpack_from("<HHIIHH", body, 0)
            if tag == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: real tag in the GUID
                tag = struct.unpack_from("<H", body, 24)[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1:  # integer PCM
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = (np.frombuffer(pcm, dtype="<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM bit depth {bits}")
    elif tag == 3:  # IEEE float
        x = np.frombuffer(pcm, dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV format tag {tag}")
    n = len(x) // ch
    return x[: n * ch].reshape(n, ch), sr


def to_mono(x: np.ndarray) -> np.ndarray:
    """Channel mean (librosa.to_mono semantics)."""
    return x if x.ndim == 1 else x.mean(axis=1, dtype=np.float32) if x.shape[1] > 1 else x[:, 0]


def resample(x: np.ndarray, orig_sr: int, target_sr: int = AUDIO_SAMPLE_RATE) -> np.ndarray:
    """Polyphase FIR resampling; output length ceil(n * target / orig) like librosa.resample."""
    if orig_sr == target_sr:
        return np.ascontiguousarray(x, dtype=np.float32)
    import scipy.signal

    fr = Fraction(int(target_sr), int(orig_sr))
    y = scipy.signal.resample_poly(x.astype(np.float64), fr.numerator, fr.denominator)
    n_out = int(np.ceil(len(x) * target_sr / orig_sr))
    if len(y) < n_out:
        y = np.pad(y, (0, n_out - len(y)))
    return np.ascontiguousarray(y[:n_out], dtype=np.float32)


def load(path: Union[str, pathlib.Path], sr: int = AUDIO_SAMPLE_RATE, mono: bool = True) -> Tuple[np.ndarray, int]:
    """`librosa.load(path, sr=sr, mono=True)` replacement for WAV input."""
    x, file_sr = read_wav(path)
    y = to_mono(x) if mono else x
    return resample(np.ascontiguousarray(y), file_sr, sr), sr


def get_duration(filename: Union[str, pathlib.Path]) -> float:
    x, sr = read_wav(filename)
    return x.shape[0] / float(sr)
