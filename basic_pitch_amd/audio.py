"""Audio ingest for the hot path: decode -> mono -> 22.05 kHz float32.

Stands in for `librosa.load(path, sr=22050, mono=True)` at basic_pitch/inference.py:239 (librosa is
a third-party dependency of the reference and is not available here).  Downmix is the channel mean
like librosa's `to_mono`.  Resampling follows the design of librosa's default `res_type="soxr_hq"`
(libsoxr at SOXR_HQ: linear phase, pass-band to 0.9136 of the lower Nyquist, stop-band from that
Nyquist, 126 dB, Kaiser-windowed sinc — 389 taps at the input rate for 2 : 1): with it the reference's
golden posteriorgrams for its 44.1 kHz clip are reproduced inside its own atol = 1e-4.
The same design runs on the device (csrc/audio_ingest.hip); this host copy serves the per-window
path and checks the device one.
"""
from __future__ import annotations

import pathlib
import struct
from fractions import Fraction
from typing import Tuple, Union

import numpy as np

AUDIO_SAMPLE_RATE = 22050


def wav_raw(path: Union[str, pathlib.Path]):
    """The samples of a RIFF/WAVE file as the file stores them: (uint8 view of the data chunk, format tag, bits, channels,
    sample_rate).  Tag 1 = integer PCM (8 / 16 / 24 / 32 bits), 3 = IEEE float (32 / 64)."""
    with open(path, "rb") as f:
        data = memoryview(f.read())  # chunks below are views, not copies (a 3-minute stereo file is 31 MB)
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    pcm = None
    while pos + 8 <= len(data):
        cid = bytes(data[pos : pos + 4])
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
    if not ((tag == 1 and bits in (8, 16, 24, 32)) or (tag == 3 and bits in (32, 64))):
        raise ValueError(f"{path}: unsupported PCM bit depth {bits}" if tag == 1 else f"{path}: unsupported WAV format tag {tag}")
    if ch < 1:
        raise ValueError(f"{path}: zero channels")
    return pcm, tag, bits, ch, sr


def read_wav(path: Union[str, pathlib.Path]) -> Tuple[np.ndarray, int]:
    """Return (float32 samples [n, channels] in [-1, 1), sample_rate) for PCM8/16/24/32 or float WAV."""
    pcm, tag, bits, ch, sr = wav_raw(path)
    if tag == 1:  # integer PCM; scale factors are powers of two: multiplying by the reciprocal is exact
        if bits == 8:
            x = np.frombuffer(pcm, dtype=np.uint8).astype(np.float32)
            x -= 128.0
            x *= np.float32(1.0 / 128.0)
        elif bits == 16:
            x = np.frombuffer(pcm[: len(pcm) // 2 * 2], dtype="<i2").astype(np.float32)
            x *= np.float32(1.0 / 32768.0)
        elif bits == 24:
            b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v >= 1 << 23, v - (1 << 24), v)
            x = v.astype(np.float32) / float(1 << 23)
        elif bits == 32:
            x = (np.frombuffer(pcm[: len(pcm) // 4 * 4], dtype="<i4").astype(np.float64) / float(1 << 31)).astype(np.float32)
        else:
            raise ValueError(f"{path}: unsupported PCM bit depth {bits}")
    elif tag == 3:  # IEEE float
        w = 4 if bits == 32 else 8
        x = np.frombuffer(pcm[: len(pcm) // w * w], dtype="<f4" if bits == 32 else "<f8").astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported WAV format tag {tag}")
    n = len(x) // ch
    return x[: n * ch].reshape(n, ch), sr


def to_mono(x: np.ndarray) -> np.ndarray:
    """Channel mean (librosa.to_mono semantics)."""
    return x if x.ndim == 1 else x.mean(axis=1, dtype=np.float32) if x.shape[1] > 1 else x[:, 0]


_SOXR_HQ_BITS = 20  # soxr_quality_spec(SOXR_HQ): 20-bit precision
_BETA_FIT = (  # libsoxr lsx_kaiser_beta: cubic fits of the Kaiser beta over the attenuation, per octave of transition width
    (-6.784957e-10, 1.02856e-05, 0.1087556, -0.8978365),
    (-6.897885e-10, 1.027433e-05, 0.10876, -0.8974658),
    (-1.000683e-09, 1.030092e-05, 0.1087677, -0.8977898),
    (-3.654474e-10, 1.040631e-05, 0.1087085, -0.8917766),
    (8.106988e-09, 6.983091e-06, 0.1091387, -0.9022048),
    (9.519571e-09, 7.272678e-06, 0.1090068, -0.8890768),
    (-5.626821e-09, 1.342186e-05, 0.1083999, -0.8565452),
    (-9.965946e-08, 5.073548e-05, 0.1040967, -0.6822778),
    (1.604808e-07, -5.856462e-05, 0.1185998, -1.24824),
    (-1.511964e-07, 6.363034e-05, 0.1064627, -0.8076665),
)


def soxr_hq_taps(up: int, down: int) -> np.ndarray:
    """The anti-alias / anti-image filter of an up : down rate change at the rate `orig_sr * up`, DC gain `up`.

    libsoxr's SOXR_HQ recipe (soxr.c soxr_quality_spec, filter.c lsx_design_lpf / lsx_kaiser_beta / lsx_make_lpf):
    pass-band end 1 - .05 / TO_3dB(20 bit) = 0.9136 and stop-band begin 1.0 of the lower Nyquist, 21 bit = 126.4 dB,
    6 dB point half way, beta from libsoxr's fit, taps from its length formula rounded up to 1 (mod 4)."""
    from scipy.special import i0

    db = 20.0 * np.log10(2.0)
    rej = _SOXR_HQ_BITS * db
    fp = 1.0 - 0.05 / ((1.6e-6 * rej - 7.5e-4) * rej + 0.646)
    att = (_SOXR_HQ_BITS + 1) * db
    fn = float(max(up, down))  # Nyquist of the filter's rate in units of the lower Nyquist
    tr_bw = 0.5 * (1.0 - fp) / fn
    fc = 1.0 / fn - tr_bw
    realm = np.log2(tr_bw * 0.5 / fc / 0.0005)
    r0 = int(np.clip(int(realm), 0, 9))
    r1 = int(np.clip(int(realm) + 1, 0, 9))
    b0, b1 = (((c[0] * att + c[1]) * att + c[2]) * att + c[3] for c in (_BETA_FIT[r0], _BETA_FIT[r1]))
    beta = b0 + (b1 - b0) * (realm - int(realm))
    n = int(np.ceil((((0.0007528358 - 1.577737e-05 * beta) * beta + 0.6248022) * beta + 0.06186902) / tr_bw + 1))
    n = (n + 2) // 4 * 4 + 1
    z = np.arange(n, dtype=np.float64) - 0.5 * (n - 1)
    sinc = np.where(z != 0, np.sin(fc * np.pi * z) / np.where(z != 0, np.pi * z, 1.0), fc)
    y = z / (0.5 * (n - 1) + 0.5)
    return sinc * i0(beta * np.sqrt(1.0 - y * y)) / i0(beta) * up


def resample(x: np.ndarray, orig_sr: int, target_sr: int = AUDIO_SAMPLE_RATE) -> np.ndarray:
    """`librosa.resample(x, orig_sr, target_sr)` with its default soxr_hq response: one zero-phase polyphase stage,
    the signal zero outside the file, output length ceil(n * target / orig)."""
    if orig_sr == target_sr:
        return np.ascontiguousarray(x, dtype=np.float32)
    import scipy.signal

    fr = Fraction(int(target_sr), int(orig_sr))
    up, down = fr.numerator, fr.denominator
    h = soxr_hq_taps(up, down)
    c = (len(h) - 1) // 2
    n_out = int(np.ceil(len(x) * target_sr / orig_sr))
    # upfirdn output i = sum_j x[j] h[i * down - j * up]; output sample k sits at filter index k * down + c
    lead = -c % down
    y = scipy.signal.upfirdn(np.concatenate([np.zeros(lead), h]), x.astype(np.float64), up, down)
    k0 = (c + lead) // down
    y = y[k0 : k0 + n_out]
    if len(y) < n_out:
        y = np.pad(y, (0, n_out - len(y)))
    return np.ascontiguousarray(y, dtype=np.float32)


def read_flac(path: Union[str, pathlib.Path]) -> Tuple[np.ndarray, int]:
    """FLAC -> (float32 samples [n, channels] in [-1, 1), sample_rate): the native decoder (csrc/flac_decode.cpp),
    CRC- and MD5-checked; value / 2^(bits - 1) like libsndfile's float read that librosa.load uses."""
    import ctypes as C

    from . import _native

    lib = _native.load_library()
    with open(path, "rb") as f:
        data = f.read()
    ch, sr, bits, n = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    rc = lib.bp_flac_info(data, len(data), C.byref(ch), C.byref(sr), C.byref(bits), C.byref(n))
    if rc != _native.BP_OK:
        raise ValueError(f"{path}: {lib.bp_audio_last_error().decode(errors='replace')}")
    pcm = np.empty((n.value, ch.value), dtype=np.float32)
    got = C.c_int64()
    rc = lib.bp_flac_decode(data, len(data), pcm.ctypes.data, n.value, C.byref(got))
    if rc != _native.BP_OK or got.value != n.value:
        raise ValueError(f"{path}: {lib.bp_audio_last_error().decode(errors='replace')}")
    return pcm, sr.value


def _read_with_optional_backend(path: str) -> Tuple[np.ndarray, int]:
    """mp3 / ogg / m4a and anything else librosa.load would hand to soundfile or audioread: use whichever of those
    decoders this installation has (soundfile, audioread, an `ffmpeg` executable); none ships with this package."""
    try:
        import soundfile  # type: ignore

        x, sr = soundfile.read(path, dtype="float32", always_2d=True)
        return np.ascontiguousarray(x), int(sr)
    except ImportError:
        pass
    try:
        import audioread  # type: ignore

        with audioread.audio_open(path) as f:
            sr, ch = f.samplerate, f.channels
            buf = b"".join(f)
        x = np.frombuffer(buf, dtype="<i2").astype(np.float32) / 32768.0
        return x[: len(x) // ch * ch].reshape(-1, ch), int(sr)
    except ImportError:
        pass
    import shutil
    import subprocess

    ffmpeg = shutil.which("ffmpeg")
    if ffmpeg:
        probe = subprocess.run([ffmpeg, "-i", path], capture_output=True, text=True).stderr
        import re

        m = re.search(r"Audio:.*?(\d+) Hz, (mono|stereo|(\d+) channels|[\d.]+)", probe)
        if m:
            sr = int(m.group(1))
            ch = {"mono": 1, "stereo": 2}.get(m.group(2)) or (int(m.group(3)) if m.group(3) else 2)
            raw = subprocess.run([ffmpeg, "-v", "error", "-i", path, "-f", "f32le", "-acodec", "pcm_f32le", "-ac", str(ch),
                                  "-ar", str(sr), "-"], capture_output=True, check=True).stdout
            x = np.frombuffer(raw, dtype="<f4")
            return np.ascontiguousarray(x[: len(x) // ch * ch].reshape(-1, ch)), sr
    raise ValueError(
        f"{path}: not a WAV or FLAC file, and no decoder for other formats (mp3 / ogg / m4a) is installed: "
        "basic_pitch_amd reads RIFF/WAVE and FLAC natively and uses soundfile, audioread or an ffmpeg executable when present"
    )


def read_audio(path: Union[str, pathlib.Path]) -> Tuple[np.ndarray, int]:
    """Decode any supported file to (float32 [n, channels], sample_rate): the decode half of librosa.load
    (inference.py:239).  Dispatch is by content, not extension: RIFF/WAVE and FLAC are read natively."""
    with open(path, "rb") as f:
        head = f.read(12)
    if head[:4] == b"RIFF" and head[8:12] == b"WAVE":
        return read_wav(path)
    if head[:4] == b"fLaC" or (head[:3] == b"ID3" and str(path).lower().endswith(".flac")):
        return read_flac(path)
    return _read_with_optional_backend(str(path))


def load(path: Union[str, pathlib.Path], sr: int = AUDIO_SAMPLE_RATE, mono: bool = True) -> Tuple[np.ndarray, int]:
    """`librosa.load(path, sr=sr, mono=True)`: decode, channel mean, soxr_hq-design resampling."""
    x, file_sr = read_audio(path)
    y = to_mono(x) if mono else x
    return resample(np.ascontiguousarray(y), file_sr, sr), sr


def get_duration(filename: Union[str, pathlib.Path]) -> float:
    x, sr = read_audio(filename)
    return x.shape[0] / float(sr)
