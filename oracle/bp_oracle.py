"""ORACLE — CPU restatement of the reference's frozen Basic Pitch graph.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this file; the
product path (`basic_pitch_amd/`) never does and fails loudly without its HIP library.

What it restates (all `file:line` relative to the reference checkout, spotify/basic-pitch v0.4.0):
  * `basic_pitch/layers/nnaudio.py:623-661` CQT2010v2.call  (+ 216-256 get_cqt_complex, 259-284
    downsampling_by_n, 287-301 ReflectionPad1D)                                   -> `cqt()`
  * `basic_pitch/layers/signal.py:171-185` NormalizedLog.call, `layers/math.py:21-32`  -> `normalized_log()`
  * `basic_pitch/models.py:187-189` BatchNormalization (inference affine, folded constants read
    from the frozen artifact `saved_models/icassp_2022/nmp.onnx`)
  * `basic_pitch/nn.py:69-88` HarmonicStacking.call                              -> `harmonic_stack()`
  * `basic_pitch/models.py:241-318` the six Conv2D layers (BN folded, as frozen)  -> `cnn()`
  * `basic_pitch/inference.py:194-219,222-244,247-279` windowing / un-overlapping -> `window_track()`,
    `unwrap_output()`

The arithmetic of the reference lives in third-party runtimes (TensorFlow / onnxruntime / TFLite /
CoreML) that are not installed here, so this is a *restatement* of the graph those runtimes execute.
Pinning status (see DESIGN.md "Oracle"): the graph constants are bit-identical to the reference's
formulas and artifact (tools/extract_weights.py asserts it); behind `soxr_oracle.py` (the restatement of
librosa's soxr_hq resampler) the end-to-end output reproduces the reference's golden posteriorgrams for
`vocadito_10.wav` at the reference's OWN tolerance, atol = 1e-4 on every element (measured 2.2e-5 /
4.6e-5 / 3.4e-5 max-abs on note / onset / contour) — tests/test_oracle_golden.py; that clip is the only
numeric pin the reference has.  Graph-level agreement with the real third-party runtimes on OTHER inputs
cannot be pinned here (none is installable): there the float64 evaluation below is the yardstick.

Two precisions: `dtype=np.float64` is the "truth" oracle; `dtype=np.float32` mimics the
reference's fp32 execution (summation order is torch-CPU's, not TF's — see SURVEY.md §7 hard
part 1 for why the tolerance in the parity tests is noise-aware).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# ---- geometry (basic_pitch/constants.py:25-47) --------------------------------------------------
AUDIO_SAMPLE_RATE = 22050
FFT_HOP = 256
AUDIO_N_SAMPLES = 43844
ANNOT_N_FRAMES = 172
N_BINS_CQT = 309
N_FREQ_CONTOUR = 264
N_FREQ_NOTE = 88
N_OCTAVES = 9
BINS_PER_OCTAVE = 36
N_OVERLAPPING_FRAMES = 30  # inference.py:190
OVERLAP_LEN = N_OVERLAPPING_FRAMES * FFT_HOP  # 7680, inference.py:304
HOP_SIZE = AUDIO_N_SAMPLES - OVERLAP_LEN  # 36164, inference.py:305
HARMONIC_SHIFTS = (-36, 0, 36, 57, 72, 84, 93, 101)  # nn.py:51-54 with models.py:213-218

DEFAULT_WEIGHTS = os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "..", "basic_pitch_amd", "assets", "nmp_weights.bin"
)


def load_weights(path: str = DEFAULT_WEIGHTS) -> Dict[str, np.ndarray]:
    """Parse the weights blob (format documented in include/basic_pitch_amd.h)."""
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:8] != b"BPAMDW01":
        raise ValueError("bad weights blob magic")
    version, n = struct.unpack_from("<II", blob, 8)
    if version != 1:
        raise ValueError("unsupported weights blob version")
    ent = 16
    data0 = ent + 52 * n
    out: Dict[str, np.ndarray] = {}
    for i in range(n):
        name, ndim, d0, d1, d2, d3, off, cnt = struct.unpack_from("<24sI4III", blob, ent + 52 * i)
        arr = np.frombuffer(blob, dtype="<f4", count=cnt, offset=data0 + 4 * off)
        out[name.rstrip(b"\0").decode()] = arr.reshape([d0, d1, d2, d3][:ndim]).copy()
    return out


def _t(a: np.ndarray, dtype) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a)).to(torch.float64 if dtype == np.float64 else torch.float32)


# ---- CQT -----------------------------------------------------------------------------------------
# ---- extended range for 44.1 kHz input (BASELINE.json configs[4]) --------------------------
# NOT a behaviour of the reference (it always resamples to 22.05 kHz): nnaudio.CQT2010v2 re-parametrised with
# sr = 44100, hop = 512, n_bins = 345 -> n_octaves = 10, remainder 21 again; fmin_t / sr is unchanged, so the SAME 36
# kernels and the same low-pass apply; `lengths` (nnaudio.py:590-593) use sr = 44100; windows are 87,688 samples.
EXT_AUDIO_N_SAMPLES = 2 * AUDIO_N_SAMPLES
EXT_N_OCTAVES = N_OCTAVES + 1
EXT_N_BINS_CQT = N_BINS_CQT + BINS_PER_OCTAVE
EXT_FFT_HOP = 2 * FFT_HOP
EXT_SAMPLE_RATE = 2 * AUDIO_SAMPLE_RATE


def ext_sqrt_len() -> np.ndarray:
    """sqrt(ceil(Q * 44100 / f_b)), f_b = 27.5 * 2^(b/36), b < 345 (nnaudio.py:532, 590-593, 649-650)."""
    q = 1.0 / (2.0 ** (1.0 / BINS_PER_OCTAVE) - 1.0)
    f = 27.5 * 2.0 ** (np.arange(EXT_N_BINS_CQT) / float(BINS_PER_OCTAVE))
    return np.sqrt(np.ceil(q * EXT_SAMPLE_RATE / f)).astype(np.float32)


def pyramid(x: torch.Tensor, lowpass: torch.Tensor, n_octaves: int = N_OCTAVES):
    """Levels 0..n_octaves-1: level k+1 = 256-tap FIR, stride 2, zero-pad 127 each side (nnaudio.py:269-279)."""
    levels = [x]
    cur = x[:, None, :]
    for _ in range(n_octaves - 1):
        cur = F.conv1d(F.pad(cur, (127, 127)), lowpass[None, None, :], stride=2)
        levels.append(cur[:, 0, :])
    return levels


def cqt(x: torch.Tensor, W: Dict[str, np.ndarray], dtype=np.float64, return_levels: bool = False, ext: bool = False):
    """x (B, 43844) -> magnitude (B, 172, 309)  (nnaudio.py:623-661, 'Magnitude' output).
    ext: x (B, 87688) at 44.1 kHz -> (B, 172, 345), the re-parametrisation described above."""
    k_re = _t(W["cqt_kernel_re"], dtype)[:, None, :]
    k_im = _t(W["cqt_kernel_im"], dtype)[:, None, :]
    lowpass = _t(W["cqt_lowpass"], dtype)
    sqrt_len = _t(ext_sqrt_len() if ext else W["cqt_sqrt_len"], dtype)
    n_bins = EXT_N_BINS_CQT if ext else N_BINS_CQT
    levels = pyramid(x, lowpass, EXT_N_OCTAVES if ext else N_OCTAVES)
    re_oct, im_oct = [], []
    hop = EXT_FFT_HOP if ext else FFT_HOP
    for lvl in levels:
        xp = F.pad(lvl[:, None, :], (128, 128), mode="reflect")  # nnaudio.py:229, 300-301
        re_oct.append(F.conv1d(xp, k_re, stride=hop))  # (B, 36, 172)
        im_oct.append(-F.conv1d(xp, k_im, stride=hop))  # nnaudio.py:246
        hop //= 2
    # lower octaves are prepended (nnaudio.py:640) and the bottom 15 bins dropped (642)
    re = torch.cat(re_oct[::-1], dim=1)[:, -n_bins:, :]
    im = torch.cat(im_oct[::-1], dim=1)[:, -n_bins:, :]
    re = re * sqrt_len[None, :, None]  # nnaudio.py:650 (scale BEFORE squaring)
    im = im * sqrt_len[None, :, None]
    mag = torch.sqrt(re * re + im * im).permute(0, 2, 1).contiguous()  # nnaudio.py:661
    if return_levels:
        return mag, levels
    return mag


def normalized_log(mag: torch.Tensor, W: Dict[str, np.ndarray], dtype=np.float64):
    """signal.py:171-185 as frozen (ONNX nodes 190-209: ln * (1/ln10) * 10).

    Returns (normalised, log_power, min, max) with min/max the per-window extrema of log_power.
    """
    eps = float(W["log_eps"][0]) if dtype == np.float32 else 1e-10
    s0, s1 = (float(W["log_scale"][0]), float(W["log_scale"][1]))
    if dtype == np.float64:
        s0 = 1.0 / np.log(10.0)
    power = mag * mag
    lp = torch.log(power + eps) * s0 * s1
    mn = lp.amin(dim=(1, 2), keepdim=True)
    off = lp - mn
    mx = off.amax(dim=(1, 2), keepdim=True)
    norm = torch.where(mx == 0, torch.zeros_like(off), off / torch.where(mx == 0, torch.ones_like(mx), mx))
    return norm, lp, mn.reshape(-1), lp.amax(dim=(1, 2))


def harmonic_stack(z: torch.Tensor) -> torch.Tensor:
    """z (B,172,309) -> (B,8,172,264) NCHW; zero outside [0,309) (nn.py:69-88)."""
    B, T, Fq = z.shape
    out = z.new_zeros((B, len(HARMONIC_SHIFTS), T, N_FREQ_CONTOUR))
    for c, s in enumerate(HARMONIC_SHIFTS):
        lo = max(0, -s)
        hi = min(N_FREQ_CONTOUR, Fq - s)
        out[:, c, :, lo:hi] = z[:, :, lo + s : hi + s]
    return out


def cnn(stack: torch.Tensor, W: Dict[str, np.ndarray], dtype=np.float64) -> Dict[str, torch.Tensor]:
    """models.py:241-318 with the BN-folded weights of the frozen graph."""
    g = lambda k: _t(W[k], dtype)  # noqa: E731
    c1 = F.relu(F.conv2d(stack, g("contour1_w"), g("contour1_b"), padding=(1, 19)))
    contour = torch.sigmoid(F.conv2d(c1, g("contour2_w"), g("contour2_b"), padding=(2, 2)))  # (B,1,172,264)
    n1 = F.relu(F.conv2d(F.pad(contour, (2, 2, 3, 3)), g("note1_w"), g("note1_b"), stride=(1, 3)))
    note = torch.sigmoid(F.conv2d(n1, g("note2_w"), g("note2_b"), padding=(3, 1)))  # (B,1,172,88)
    o1 = F.relu(F.conv2d(F.pad(stack, (1, 1, 2, 2)), g("onset1_w"), g("onset1_b"), stride=(1, 3)))
    cat = torch.cat([note, o1], dim=1)  # models.py:305: channel 0 = post-sigmoid note map
    onset = torch.sigmoid(F.conv2d(cat, g("onset2_w"), g("onset2_b"), padding=(1, 1)))
    return {"c1": c1, "contour": contour[:, 0], "n1": n1, "note": note[:, 0], "o1": o1, "onset": onset[:, 0]}


def forward(
    audio: np.ndarray, W: Dict[str, np.ndarray] | None = None, dtype=np.float64, intermediates: bool = False,
    ext: bool = False,
) -> Dict[str, np.ndarray]:
    """audio (B, 43844) or (B, 43844, 1) -> {"note","onset","contour"} (+ intermediates).
    ext: audio (B, 87688) at 44.1 kHz through the extended 345-bin CQT; CNN and output shapes unchanged."""
    if W is None:
        W = load_weights()
    x = np.asarray(audio)
    if x.ndim == 3:
        x = x[:, :, 0]  # nn.py:91-102 FlattenAudioCh
    assert x.ndim == 2 and x.shape[1] == (EXT_AUDIO_N_SAMPLES if ext else AUDIO_N_SAMPLES), x.shape
    xt = _t(x, dtype)
    with torch.no_grad():
        mag, levels = cqt(xt, W, dtype, return_levels=True, ext=ext)
        norm, lp, mn, mx = normalized_log(mag, W, dtype)
        z = norm * float(W["bn_affine"][0]) + float(W["bn_affine"][1])
        stack = harmonic_stack(z)
        r = cnn(stack, W, dtype)
    out = {k: r[k].numpy() for k in ("note", "onset", "contour")}
    if intermediates:
        out.update(
            {
                "levels": [l.numpy() for l in levels],
                "mag": mag.numpy(),
                "lp": lp.numpy(),
                "minmax": np.stack([mn.numpy(), mx.numpy()], axis=1),
                "z": z.numpy(),
                "c1": r["c1"].numpy(),
                "n1": r["n1"].numpy(),
                "o1": r["o1"].numpy(),
            }
        )
    return out


# ---- windowing / stitching (inference.py) ---------------------------------------------------------
def window_track(samples: np.ndarray) -> Tuple[np.ndarray, int]:
    """inference.py:222-244 + 194-219: prepend 3840 zeros, hop 36164, zero-pad the tail."""
    samples = np.asarray(samples, dtype=np.float32)
    original_length = samples.shape[0]
    padded = np.concatenate([np.zeros(OVERLAP_LEN // 2, dtype=np.float32), samples])
    wins = []
    for i in range(0, padded.shape[0], HOP_SIZE):
        w = padded[i : i + AUDIO_N_SAMPLES]
        if len(w) < AUDIO_N_SAMPLES:
            w = np.pad(w, [[0, AUDIO_N_SAMPLES - len(w)]])
        wins.append(w)
    return np.stack(wins), original_length


def unwrap_output(output: np.ndarray, audio_original_length: int) -> np.ndarray:
    """inference.py:247-279."""
    n_olap = N_OVERLAPPING_FRAMES // 2
    output = output[:, n_olap:-n_olap, :]
    flat = output.reshape(output.shape[0] * output.shape[1], output.shape[2])
    n_expected_windows = audio_original_length / HOP_SIZE
    n_frames_per_window = (2 * 86) - N_OVERLAPPING_FRAMES
    return flat[: int(n_expected_windows * n_frames_per_window), :]


def run_track(samples: np.ndarray, W=None, dtype=np.float64, batch: int = 8) -> Dict[str, np.ndarray]:
    """inference.py:282-330 run_inference() on already-decoded 22.05 kHz mono samples."""
    wins, n = window_track(samples)
    outs = {"note": [], "onset": [], "contour": []}
    for i in range(0, len(wins), batch):
        r = forward(wins[i : i + batch], W, dtype)
        for k in outs:
            outs[k].append(r[k])
    return {k: unwrap_output(np.concatenate(v), n) for k, v in outs.items()}


# ---- the C restatement (oracle/bp_oracle.c), used as a cross-check and as bench.py's CPU baseline -----------
def c_library():
    """ctypes handle of oracle/_build/libbp_oracle.so (built by `make -C oracle`, i.e. __graft_entry__.build())."""
    import ctypes as C

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libbp_oracle.so")
    lib = C.CDLL(path)
    lib.bpo_forward.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.bpo_forward.restype = C.c_int
    return lib


def forward_c(audio: np.ndarray, n_threads: int = 0, weights_path: str = DEFAULT_WEIGHTS) -> Dict[str, np.ndarray]:
    """audio (B, 43844) -> {"note","onset","contour"} through the C restatement (fp32, OpenMP over windows)."""
    x = np.ascontiguousarray(audio, dtype=np.float32)
    if x.ndim == 3:
        x = np.ascontiguousarray(x[:, :, 0])
    n = x.shape[0]
    with open(weights_path, "rb") as f:
        blob = f.read()
    out = {
        "note": np.empty((n, ANNOT_N_FRAMES, N_FREQ_NOTE), np.float32),
        "onset": np.empty((n, ANNOT_N_FRAMES, N_FREQ_NOTE), np.float32),
        "contour": np.empty((n, ANNOT_N_FRAMES, N_FREQ_CONTOUR), np.float32),
    }
    rc = c_library().bpo_forward(
        blob, len(blob), x.ctypes.data, n, out["note"].ctypes.data, out["onset"].ctypes.data,
        out["contour"].ctypes.data, n_threads or (os.cpu_count() or 1),
    )
    if rc != 0:
        raise RuntimeError(f"bpo_forward failed: {rc}")
    return out
