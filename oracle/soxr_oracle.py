"""ORACLE — restatement of the resampler behind `librosa.load(path, sr=22050, mono=True)`.  TEST INFRASTRUCTURE ONLY.

Reference call site: `basic_pitch/inference.py:239`.  librosa (>= 0.10, `pyproject.toml:22` pins `librosa>=0.8.0`) resamples
with `res_type="soxr_hq"`, i.e. libsoxr (0.1.3, through python-soxr) at quality `SOXR_HQ`.  Neither librosa nor libsoxr is
in `/root/reference` or installable here, so this file restates libsoxr's *published* design:

  * `soxr_quality_spec(SOXR_HQ, 0)` (soxr.c): precision = 20 bit, linear phase, `stopband_begin = 1` (of the lower
    Nyquist), `passband_end = 1 - .05 / TO_3dB(rej)` with `rej = 20 bit * 6.0206 dB`, `TO_3dB(a) = (1.6e-6 a - 7.5e-4) a + .646`
    -> 0.913628;
  * stop-band attenuation `(precision + 1) * 6.0206 dB` = 126.43 dB (cr.c, "+1: pass+stop");
  * `lsx_design_lpf` / `lsx_kaiser_params` / `lsx_kaiser_beta` / `lsx_make_lpf` (filter.c): Kaiser-windowed sinc, the 6 dB point
    half way between pass-band end and stop-band begin, beta from libsoxr's polynomial fit (13.04 here), the tap count
    from its empirical formula rounded up to 1 (mod 4) -> 389 taps at the input rate for 2 : 1;
  * an exact 2 : 1 ratio is ONE stage in libsoxr (a `dft_stage` with L = 1, M = 2: overlap-save FFT convolution with
    that filter, decimation folded into the frequency domain), the delay compensated so that output sample k sits at
    input sample 2 k, the signal taken as zero outside the file; `librosa.resample` then fixes the length to
    ceil(n * 22050 / sr).
  FFT convolution == direct convolution up to rounding, so the direct form below IS that stage in float64.

Pin: with this resampler in front, the graph oracle (`bp_oracle.py`) reproduces the reference's golden posteriorgrams
for its 44.1 kHz test clip to <= 5e-5 max-abs, inside the reference's own `atol=1e-4`
(`tests/test_inference.py:66-70`) — `tests/test_oracle_golden.py::test_oracle_reproduces_golden_posteriorgrams`.
With scipy's default polyphase design in front instead the same oracle is 4e-3 away: the test keeps that
comparison, so "the residual was the resampler" is a tested statement.  Other ratios than 2 : 1: libsoxr splits the work
into several stages (half-band stages, a polyphase stage with 147 phases for 48 kHz, ...); here they are ONE polyphase
stage with the same pass-band / stop-band / attenuation — the same response to the filter's 1e-6 ripple, but no golden
vector exists for them: unpinned beyond the design.
"""
from __future__ import annotations

from fractions import Fraction

import numpy as np
from scipy.special import i0

SOXR_HQ_BITS = 20
_DB_PER_BIT = 20.0 * np.log10(2.0)

# lsx_kaiser_beta (libsoxr filter.c): cubic fits of beta over the attenuation, one row per octave of tr_bw / .0005
_BETA_FIT = (
    (-6.784957e-10, 1.02856e-05, 0.1087556, -0.8988365 + 0.001),
    (-6.897885e-10, 1.027433e-05, 0.10876, -0.8994658 + 0.002),
    (-1.000683e-09, 1.030092e-05, 0.1087677, -0.9007898 + 0.003),
    (-3.654474e-10, 1.040631e-05, 0.1087085, -0.8977766 + 0.006),
    (8.106988e-09, 6.983091e-06, 0.1091387, -0.9172048 + 0.015),
    (9.519571e-09, 7.272678e-06, 0.1090068, -0.9140768 + 0.025),
    (-5.626821e-09, 1.342186e-05, 0.1083999, -0.9065452 + 0.05),
    (-9.965946e-08, 5.073548e-05, 0.1040967, -0.7672778 + 0.085),
    (1.604808e-07, -5.856462e-05, 0.1185998, -1.34824 + 0.1),
    (-1.511964e-07, 6.363034e-05, 0.1064627, -0.9876665 + 0.18),
)


def hq_spec():
    """(passband_end, stopband_begin, attenuation_dB) of SOXR_HQ, frequencies as fractions of the lower Nyquist."""
    rej = SOXR_HQ_BITS * _DB_PER_BIT
    to_3db = (1.6e-6 * rej - 7.5e-4) * rej + 0.646
    return 1.0 - 0.05 / to_3db, 1.0, (SOXR_HQ_BITS + 1) * _DB_PER_BIT


def kaiser_beta(att: float, tr_bw: float) -> float:
    """lsx_kaiser_beta for att >= 60 dB."""
    realm = np.log(tr_bw / 0.0005) / np.log(2.0)
    lo = int(np.clip(int(realm), 0, len(_BETA_FIT) - 1))
    hi = int(np.clip(1 + int(realm), 0, len(_BETA_FIT) - 1))
    b0 = ((_BETA_FIT[lo][0] * att + _BETA_FIT[lo][1]) * att + _BETA_FIT[lo][2]) * att + _BETA_FIT[lo][3]
    b1 = ((_BETA_FIT[hi][0] * att + _BETA_FIT[hi][1]) * att + _BETA_FIT[hi][2]) * att + _BETA_FIT[hi][3]
    return b0 + (b1 - b0) * (realm - int(realm))


def design_lpf(Fp: float, Fs: float, Fn: float, att: float, modulo: int = 4) -> np.ndarray:
    """lsx_design_lpf(Fp, Fs, Fn, att, &num_taps = 0, k = -modulo, beta = -1) + lsx_make_lpf(rho = .5, scale = 1)."""
    Fp, Fs = Fp / Fn, Fs / Fn
    tr_bw = min(0.5 * (Fs - Fp), 0.5 * Fs)
    Fc = Fs - tr_bw
    beta = kaiser_beta(att, tr_bw * 0.5 / Fc)
    a = ((0.0007528358 - 1.577737e-05 * beta) * beta + 0.6248022) * beta + 0.06186902  # lsx_kaiser_params, att >= 60
    n = int(np.ceil(a / tr_bw + 1))
    n = (n + modulo - 2) // modulo * modulo + 1
    m = n - 1
    z = np.arange(n, dtype=np.float64) - 0.5 * m
    x = z * np.pi
    h = np.where(x != 0, np.sin(Fc * x) / np.where(x != 0, x, 1.0), Fc)
    y = z / (0.5 * m + 0.5)
    return h * i0(beta * np.sqrt(1.0 - y * y)) / i0(beta)


def resample(x: np.ndarray, orig_sr: int, target_sr: int = 22050) -> np.ndarray:
    """mono float signal at orig_sr -> float32 at target_sr, ceil(n * target / orig) samples, soxr_hq response.

    Direct form, float64: y[k] = up * sum_j x[j] h[k * down - j * up + c] (zero outside the file, c = filter centre)."""
    x = np.asarray(x, dtype=np.float64)
    n_out = int(np.ceil(len(x) * target_sr / orig_sr))
    if orig_sr == target_sr:
        return x.astype(np.float32)
    fr = Fraction(int(target_sr), int(orig_sr))
    up, down = fr.numerator, fr.denominator
    Fp, Fs, att = hq_spec()
    h = design_lpf(Fp, Fs, float(max(up, down)), att) * up
    c = (len(h) - 1) // 2
    if up == 1:
        xx = np.concatenate([np.zeros(c), x, np.zeros(c + down)])
        return np.convolve(xx, h, mode="valid")[::down][:n_out].astype(np.float32)
    y = np.zeros(n_out)
    for k in range(n_out):  # small cases only
        base = k * down + c
        j_lo = max(0, -(-(base - (len(h) - 1)) // up))
        j_hi = min(len(x) - 1, base // up)
        j = np.arange(j_lo, j_hi + 1)
        y[k] = np.dot(x[j], h[base - j * up])
    return y.astype(np.float32)
