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
comparison, so "the residual was the resampler" is a tested statement.  Other ratios than 2 : 1 (round 3): `stage_plan`
restates libsoxr's stage determination (cr.c) — 48 kHz, 16 kHz and 8 kHz come out as ONE rational poly-phase stage
(147 : 320, 441 : 320, 441 : 160: no half-band pre-stage), 11.025 kHz as one 2 x interpolating stage; 88.2 / 96 kHz put a
half-band decimation (a fixed coefficient table of libsoxr) and 32 / 24 kHz a 2 x interpolation in front of the poly-phase
stage.  `design_lpf(..., phases)` restates the poly-phase design rule (tap count k * phases - 1, Kaiser support rho = .75).
No golden vector exists for these ratios; what is open is measured instead of claimed: the designs that meet the SOXR_HQ
specification differ by the filter's alignment (half a filter-rate tick for the even-length poly-phase filter) — on the
reference's clip brought to 48 kHz the posteriorgrams of the candidates are within 9.4e-5 of each other, at 16 kHz
within 4.3e-4 (onset map; 1.1e-4 on note / contour): profiles/r03_soxr_plan_distance.md.
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


def design_lpf(Fp: float, Fs: float, Fn: float, att: float, modulo: int = 4, phases: int = 1) -> np.ndarray:
    """lsx_design_lpf(Fp, Fs, Fn, att, &num_taps = 0, k, beta = -1) + lsx_make_lpf(rho, scale = 1).

    k < 0 (`modulo`, phases = 1: the single-rate `dft_stage` filters): num_taps = 1 (mod modulo), rho = .5.
    k > 0 (`phases` > 1: the coefficient set of a poly-phase stage, designed at phases x the stage's input rate):
    transition band and stop-band scaled by 1 / phases, num_taps rounded to a multiple of phases minus 1, and the Kaiser
    window's support parameter rho = .63 below 120 dB, .75 from 120 dB (filter.c)."""
    Fp, Fs = Fp / Fn, Fs / Fn
    tr_bw = 0.5 * (Fs - Fp) / phases
    Fs = Fs / phases
    tr_bw = min(tr_bw, 0.5 * Fs)
    Fc = Fs - tr_bw
    beta = kaiser_beta(att, tr_bw * 0.5 / Fc)
    a = ((0.0007528358 - 1.577737e-05 * beta) * beta + 0.6248022) * beta + 0.06186902  # lsx_kaiser_params, att >= 60
    n = int(np.ceil(a / tr_bw + 1))
    if phases > 1:
        n = n // phases * phases + phases - 1
        rho = 0.63 if att < 120 else 0.75
    else:
        n = (n + modulo - 2) // modulo * modulo + 1
        rho = 0.5
    m = n - 1
    z = np.arange(n, dtype=np.float64) - 0.5 * m
    x = z * np.pi
    h = np.where(x != 0, np.sin(Fc * x) / np.where(x != 0, x, 1.0), Fc)
    y = z / (0.5 * m + rho)
    return h * i0(beta * np.sqrt(1.0 - y * y)) / i0(beta)


def stage_plan(io_ratio: float, coef_size_kbytes: float = 400.0, sizeof_real: int = 4) -> dict:
    """The stage determination of libsoxr's `_soxr_init` (cr.c, "Determine stages"), restated from the published source
    for a quality with `precision` > 16 bit, the default runtime spec (SOXR_COEF_INTERP_AUTO, small-integer
    optimisation on) and float32 samples.  Returns the factors of the pipeline
        [shr half-band decimations] -> [pre stage preL : preM] -> [arbitrary / poly-phase stage arbL : arbM] -> [post stage]
    `rational` says the middle stage is an exact poly-phase FIR with arbL phases (no coefficient interpolation).

    What this restatement is used for: deciding WHICH ratios libsoxr runs as one stage.  2 : 1 is one `dft_stage`
    (preL : preM = 1 : 2, pinned by the reference's golden vectors); 48 kHz -> 22.05 kHz and 16 kHz -> 22.05 kHz come out
    as ONE rational poly-phase stage (147 : 320 and 441 : 320) — no half-band pre-stage, contrary to what this file's
    header said in round 2; 88.2 / 96 kHz have one half-band decimation in front (a fixed coefficient table in libsoxr
    that is not reproducible from memory) and 32 kHz a 2 x interpolating pre-stage: for those the restatement stops at
    the plan."""
    U100_l = 42
    MULT32 = 65536.0 * 65536.0
    arbM, mode, n = float(io_ratio), 0, 0
    postL, iOpt = 1, True
    while True:
        n += 1
        if n != 1:
            break
        maxL = 2048 if mode else int(np.ceil(coef_size_kbytes * 1000.0 / (U100_l * sizeof_real)))
        upsample = arbM < 1
        shr, i = 0, int(0.5 * arbM)
        while True:
            i >>= 1
            if not i:
                break
            arbM *= 0.5
            shr += 1
        preM = 1 if (upsample or 1.5 < arbM < 2) else 0
        postM = 1 + (1 if (arbM > 1 and preM) else 0)
        arbM /= postM
        preL = 1 + (1 if (not preM and arbM < 2) else 0) + (1 if (upsample and mode) else 0)
        arbM *= preL
        frac, epsilon, arbL = arbM - int(arbM), 0.0, 1
        if frac != 0:
            epsilon = abs(np.floor(frac * MULT32 + 0.5) / (frac * MULT32) - 1)
        rational = frac == 0
        i = 1
        while i <= maxL and not rational:
            d = frac * i
            tr = int(d + 0.5)
            rational = tr > 0 and abs(tr / d - 1) <= epsilon
            if rational:
                if tr == i:
                    arbM = float(np.ceil(arbM))
                    x = 1 if arbM > 3 else 0
                    shr += x
                    arbM /= 1 + x
                else:
                    arbM, arbL = float(i * int(arbM) + tr), i
            i += 1
        L, M = preL * arbL, int(arbM * postM)
        x = (L | M) & 1
        if not x:
            L, M = L >> 1, M >> 1
        d = preL * arbL / arbM
        if iOpt and postL == 1 and d > 4 and d != 5:
            postL, i = 4, int(d / 16)
            while True:
                i >>= 1
                if not i or postL >= 256:
                    break
                postL <<= 1
            arbM, arbL, n = arbM * postL / arbL / preL, 1, 0
        elif rational and (max(L, M) < 3 + 2 * iOpt or L * M < 6 * iOpt):
            preL, preM, arbM, arbL, postM = L, M, 1.0, 1, 1
        if not mode and (not rational or not n):
            mode, n = mode + 1, 0
    return {"shr": shr, "preL": preL, "preM": max(preM, 1), "arbL": arbL, "arbM": arbM, "postL": postL, "postM": postM,
            "rational": bool(rational), "single_stage": shr == 0 and postL == 1 and postM == 1 and
            ((arbL == 1 and arbM == 1) or (preL == 1 and max(preM, 1) == 1))}


def taps(up: int, down: int, poly_rule: bool = True) -> np.ndarray:
    """The filter of an up : down rate change at the rate `orig_sr * up`, DC gain `up`.  up = 1: the single-rate design
    (k = -4); up > 1 and `poly_rule`: lsx_design_lpf's poly-phase rule (k = up phases) — the filter is then specified at
    the stage's input rate (Fn = down / up input Nyquists per output Nyquist when down-sampling) and designed at `up` times
    that rate, which is the same cut-off and transition band as the single-rate formula at the high rate, with the
    poly-phase tap count and window support."""
    Fp, Fs, att = hq_spec()
    if up > 1 and poly_rule:
        return design_lpf(Fp, Fs, max(float(down) / up, 1.0), att, phases=up) * up
    return design_lpf(Fp, Fs, float(max(up, down)), att) * up


def resample(x: np.ndarray, orig_sr: int, target_sr: int = 22050, poly_rule: bool = False) -> np.ndarray:
    """mono float signal at orig_sr -> float32 at target_sr, ceil(n * target / orig) samples, soxr_hq response.

    Direct form, float64: y[k] = up * sum_j x[j] h[k * down - j * up + c] (zero outside the file, c = filter centre).
    Ratios whose libsoxr pipeline is one stage (`stage_plan(...)["single_stage"]`: 2 : 1, 48 kHz, 16 kHz, 8 kHz, ...) are
    that stage; the others (a half-band decimation or a 2 x interpolation in front) run as one stage of the same
    specification.  `poly_rule` selects lsx_design_lpf's poly-phase tap-count / window rule for up > 1 (what libsoxr's
    stage runs); the default is the single-rate rule with an odd, exactly centred filter (zero phase: output k sits
    exactly at input time k * down / up) — the product's design.  The poly-phase rule's even tap count leaves half a
    filter-rate tick of alignment that cannot be settled without libsoxr itself; tools/soxr_plan_distance.py measures what
    the choice is worth on the posteriorgrams (profiles/r03_soxr_plan_distance.md)."""
    x = np.asarray(x, dtype=np.float64)
    n_out = int(np.ceil(len(x) * target_sr / orig_sr))
    if orig_sr == target_sr:
        return x.astype(np.float32)
    fr = Fraction(int(target_sr), int(orig_sr))
    up, down = fr.numerator, fr.denominator
    h = taps(up, down, poly_rule)
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
