"""Pin the oracle: reference golden vectors (tests/golden, from the reference's own known-answer test
`tests/test_inference.py:43-76,164-194`) and the frozen-graph constants.  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, make_windows
from oracle import bp_oracle as O


def test_weights_blob_tensors(weights):
    shapes = {k: v.shape for k, v in weights.items()}
    assert shapes["cqt_kernel_re"] == (36, 256) and shapes["cqt_kernel_im"] == (36, 256)
    assert shapes["cqt_lowpass"] == (256,) and shapes["cqt_sqrt_len"] == (309,)
    assert shapes["contour1_w"] == (8, 8, 3, 39) and shapes["onset2_w"] == (1, 33, 3, 3)
    assert shapes["note1_w"] == (32, 1, 7, 7) and shapes["note2_w"] == (1, 32, 7, 3)
    # folded BatchNorm affine and log constants of the frozen graph
    np.testing.assert_allclose(weights["bn_affine"], [2.480741024017334, -0.8769183158874512], rtol=0, atol=0)
    np.testing.assert_allclose(weights["log_scale"], [0.4342944622039795, 10.0], rtol=0, atol=0)
    assert abs(float(weights["log_eps"][0]) - 1e-10) < 1e-17


def test_cqt_constants_match_reference_formulas(weights):
    """nnaudio.py:45-76 (firwin2 low-pass), 158-213 (kernels), 590-593 (lengths) regenerated with scipy."""
    import scipy.signal

    sr, fmin, bpo, n_bins = 22050.0, 27.5, 36, 309
    Q = 1.0 / (2 ** (1 / bpo) - 1)
    lowpass = scipy.signal.firwin2(256, [0.0, 0.5 / 1.001, 0.5 * 1.001, 1.0], [1.0, 1.0, 0.0, 0.0]).astype(np.float32)
    assert np.array_equal(lowpass, weights["cqt_lowpass"])
    lengths = np.ceil(Q * sr / (fmin * 2.0 ** (np.arange(n_bins) / bpo)))
    assert np.array_equal(np.sqrt(lengths.astype(np.float32)), weights["cqt_sqrt_len"])
    fmax_t = fmin * 2 ** 8 * 2 ** (20 / bpo)
    fmin_t = fmax_t / 2 ** (1 - 1 / bpo)
    k = 7
    f = fmin_t * 2.0 ** (k / bpo)
    ln = np.ceil(Q * sr / f)
    start = int(np.ceil(128 - ln / 2.0)) - int(ln % 2)
    sig = scipy.signal.get_window("hann", int(ln), fftbins=True) * np.exp(np.r_[-ln // 2 : ln // 2] * 1j * 2 * np.pi * f / sr) / ln
    sig = (sig / np.linalg.norm(sig, 1)).astype(np.complex64)
    assert np.array_equal(weights["cqt_kernel_re"][k, start : start + int(ln)], sig.real)
    assert np.array_equal(weights["cqt_kernel_im"][k, start : start + int(ln)], sig.imag)


def test_windowing_known_answers(clip_22k):
    """tests/test_inference.py:164-194: 6 windows, original_length == 200607, first window = first samples."""
    assert clip_22k.shape[0] == 200607
    wins, n = O.window_track(clip_22k)
    assert n == 200607 and wins.shape == (6, O.AUDIO_N_SAMPLES)
    assert np.array_equal(wins[0][3840:], clip_22k[: O.AUDIO_N_SAMPLES - 3840])
    assert not wins[0][:3840].any()


def test_oracle_reproduces_golden_posteriorgrams(weights, clip_22k):
    """Reference: atol=1e-4 with its own (librosa/soxr) resampler.  Here the resampler is scipy's
    polyphase FIR, which alone moves the posteriorgrams by up to ~4e-3 — so the pin is
    5e-3 max-abs and 1e-4 mean-abs."""
    g = np.load(os.path.join(GOLDEN, "vocadito_10_model_output.npz"))
    r = O.run_track(clip_22k, weights, np.float32, batch=6)
    for k in ("note", "onset", "contour"):
        assert r[k].shape == g[k].shape == ((787, 88) if k != "contour" else (787, 264))
        d = np.abs(r[k] - g[k])
        assert d.max() <= 5e-3, (k, d.max())
        assert d.mean() <= 1e-4, (k, d.mean())


@pytest.mark.parametrize("kind,bound", [("uniform", 2e-4), ("normal", 2e-5), ("tones", 2e-3)])
def test_fp32_vs_fp64_noise_floor(weights, kind, bound):
    """SURVEY.md §7 hard part 1: the same graph in fp32 vs fp64 differs by up to ~1e-4 on noise-like
    input and ~1e-3 on tonal input; the parity tolerance in the GPU tests is derived from this."""
    x = make_windows(kind, 2, seed=3)
    a = O.forward(x, weights, np.float32)
    b = O.forward(x, weights, np.float64)
    for k in a:
        assert np.abs(a[k] - b[k]).max() <= bound, (kind, k, np.abs(a[k] - b[k]).max())


def test_all_zero_window_is_finite(weights):
    """divide_no_nan (signal.py:183): a silent window gives finite, constant-ish outputs."""
    r = O.forward(np.zeros((1, O.AUDIO_N_SAMPLES), np.float32), weights, np.float32, intermediates=True)
    for k in ("note", "onset", "contour"):
        assert np.isfinite(r[k]).all()
    assert np.allclose(r["z"], weights["bn_affine"][1])


def test_harmonic_stack_shifts():
    import torch

    z = torch.arange(309, dtype=torch.float64).reshape(1, 1, 309) + 1.0
    s = O.harmonic_stack(z)[0, :, 0, :]
    for c, sh in enumerate(O.HARMONIC_SHIFTS):
        for f in (0, 1, 40, 200, 263):
            expect = f + sh + 1.0 if 0 <= f + sh < 309 else 0.0
            assert s[c, f].item() == expect


def test_c_restatement_agrees_with_torch_oracle(weights):
    """oracle/bp_oracle.c (the CPU baseline bench.py times) is an independent restatement of the same graph:
    it must agree with the torch fp32 oracle to summation-order noise, and be as close to fp64 as it is."""
    import subprocess

    from conftest import make_windows

    subprocess.run(["make", "-s", "-C", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")], check=True)
    x = np.concatenate([make_windows("uniform", 2, 0), make_windows("normal", 1, 1)])
    r32 = O.forward(x, weights, np.float32)
    r64 = O.forward(x, weights, np.float64)
    rc = O.forward_c(x, 4)
    for k in ("note", "onset", "contour"):
        assert rc[k].shape == r32[k].shape
        assert np.abs(rc[k] - r32[k]).max() <= 5e-5, k
        assert np.abs(rc[k] - r64[k]).max() <= 2 * np.abs(r32[k] - r64[k]).max() + 2e-5, k


def test_extended_cqt_restatement_is_consistent(weights):
    """The 44.1 kHz / 345-bin re-parametrisation (BASELINE.json configs[4]) has no reference
    fixture; it is pinned to the pinned 22.05 kHz restatement instead: its sqrt(lengths) table continues the model's
    (bin b + 36 == bin b), and its bins 0..308 are the 22.05 kHz CQT of its own first pyramid level up to that ratio."""
    import torch

    ext = O.ext_sqrt_len()
    assert ext.shape == (345,) and np.array_equal(ext[36:], weights["cqt_sqrt_len"])
    x = np.random.default_rng(4).uniform(-1, 1, (1, O.EXT_AUDIO_N_SAMPLES)).astype(np.float32)
    r = O.forward(x, weights, np.float64, intermediates=True, ext=True)
    assert r["mag"].shape == (1, 172, 345) and r["contour"].shape == (1, 172, 264) and len(r["levels"]) == 10
    assert r["levels"][1].shape[1] == O.AUDIO_N_SAMPLES
    std = O.cqt(torch.from_numpy(r["levels"][1]), weights, np.float64).numpy()
    ratio = ext[:309].astype(np.float64) / weights["cqt_sqrt_len"].astype(np.float64)
    assert np.abs(r["mag"][:, :, :309] - std * ratio).max() <= 1e-12
    assert r["mag"][:, :, 309:].max() > 0  # the new top octave carries data
