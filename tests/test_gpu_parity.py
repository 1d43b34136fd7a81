"""GPU parity tests proper: every HIP stage and the whole path vs the oracle, through the C ABI.

Tolerances (floating point path; north-star bar: posteriorgrams within 1e-4 fp32):
  * stage tests feed the ORACLE's fp32 tensors into one HIP stage and compare with the oracle's fp32
    output of that stage: what remains is summation-order noise, bounds stated per stage below;
  * end-to-end: |hip - fp64 oracle| <= max(1e-4, 2 * |fp32 oracle - fp64 oracle|) per tensor (profiles/r02_parity.md) — the
    reference's own fp32 execution is only defined up to that noise (SURVEY.md §7 hard part 1), and
    plainly <= 1e-4 on the noise-like synthetic inputs BASELINE.json's configs use.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, make_windows
from oracle import bp_oracle as O

pytestmark = pytest.mark.gpu

F32 = torch.float32


@pytest.fixture(scope="module")
def runner():
    from stage_harness import StageRunner

    return StageRunner()


@pytest.fixture(scope="module")
def cases(weights):
    x = np.concatenate([make_windows("uniform", 2, 0), make_windows("normal", 1, 1), make_windows("tones", 1, 2)])
    r32 = O.forward(x, weights, np.float32, intermediates=True)
    r64 = O.forward(x, weights, np.float64, intermediates=True)
    return x, r32, r64


def test_native_library_is_loaded(runner):
    import ctypes

    info = runner.model.info()
    assert info["arch"].startswith("gfx950")
    maps = open("/proc/self/maps").read()
    assert "libbasicpitch_amd.so" in maps
    assert isinstance(runner.lib, ctypes.CDLL)


def test_stage_pyramid(runner, cases):
    from stage_harness import pyr_unpack

    x, r32, r64 = cases
    n = x.shape[0]
    out = runner.run("pyramid", n, {"audio": x}, {"pyr": ((n, 43712), F32)})
    lv = pyr_unpack(out["pyr"], runner.lib)
    for k in range(1, 9):
        # 256-tap fp32 dot products of O(1) data: 2e-6 absolute
        assert np.abs(lv[k] - r64["levels"][k]).max() <= 2e-6, k


def test_stage_filterbank(runner, cases):
    from stage_harness import ord_decode, pyr_pack

    x, r32, r64 = cases
    n = x.shape[0]
    out = runner.run(
        "filterbank", n, {"audio": x, "pyr": pyr_pack(r32["levels"], runner.lib)},
        {"lp": ((n, 172, 309), F32), "mm": ((n, 2), torch.int32)},
    )
    assert np.isfinite(out["lp"]).all()
    mag = np.sqrt(np.maximum(10.0 ** (out["lp"].astype(np.float64) / 10.0) - 1e-10, 0))
    # magnitudes are O(1); fp32 oracle itself is ~2.5e-6 from fp64
    assert np.abs(mag - r64["mag"]).max() <= 1e-5
    # log-power: compare where the bin is not cancellation noise (mag > 1e-3 of the window max)
    big = r64["mag"] > 1e-3 * r64["mag"].max(axis=(1, 2), keepdims=True)
    assert np.abs(out["lp"] - r64["lp"])[big].max() <= 2e-3
    mm = ord_decode(out["mm"])
    assert np.array_equal(mm[:, 0], out["lp"].min(axis=(1, 2)))
    assert np.array_equal(mm[:, 1], out["lp"].max(axis=(1, 2)))


@pytest.mark.parametrize(
    "stage,ins,outs,key,tol",
    [
        ("contour1", ("lp", "mm"), {"c1": (8, 172, 264)}, "c1", 5e-5),
        ("contour2", ("c1",), {"contour": (172, 264)}, "contour", 2e-6),
        ("note1", ("contour",), {"n1": (32, 172, 88)}, "n1", 5e-6),
        ("note2", ("n1",), {"note": (172, 88)}, "note", 2e-6),
        ("onset1", ("lp", "mm"), {"o1": (32, 172, 88)}, "o1", 2e-4),
        ("onset2", ("note", "o1"), {"onset": (172, 88)}, "onset", 2e-6),
    ],
)
def test_stage_cnn(runner, cases, stage, ins, outs, key, tol):
    from stage_harness import ord_encode

    x, r32, r64 = cases
    n = x.shape[0]
    feed = {}
    for k in ins:
        feed[k] = ord_encode(r32["minmax"]) if k == "mm" else r32[k]
    out = runner.run(stage, n, feed, {k: ((n,) + s, F32) for k, s in outs.items()})
    got = out[key]
    assert np.isfinite(got).all()
    d = np.abs(got - r32[key]).max()
    scale = max(1.0, float(np.abs(r32[key]).max()))
    assert d <= tol * scale, (stage, d)


def test_stage_zpack(runner, cases):
    """norm + BN written as pre-split f16 hi|lo words: hi + lo reproduces z to ~2^-22 relative."""
    from stage_harness import ord_encode, zp_unpack

    x, r32, r64 = cases
    n = x.shape[0]
    out = runner.run("zpack", n, {"lp": r32["lp"], "mm": ord_encode(r32["minmax"])},
                     {"zp": ((n, 174, 448), torch.int32)})
    zp = out["zp"].view(np.uint32)
    pad = zp.copy()
    pad[:, 1:173, 56 : 56 + 309] = 0
    assert (pad == 0).all()  # the zero padding of the stack / frame halo is part of the tensor
    z = zp_unpack(zp)
    assert np.abs(z - r32["z"]).max() <= 2e-6  # fp32 rounding of the normalisation + 2^-22 split residue


def test_zpack_folded_affine_is_within_two_ulp_of_the_graph_order(runner, weights):
    """the default path folds the frozen graph's sub, IEEE div, mul, add (signal.py:177-183, models.py:187-189)
    into z = (lp - min) * (bn_a / range) + bn_b with an FMA (bp_common.h norm_bn_k): no longer the reference's operation
    order, bounded here against it — <= 2 ulp of the product nrm x bn_a (4.8e-7 absolute) plus the operand split's 2^-22 — on the windows where the fold could
    bite: a silent window (range exactly 0: divide_no_nan's constant map, z = bn_b), ranges of 1e-6 and 3e-5 dB (a scale
    of ~1e6 on differences of a few ulp of lp), an ordinary window, and one whose extrema are extreme."""
    from stage_harness import ord_encode, zp_unpack

    rng = np.random.default_rng(21)
    n = 5
    lp = np.empty((n, 172, 309), np.float32)
    lp[0] = -100.0                                                      # silent: range 0
    lp[1] = np.float32(-37.5) + rng.integers(0, 2, (172, 309)).astype(np.float32) * np.float32(1e-6 * 4)   # a few ulp of range
    lp[2] = np.float32(-12.25) + rng.random((172, 309), dtype=np.float32) * np.float32(3e-5)
    lp[3] = rng.uniform(-100.0, 40.0, (172, 309)).astype(np.float32)
    lp[4] = rng.uniform(-0.001, 0.001, (172, 309)).astype(np.float32)
    lp[4, 0, 0], lp[4, 171, 308] = -100.0, 75.0
    mn = lp.reshape(n, -1).min(1)
    mx = lp.reshape(n, -1).max(1)
    mm = np.stack([mn, mx], 1).astype(np.float32)
    out = runner.run("zpack", n, {"lp": lp, "mm": ord_encode(mm)}, {"zp": ((n, 174, 448), torch.int32)})
    z = zp_unpack(out["zp"].view(np.uint32))
    bn_a, bn_b = np.float32(weights["bn_affine"][0]), np.float32(weights["bn_affine"][1])
    off = lp - mn[:, None, None]
    rngs = (mx - mn).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        nrm = np.where(rngs[:, None, None] == 0, np.float32(0), off / rngs[:, None, None]).astype(np.float32)
    ref = (nrm * bn_a).astype(np.float32) + bn_b                          # the graph's order: Mul, then Add
    assert np.array_equal(z[0], np.full((172, 309), bn_b, np.float32))    # the silent window: exactly bn_b
    # two roundings differ (the scale bn_a / range instead of the quotient, one FMA instead of Mul + Add): each is half an ulp
    # of the product nrm * bn_a <= 2.48, so the bound is absolute — 2 ulp of 2.48 — not relative to z, which passes through 0
    err = np.abs(z - ref)
    print("folded affine vs graph order, max |dz| per window:", err.reshape(n, -1).max(1))
    assert (err <= 2 * np.spacing(np.float32(2.48)) + np.abs(ref) * np.float32(2.0 ** -21)).all()


@pytest.mark.parametrize("branch", ["contour", "note", "onset"])
def test_stage_fused_branch(runner, cases, branch):
    """The fused split-precision branches (conv -> ReLU -> conv -> sigmoid in one kernel) against the
    oracle's fp32 posteriorgrams, fed with the oracle's own inputs of that branch: all products are split-f16
    (hi hi + lo hi + hi lo, fp32 accumulate): 5e-6.  (The fp8-corrections mode's kernels: A/B library,
    test_fp8_corrections_mode_lives_in_the_ab_library.)"""
    from stage_harness import zp_pack

    x, r32, r64 = cases
    n = x.shape[0]
    tol = 5e-6
    if branch == "note":
        feed = {"contour": r32["contour"]}
    elif branch == "contour":
        feed = {"zp": zp_pack(r32["z"]).view(np.int32)}
    else:
        feed = {"zp": zp_pack(r32["z"]).view(np.int32), "note": r32["note"]}
    width = 264 if branch == "contour" else 88
    got = runner.run(branch, n, feed, {branch: ((n, 172, width), F32)})[branch]
    assert np.isfinite(got).all()
    d32 = np.abs(got - r32[branch]).max()
    assert d32 <= tol, (branch, d32)


def test_exact_f32_reference_path(cases):
    """BP_FLAG_F32_MFMA runs the whole CNN on the exact-f32 kernels (f32 MFMA / VALU, intermediates in
    HBM).  It is the A/B reference of the default split-precision path: both must sit within the same
    distance of the fp64 oracle on the noise-like windows."""
    from basic_pitch_amd import Model

    x, r32, r64 = cases
    outs = {}
    for name, exact in (("split", False), ("f32", True)):
        m = Model(max_windows=8, exact_f32_mfma=exact)
        outs[name] = m.predict(x)
        m.close()
    for k in ("note", "onset", "contour"):
        e_split = np.abs(outs["split"][k][:3] - r64[k][:3]).max()
        e_f32 = np.abs(outs["f32"][k][:3] - r64[k][:3]).max()
        assert e_f32 <= 1e-4 and e_split <= 1e-4, (k, e_split, e_f32)
        assert e_split <= e_f32 + 3e-5, (k, e_split, e_f32)


def test_fp8_corrections_mode_lives_in_the_ab_library(cases, tmp_path):
    """BP_FLAG_FP8_CORRECTIONS (the correction products of the contour / onset conv1 on the block-scaled fp8 instruction: an
    opt-in, reduced-precision mode of rounds 2 - 5) left the product library in round 6 — it was no faster than the default
    any more and narrower than the config's fp32.  The product library refuses the flag loudly (ValueError naming the A/B
    library); BP_FLAG_F16_CORRECTIONS, the old name of today's default, is accepted, changes nothing, and wins over the fp8
    flag.  The mode itself still works in the A/B library (one subprocess): its branch kernels follow the fp32 oracle to
    2e-5 (contour) / 5e-5 (onset) on the oracle's own stage inputs, the whole path follows the fp64 graph to 1e-4 on these
    noise-like windows, and the default sits closer to it."""
    import subprocess
    import sys

    from basic_pitch_amd import Model

    x, r32, r64 = cases
    with pytest.raises(ValueError, match="A/B library"):
        Model(max_windows=8, fp8_corrections=True)
    outs = {}
    for name, kw in (("f16", {}), ("f16_flag", {"f16_corrections": True}), ("both", {"f16_corrections": True, "fp8_corrections": True})):
        m = Model(max_windows=8, **kw)
        outs[name] = m.predict(x)
        m.close()
    for k in ("note", "onset", "contour"):
        assert np.array_equal(outs["f16"][k], outs["f16_flag"][k]) and np.array_equal(outs["f16"][k], outs["both"][k]), k
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "experiments", "fp8_mode_ab.py")
    np.savez(str(tmp_path / "in.npz"), x=x, z=r32["z"], note=r32["note"])
    env = _ab_env(BP_NOTE="march16")  # any switch: selects the A/B library
    subprocess.run([sys.executable, tool, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")], check=True, env=env, timeout=900)
    ab = np.load(str(tmp_path / "out.npz"))
    assert np.abs(ab["stage_contour"] - r32["contour"]).max() <= 2e-5
    assert np.abs(ab["stage_onset"] - r32["onset"]).max() <= 5e-5
    for k, bound in (("note", 5e-5), ("onset", 6e-5), ("contour", 3e-5)):
        assert np.array_equal(ab[f"f16_{k}"], outs["f16"][k]), k  # the A/B library's default path is the product's
        assert np.abs(ab[f"fp8_{k}"] - outs["f16"][k]).max() <= bound, k
        for name in (ab[f"fp8_{k}"], outs["f16"][k]):
            assert np.abs(name[:3] - r64[k][:3]).max() <= 1e-4, k
        assert np.abs(outs["f16"][k][:3] - r64[k][:3]).max() <= np.abs(ab[f"fp8_{k}"][:3] - r64[k][:3]).max() + 1e-5, k
    assert np.abs(ab["fp8_contour"] - outs["f16"]["contour"]).max() > 0.0


def test_bench_batch_parity(weights):
    """The benchmark's own input, all of it: the 256 uniform[-1, 1) windows bench.py times (same torch generator, seed
    1234 + rank 0, drawn on the device) plus 256 normal(0, 0.01) windows, through the DEFAULT path (all-f16 split
    products) and the fp64 / fp32 oracles, window by window.

    What holds, and is asserted (profiles/r03_parity_bench_batch.md has the table this test condenses):
      * normal windows: all 256 within the plain 1e-4 of the north star, and within SURVEY.md 8c's bound;
      * full-scale white noise: the plain 1e-4 is NOT a property of fp32 evaluations of this graph — on ~2 % of such
        windows the per-window minimum of the log-power (signal.py:177) sits on a bin deep enough that fp32 rounding in
        the CQT moves every output of the window by 1e-4 .. 8e-4.  Measured on this batch: torch fp32 oracle 251/256
        within 1e-4 (max 5.4e-4), C fp32 oracle 250/256 (7.7e-4), the exact-f32 HIP kernels 251/256 (4.0e-4), the default
        path 252/256 (2.4e-4) — the same few windows for all of them, each evaluation an independent draw there
        (profiles/r04_parity_minbin.md: at the window's minimum bin the planes CQT's log-power error is SMALLER than the
        fp32 oracle's at every quantile — median 4.6e-4 dB vs 8.5e-4, max 9.5e-2 vs 1.6e-1).  So the
        path is held to the fp32 oracle's own distribution: at least as many windows inside 1e-4 (minus a binomial slack
        of 3), worst window and 99th percentile within 2x the oracle's, median not above the oracle's, SURVEY 8c's
        per-window bound max(1e-4, 2 |fp32 oracle - fp64|) on >= 98 % of the windows."""
    from basic_pitch_amd import Model

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    xb = (torch.rand((256, 43844), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()
    m = Model(max_windows=256)
    got_b = {k: v.cpu().numpy() for k, v in m.predict(xb).items()}
    xn = make_windows("normal", 256, seed=1)
    got_n = m.predict(xn)
    m.close()
    for name, x, got in (("bench uniform", xb.cpu().numpy(), got_b), ("normal", xn, got_n)):
        h64 = np.zeros(len(x))
        o64 = np.zeros(len(x))
        for i0 in range(0, len(x), 32):  # the oracle in slices: bounded memory, ~1.5 min for all 512 on the box's host cores
            sl = slice(i0, i0 + 32)
            r64 = O.forward(x[sl], weights, np.float64)
            r32 = O.forward(x[sl], weights, np.float32)
            for k in ("note", "onset", "contour"):
                h64[sl] = np.maximum(h64[sl], np.abs(got[k][sl] - r64[k]).max(axis=(1, 2)))
                o64[sl] = np.maximum(o64[sl], np.abs(r32[k] - r64[k]).max(axis=(1, 2)))
        bound = np.maximum(1e-4, 2.0 * o64)
        print(f"{name}: |hip-fp64| max {h64.max():.2e} p99 {np.quantile(h64, 0.99):.2e} median {np.median(h64):.2e}, within 1e-4 "
              f"{(h64 <= 1e-4).sum()}/{len(x)}, within max(1e-4, 2 x fp32 oracle) {(h64 <= bound).sum()}/{len(x)}; |fp32 oracle-fp64| "
              f"max {o64.max():.2e} p99 {np.quantile(o64, 0.99):.2e} median {np.median(o64):.2e}, within 1e-4 {(o64 <= 1e-4).sum()}/{len(x)}")
        assert np.isfinite(h64).all()
        assert (h64 <= bound).mean() >= 0.98, (name, int((h64 > bound).sum()))
        assert (h64 <= 1e-4).sum() >= (o64 <= 1e-4).sum() - 3, name
        assert h64.max() <= 2.0 * max(o64.max(), 1e-4), (name, float(h64.max()), float(o64.max()))
        assert np.quantile(h64, 0.99) <= 2.0 * max(np.quantile(o64, 0.99), 1e-4), name
        assert np.median(h64) <= np.median(o64), name
        if name == "normal":
            assert (h64 <= 1e-4).all() and (h64 <= bound).all(), (name, float(h64.max()))


def test_bf16_weights_mode(weights, cases):
    """BASELINE.json configs[3]: bf16 CNN weights + fp32 CQT (BP_FLAG_BF16_WEIGHTS, 2 matrix instructions per
    conv product instead of 3).  The parity claim is against the SAME graph with bf16-rounded conv weights (fp64
    oracle), to the usual 1e-4; what the rounding itself costs against the fp32-weight graph is reported and
    bounded by SURVEY.md §8d's estimate (2.4e-3 / 5.3e-3 / 1.2e-3 on its probe set)."""
    from basic_pitch_amd import Model

    x, r32, r64 = cases

    def bf16(a):
        u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)

    wq = dict(weights)
    for k in ("contour1_w", "contour2_w", "note1_w", "note2_w", "onset1_w", "onset2_w"):
        wq[k] = bf16(weights[k])
    q64 = O.forward(x, wq, np.float64)
    m = Model(max_windows=8, bf16_weights=True)
    got = m.predict(x)
    m.close()
    for k in ("note", "onset", "contour"):
        assert np.isfinite(got[k]).all()
        assert np.abs(got[k][:3] - q64[k][:3]).max() <= 1e-4, (k, np.abs(got[k][:3] - q64[k][:3]).max())
        cost = np.abs(got[k] - r64[k]).max()
        print(f"bf16 weights, {k}: max |out - fp32-weight graph| = {cost:.2e}")
        assert cost <= 1e-2, (k, cost)


def test_extended_cqt_44k_mode(weights):
    """BASELINE.json configs[4] (not a reference behaviour): 44.1 kHz windows of 87,688 samples
    through the 10-octave / 345-bin CQT (BP_FLAG_EXT_CQT_44K), CNN unchanged.  Parity is against the re-parametrised
    restatement in the oracle: end to end to the usual noise-aware 1e-4, and the track path (hop 72,328, lead-in 7,680)
    against host windowing of the same samples."""
    from basic_pitch_amd import Model

    rng = np.random.default_rng(21)
    x = rng.uniform(-1, 1, (3, O.EXT_AUDIO_N_SAMPLES)).astype(np.float32)
    t = np.arange(O.EXT_AUDIO_N_SAMPLES) / 44100.0
    x[2] = (0.4 * np.sin(2 * np.pi * 9000.0 * t) + 0.3 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    r64 = O.forward(x, weights, np.float64, ext=True)
    r32 = O.forward(x, weights, np.float32, ext=True)
    m = Model(max_windows=8, ext_cqt_44k=True)
    assert m.audio_n_samples == 87688 and m.sample_rate == 44100
    got = m.predict(x)
    _noise_aware(got, r32, r64)
    for k in ("note", "onset", "contour"):
        assert np.abs(got[k][:2] - r64[k][:2]).max() <= 1e-4, k
    with pytest.raises(ValueError):
        m.predict(np.zeros((1, 43844), np.float32))
    # whole track at 44.1 kHz: on-device windowing == host windowing + predict + unwrap with the doubled hop
    n = 72328 * 3 + 1234
    y = rng.uniform(-0.5, 0.5, n).astype(np.float32)
    tr = m.predict_track(y)
    n_win = -(-(n + 7680) // 72328)
    pad = np.concatenate([np.zeros(7680, np.float32), y, np.zeros(n_win * 72328 + 87688, np.float32)])
    wins = np.stack([pad[w * 72328 : w * 72328 + 87688] for w in range(n_win)])
    pw = m.predict(wins)
    T = int(n / 72328 * 142)
    for k in tr:
        ref = pw[k][:, 15:157].reshape(n_win * 142, -1)[:T]
        assert tr[k].shape == ref.shape and np.array_equal(tr[k], ref), k
    # 48 kHz stereo PCM is resampled to the handle's 44.1 kHz on the device
    pcm = rng.uniform(-0.5, 0.5, (48000, 2)).astype(np.float32)
    assert m.resample(pcm, 48000).shape == (44100,)
    # a launch with a window for every second CU takes the fused filterbank + normalise path (10 levels, 345 bins):
    # bit-identical to the strided path of the small handle
    xs = np.concatenate([x, x[::-1]] * 22)[:130]
    big = Model(max_windows=130, ext_cqt_44k=True)
    e = big.predict(xs)
    for k in got:
        assert np.array_equal(e[k][:3], got[k]) and np.array_equal(e[k][3:6], got[k][::-1]), k
    big.close()
    m.close()


@pytest.mark.parametrize("mode,batch", [("bf16_weights", 1024), ("ext_cqt_44k", 512)])
def test_modes_are_batch_invariant_at_their_bench_batches(mode, batch):
    """BASELINE.json configs[3] / configs[4] are benchmarked at B = 1024 / 512 and parity-tested above at a handful of
    windows: a window's result must not depend on which of the two handles computed it (the small one takes the strided
    filterbank + zpack launches and the wide decimators, the big one the per-window fused kernels; chunk counts of the
    marches differ).  Bit-equality of probe windows between `max_windows = 8` and the bench-size handle, as
    test_batch_invariance_and_chunking holds for the default mode."""
    from basic_pitch_amd import Model

    ext = mode == "ext_cqt_44k"
    n_s = O.EXT_AUDIO_N_SAMPLES if ext else O.AUDIO_N_SAMPLES
    rng = np.random.default_rng(31)
    x = rng.uniform(-1, 1, (batch, n_s)).astype(np.float32)
    x[5] = 0.0                                      # a silent window (range 0: divide_no_nan)
    t = np.arange(n_s) / (44100.0 if ext else 22050.0)
    x[6] = (0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.01 * rng.standard_normal(n_s)).astype(np.float32)
    x[batch - 1] *= 0.01
    idx = [0, 5, 6, batch // 2, batch - 1]
    big = Model(max_windows=batch, **{mode: True})
    a = big.predict(x)
    big.close()
    small = Model(max_windows=8, **{mode: True})
    b = small.predict(x[idx])
    c = small.predict(x[: 3 * 8 + 5])               # four chunks incl. a ragged tail
    small.close()
    for k in a:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k][idx], b[k]), (mode, k)
        assert np.array_equal(a[k][: 3 * 8 + 5], c[k]), (mode, k)
    # the silent window is the constant map of divide_no_nan in every mode: away from the window's ends (the
    # convolutions' zero padding in time) all frames are alike
    for k in a:
        assert np.ptp(a[k][5][16:156], axis=0).max() <= 1e-6, (mode, k)


def test_integration_md_binding_runs_verbatim(weights):
    """INTEGRATION.md section 1 is the ctypes stub a maintainer of the reference would paste into inference.py: the code
    block is executed here exactly as printed, against the built library and the shipped weights blob, and must return
    what Model.predict returns (same bits: same library, same entry point)."""
    import re

    from basic_pitch_amd import Model, build

    md = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "INTEGRATION.md")).read()
    sec = md[md.index("## 1. The binding"):md.index("## 2.")]
    (code,) = re.findall(r"```python\n(.*?)```", sec, flags=re.S)
    ns = {}
    exec(compile(code, "INTEGRATION.md#1", "exec"), ns)
    blob = os.path.join(os.path.dirname(build.LIB_PATH), "..", "assets", "nmp_weights.bin")
    stub = ns["_MI355X"](build.build_library(), blob, 0, 16)
    x = make_windows("uniform", 3, seed=41)
    note, onset, contour = stub.run(x[:, :, None])   # the reference's (n, 43844, 1)
    m = Model(max_windows=16)
    ref = m.predict(x)
    m.close()
    assert note.flags.writeable and note.flags.c_contiguous and note.dtype == np.float32
    assert np.array_equal(note, ref["note"]) and np.array_equal(onset, ref["onset"]) and np.array_equal(contour, ref["contour"])
    r64 = O.forward(x, weights, np.float64)
    assert np.abs(note - r64["note"]).max() <= 2e-4
    # an unloadable model is a ValueError, as for the reference's other loaders (inference.py:148-154)
    bad = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conftest.py")
    with pytest.raises(ValueError):
        ns["_MI355X"](build.build_library(), bad)
    del stub


def _noise_aware(got, r32, r64, floor=1e-4, factor=2.0):
    """|hip - fp64| <= max(1e-4, 2 * |fp32 oracle - fp64|), per tensor (SURVEY.md §8c Tier A).

    Two fp32-class evaluations of this graph are two draws of the same heavy-tailed noise: the per-window MINIMUM of
    the log-power (signal.py:177) sits on a cancellation-noise bin for tonal input and shifts every output of the
    window.  Measured distribution on MI355X, profiles/r02_parity.md (46 windows: |hip - fp64|, |hip - fp32 oracle|,
    |fp32 oracle - fp64| per window): noise-like windows 3e-6..4e-5 (fp32 oracle: 3e-6..7e-5), the reference clip's
    windows <= 2.7e-5 (fp32 oracle <= 1.4e-4), 32 tonal windows median 1.0e-4 / max 1.5e-3 (fp32 oracle: median 2.6e-4 /
    max 2.3e-3) — the HIP path is the CLOSER of the two to fp64, and all 46 windows sit inside the factor-2 bound."""
    for k in ("note", "onset", "contour"):
        ours = np.abs(got[k] - r64[k]).max()
        orc = np.abs(r32[k] - r64[k]).max()
        assert ours <= max(floor, factor * orc), (k, ours, orc)


def test_tonal_windows_distribution(weights):
    """Tonal input (music) is where fp32 evaluations of this graph scatter most (the per-window minimum of the log-power
    sits on a cancellation-noise bin).  Over 32 tonal windows + a loud and a quiet 440 Hz sine the path is held to the
    fp32 oracle's own distribution: worst window within 2x the oracle's worst, median distance to fp64 not above the
    oracle's, SURVEY.md 8c's per-window bound max(1e-4, 2 x the fp32 oracle's own distance) on all but at most one of the
    34 windows (and that one within 1.5 x the bound) — the two evaluations are independent draws of a heavy-tailed noise, so a per-window bound on EVERY window
    holds only for a lucky summation order (it did for the round-2 CQT kernels, it does not for the round-3 ones; see
    profiles/r03_parity_bench_batch.md for the same effect on white noise).  |hip - fp32 oracle| is printed (SURVEY.md
    §8c asks for it) — it is the sum of two independent noises and not a parity criterion."""
    from basic_pitch_amd import Model

    t = np.arange(43844) / 22050.0
    x = np.concatenate([make_windows("tones", 32, 2), (0.5 * np.sin(2 * np.pi * 440.0 * t))[None].astype(np.float32),
                        (0.01 * np.sin(2 * np.pi * 440.0 * t))[None].astype(np.float32)])
    m = Model(max_windows=64)
    got = m.predict(x)
    m.close()
    r32 = O.forward(x, weights, np.float32)
    r64 = O.forward(x, weights, np.float64)
    d = lambda a, b, i: max(float(np.abs(a[k][i] - b[k][i]).max()) for k in ("note", "onset", "contour"))  # noqa: E731
    h64 = np.asarray([d(got, r64, i) for i in range(len(x))])
    o64 = np.asarray([d(r32, r64, i) for i in range(len(x))])
    h32 = np.asarray([d(got, r32, i) for i in range(len(x))])
    print(f"tonal: |hip-fp64| median {np.median(h64):.2e} max {h64.max():.2e}; |fp32-fp64| median {np.median(o64):.2e} "
          f"max {o64.max():.2e}; |hip-fp32| median {np.median(h32):.2e} max {h32.max():.2e}")
    bound = np.maximum(1e-4, 2.0 * o64)
    print(f"tonal: within max(1e-4, 2 x fp32 oracle): {(h64 <= bound).sum()}/{len(x)}; worst ratio {(h64 / bound).max():.2f}")
    # the gate sits at the measured level (33 / 34 windows inside the bound, worst ratio 1.28: rounds 3 and 4), not at a
    # loose fraction: at most ONE of the 34 windows outside, and by no more than a factor 1.5
    outside = int((h64 > bound).sum())
    assert outside <= 1, (outside, h64, o64)
    assert (h64 / bound).max() <= 1.5, (h64 / bound).max()
    assert h64.max() <= 2.0 * o64.max(), (h64.max(), o64.max())
    assert np.median(h64) <= np.median(o64)
    assert h64[33] <= 1e-4  # the quiet sine: no cancellation floor, plain 1e-4


def test_end_to_end_synthetic(runner, cases):
    x, r32, r64 = cases
    got = runner.model.predict(x)
    _noise_aware(got, r32, r64)
    # these three noise-like windows meet 1e-4 outright (the whole bench batch: test_bench_batch_parity)
    for k in ("note", "onset", "contour"):
        assert np.abs(got[k][:3] - r64[k][:3]).max() <= 1e-4, k
    # reference I/O contract: fresh, writable, C-contiguous float32 (note_creation.py:338-341 mutates)
    for k, shape in (("note", (4, 172, 88)), ("onset", (4, 172, 88)), ("contour", (4, 172, 264))):
        a = got[k]
        assert a.shape == shape and a.dtype == np.float32 and a.flags.c_contiguous and a.flags.writeable
    got3 = runner.model.predict(x[:, :, None])  # (n, 43844, 1) like the reference
    for k in got:
        assert np.array_equal(got[k], got3[k])


def test_device_path_equals_host_path(runner, cases):
    x = cases[0]
    a = runner.model.predict(x)
    b = runner.model.predict(torch.from_numpy(x).cuda())
    for k in a:
        assert b[k].is_cuda and np.array_equal(a[k], b[k].cpu().numpy())


def test_stream_switch_orders_the_shared_workspace(cases):
    """One handle, asynchronous calls issued alternately on two torch streams without synchronising in between: the
    workspace is shared, so bp_set_stream has to order the new stream after the old one's queued work."""
    from basic_pitch_amd import Model

    m = Model(max_windows=4)
    xs = [torch.from_numpy(np.roll(cases[0], i, axis=0).copy()).cuda() for i in range(4)]
    refs = [{k: v.clone() for k, v in m.predict(x).items()} for x in xs]
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for rep in range(3):
        for i, x in enumerate(xs):
            with torch.cuda.stream(s1 if (i + rep) % 2 else s2):
                outs.append((i, m._predict_device(x, sync=False)))
    torch.cuda.synchronize()
    for i, o in outs:
        for k in o:
            assert torch.equal(o[k], refs[i][k]), (i, k)
    m.close()


def test_batch_invariance_and_chunking(weights):
    """Windows are independent units: a window's result must not depend on its batch or position,
    nor on how the library chunks a large batch (BASELINE.json configs[1] size, B = 256)."""
    from basic_pitch_amd import Model

    x = make_windows("uniform", 256, seed=11)
    x[7] = 0.0  # a silent window in the middle (divide_no_nan path)
    x[8] = make_windows("tones", 1, 5)[0]
    big = Model(max_windows=256)
    small = Model(max_windows=48)  # forces 6 chunks incl. a ragged tail
    a = big.predict(x)
    b = small.predict(x)
    for k in a:
        assert np.isfinite(a[k]).all()
        assert np.array_equal(a[k], b[k]), k
    perm = np.random.default_rng(0).permutation(256)
    c = big.predict(x[perm])
    for k in a:
        assert np.array_equal(a[k][perm], c[k]), k
    idx = [0, 7, 8, 100, 255]
    d = big.predict(x[idx])
    for k in a:
        assert np.array_equal(a[k][idx], d[k]), k
    # spot-check values at full size against the fp64 oracle
    r64 = O.forward(x[idx], weights, np.float64)
    r32 = O.forward(x[idx], weights, np.float32)
    _noise_aware(d, r32, r64)
    big.close(), small.close()
    # more windows than CUs in one launch: the fused filterbank's workgroups then own several windows each, and the
    # decimators' queues wrap (the round-3 CQT kernels) — still the same bits
    many = Model(max_windows=600)
    x600 = np.concatenate([x, x[::-1], x[:88]])
    e = many.predict(x600)
    for k in a:
        assert np.array_equal(e[k][:256], a[k]) and np.array_equal(e[k][256:512], a[k][::-1]) and np.array_equal(e[k][512:], a[k][:88]), k
    many.close()


def test_edge_cases():
    from basic_pitch_amd import Model

    m = Model(max_windows=4)
    e = m.predict(np.zeros((0, 43844), np.float32))
    assert e["note"].shape == (0, 172, 88) and e["contour"].shape == (0, 172, 264)
    with pytest.raises(ValueError):
        m.predict(np.zeros((2, 1000), np.float32))
    with pytest.raises(ValueError):
        m.predict(np.zeros((43844,), np.float32))
    z = m.predict(np.zeros((1, 43844), np.float32))
    for k in z:  # silent window: finite, and constant along time away from the window edges
        assert np.isfinite(z[k]).all() and np.ptp(z[k][0, 20:150, :], axis=0).max() < 1e-6
    big = m.predict(np.full((1, 43844), 1.0, np.float32))  # DC input
    assert all(np.isfinite(v).all() for v in big.values())
    # tracks: empty, shorter than one hop, exactly one hop
    for n in (0, 1, 5000, 36164, 36165):
        r = m.predict_track(np.zeros(n, np.float32) + 0.01)
        T = min(int(np.ceil((n + 3840) / 36164)) * 142, int(n / 36164 * 142)) if n else 0
        assert r["note"].shape == (T, 88) and r["contour"].shape == (T, 264)
    m.close()


def test_time_dominant_flag_times_one_stage_and_changes_nothing():
    """BP_FLAG_TIME_DOMINANT (what bench.py's timed steps use): events around the folded contour conv1 only; the
    outputs are bit-identical to an untimed and to a fully timed handle."""
    from basic_pitch_amd import Model

    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (3, 43844)).astype(np.float32)
    ref = None
    for kw in ({}, {"time_dominant": True}, {"stage_timing": True}):
        m = Model(max_windows=4, **kw)
        out = m.predict(x)
        if ref is None:
            ref = out
            with pytest.raises(ValueError):
                m.stage_ms()  # BP_ERR_UNSUPPORTED: no timing flag
        else:
            for k in ref:
                assert np.array_equal(out[k], ref[k]), (kw, k)
            ms = m.stage_ms()
            assert ms["contour_conv1"] > 0.0
            timed = {k for k, v in ms.items() if v > 0.0}
            if "time_dominant" in kw:
                assert timed == {"contour_conv1"}, timed
            else:
                assert {"pyramid", "filterbank", "zpack", "contour_conv2", "note", "onset"} <= timed
        m.close()


def test_new_entry_points_reject_bad_arguments():
    """C-ABI error behaviour of the round's new entry points: invalid arguments come back as BP_ERR_INVALID_ARG /
    BP_ERR_UNSUPPORTED (ValueError / NativeLibraryError on the Python side), never as a crash."""
    import ctypes as C

    from basic_pitch_amd import Model, _native
    from basic_pitch_amd._native import NativeLibraryError

    with pytest.raises((ValueError, NativeLibraryError)):
        Model(exact_f32_mfma=True, ext_cqt_44k=True)  # extended range exists only on the split-precision path
    with pytest.raises(ValueError):
        Model(max_windows=1 << 20)  # above BP_MAX_WINDOWS_PER_CHUNK: a clear error instead of a failed launch later
    m = Model(max_windows=4)
    lib, h = m._lib, m._handle
    assert lib.bp_infer_tracks(h, -1, None, None, None, None, None, 0) == _native.BP_ERR_INVALID_ARG
    assert lib.bp_infer_tracks(h, 2, None, None, None, None, None, 0) == _native.BP_ERR_INVALID_ARG
    assert lib.bp_infer_tracks(h, 0, None, None, None, None, None, 0) == _native.BP_OK
    buf = np.zeros(10, np.float32)
    assert lib.bp_resample(h, buf.ctypes.data, 10, 0, 44100, buf.ctypes.data, 0) == _native.BP_ERR_INVALID_ARG   # channels
    assert lib.bp_resample(h, buf.ctypes.data, 10, 1, 500, buf.ctypes.data, 0) == _native.BP_ERR_INVALID_ARG     # rate
    out = np.zeros(5, np.float32)
    assert lib.bp_resample(h, buf.ctypes.data, 10, 1, 44101, out.ctypes.data, 0) == _native.BP_OK  # irregular ratio
    assert lib.bp_infer_pcm(h, buf.ctypes.data, 10, 1, 44100, None, None, None, 7) == _native.BP_ERR_INVALID_ARG  # mem_kind
    assert b"ingest" in lib.bp_last_error(h)
    assert m.predict_tracks([np.zeros(0, np.float32)])[0]["note"].shape == (0, 88)
    m.close()


def test_track_path_equals_windowed_path_and_golden(weights, clip_22k):
    """Config 1 of BASELINE.json = the reference's own known-answer test (tests/test_inference.py:43-70): its
    44.1 kHz clip end to end — host WAV decode, then downmix, soxr_hq-design resampling, windowing, CQT + CNN and
    un-overlapping on the device — against its golden posteriorgrams at ITS tolerance, atol = 1e-4 on every element.
    The on-device windowing / un-overlapping path must also equal the reference-structured per-window path."""
    from basic_pitch_amd import Model, audio as A, inference as inf

    m = Model()
    wav = os.path.join(GOLDEN, "vocadito_10.wav")
    a = inf.run_inference(wav, m)
    b = inf.run_inference_windowed(wav, m)
    g = np.load(os.path.join(GOLDEN, "vocadito_10_model_output.npz"))
    for k in ("note", "onset", "contour"):
        assert a[k].shape == g[k].shape
        # device resampler (fp64 accumulate) vs host resampler (upfirdn): the same fp32 samples up to 1 ulp, which the
        # per-window normalisation turns into <= 1e-5 on the posteriorgrams
        assert np.abs(a[k] - b[k]).max() <= 2e-5, (k, np.abs(a[k] - b[k]).max())
        for got in (a[k], b[k]):
            assert np.abs(got - g[k]).max() <= 1e-4, (k, np.abs(got - g[k]).max())
            assert np.abs(got - g[k]).mean() <= 2e-6
    # bit-for-bit: device windowing / un-overlapping of given 22.05 kHz samples == host windowing + predict + unwrap
    y = m.resample(*A.read_wav(wav))
    t = m.predict_track(y)
    wins, n_orig = O.window_track(y)
    pw = m.predict(wins)
    for k in t:
        assert np.array_equal(t[k], O.unwrap_output(pw[k], n_orig)), k
    for k in ("note", "onset", "contour"):
        assert np.array_equal(t[k], a[k]), k
    # and tightly against the oracle on the identical 22.05 kHz samples
    r64 = O.run_track(clip_22k, weights, np.float64, batch=6)
    r32 = O.run_track(clip_22k, weights, np.float32, batch=6)
    _noise_aware(a, r32, r64)
    m.close()


def test_multi_track_packing_is_bit_identical():
    """bp_infer_tracks packs the windows of consecutive tracks into full batches; every track must come out exactly
    as from its own bp_infer_track call (ragged lengths, an empty track, tracks spanning several chunks)."""
    from basic_pitch_amd import Model

    m = Model(max_windows=16)
    rng = np.random.default_rng(5)
    lens = [50000, 0, 36164 * 3 + 17, 1, 36164 * 20, 7000, 36164 * 16 - 3840]
    tracks = [rng.uniform(-0.5, 0.5, n).astype(np.float32) for n in lens]
    packed = m.predict_tracks(tracks)
    for t, got in zip(tracks, packed):
        ref = m.predict_track(t)
        for k in ref:
            assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), (len(t), k)
    dev = m.predict_tracks([torch.from_numpy(t).cuda() for t in tracks if len(t)])
    for t, got in zip([t for t in tracks if len(t)], dev):
        ref = m.predict_track(t)
        for k in ref:
            assert np.array_equal(got[k].cpu().numpy(), ref[k]), k
    assert m.predict_tracks([]) == []
    m.close()
    # more pieces per chunk than one windowing / un-overlapping launch describes (kMaxTrackSegs = 16): 40 one- and
    # two-window tracks in chunks of 64 windows
    m = Model(max_windows=64)
    lens = [int(n) for n in rng.integers(1, 60000, 40)]
    tracks = [rng.uniform(-0.5, 0.5, n).astype(np.float32) for n in lens]
    for t, got in zip(tracks, m.predict_tracks(tracks)):
        ref = m.predict_track(t)
        for k in ref:
            assert got[k].shape == ref[k].shape and np.array_equal(got[k], ref[k]), (len(t), k)
    m.close()


def test_device_audio_ingest(weights):
    """downmix + resampling on the device.  bp_resample against oracle/soxr_oracle.py — the
    direct-form float64 restatement of libsoxr's SOXR_HQ design, itself pinned by the reference's golden posteriorgrams
    (tests/test_oracle_golden.py) — for the rate pairs that occur in practice, mono and multi-channel, including a
    ratio irregular enough for the evaluate-in-place kernel; bp_infer_pcm must equal resample-then-bp_infer_track."""
    from basic_pitch_amd import Model, audio as A
    from oracle import soxr_oracle as S

    m = Model(max_windows=8)
    rng = np.random.default_rng(3)
    for sr, ch, n in ((44100, 2, 150001), (48000, 1, 9600), (16000, 3, 4000), (8000, 1, 999), (22050, 2, 50000),
                      (32000, 1, 1), (11025, 2, 30000), (88200, 1, 40000), (10004, 1, 600)):
        pcm = rng.uniform(-1, 1, (n, ch)).astype(np.float32)
        t = np.arange(n) / sr
        pcm[:, 0] += 0.5 * np.sin(2 * np.pi * 440.0 * t).astype(np.float32)
        mono = pcm.mean(axis=1, dtype=np.float32) if ch > 1 else pcm[:, 0]
        ref = S.resample(mono, sr, 22050)
        got = m.resample(pcm, sr)
        assert got.shape == ref.shape == (int(np.ceil(n * 22050 / sr)),), (sr, got.shape, ref.shape)
        assert np.abs(got - ref).max() <= 1e-6, (sr, ch, np.abs(got - ref).max())
        assert np.abs(got - A.resample(np.ascontiguousarray(mono), sr)).max() <= 1e-6, sr  # host copy of the design
    # whole-path equivalence on the reference's 44.1 kHz clip
    pcm, sr = A.read_wav(os.path.join(GOLDEN, "vocadito_10.wav"))
    a = m.predict_pcm(pcm, sr)
    b = m.predict_track(m.resample(pcm, sr))
    for k in a:
        assert a[k].shape[0] == 787 and np.array_equal(a[k], b[k]), k
    with pytest.raises(ValueError):
        m.resample(pcm, 0)
    e = m.predict_pcm(np.zeros((0, 2), np.float32), 44100)
    assert e["note"].shape == (0, 88)
    m.close()


def test_predict_parameter_sweeps_of_the_reference():
    """tests/test_inference.py:105-161 of the reference: predict() over its sweeps of onset_threshold, frame_threshold,
    minimum_note_length, minimum_frequency and maximum_frequency on its clip — its property assertions, and beyond them
    the same events as the numpy restatement of note_creation.py decodes from the same posteriorgrams."""
    import warnings

    from basic_pitch_amd import inference as inf
    from oracle import note_oracle as NO

    wav = os.path.join(GOLDEN, "vocadito_10.wav")
    model = inf.Model(max_windows=8)
    base = inf.run_inference(wav, model)

    def check(**kw):
        out, _, events = inf.predict(wav, model, **kw)
        for k in base:  # constrain_frequency zeroes columns of the returned maps in place, like the reference
            if "minimum_frequency" not in kw and "maximum_frequency" not in kw:
                assert np.array_equal(out[k], base[k]), k
        okw = dict(onset_thresh=kw.get("onset_threshold", 0.5), frame_thresh=kw.get("frame_threshold", 0.3),
                   min_note_len=int(np.round(kw.get("minimum_note_length", inf.DEFAULT_MINIMUM_NOTE_LENGTH_MS) / 1000 * (22050 / 256))),
                   min_freq=kw.get("minimum_frequency"), max_freq=kw.get("maximum_frequency"))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref, _ = NO.model_output_to_notes({k: v.copy() for k, v in base.items()}, **okw)
        assert len(events) == len(ref), (kw, len(events), len(ref))
        for a, b in zip(events, ref):
            assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2] and list(a[4]) == list(b[4]), kw
            assert np.float32(a[3]).tobytes() == np.float32(b[3]).tobytes(), kw
        return events

    hz_to_midi = lambda f: 12.0 * (np.log2(f) - np.log2(440.0)) + 69.0
    for v in (0, 0.3, 0.8, 1):
        check(onset_threshold=v)
        check(frame_threshold=v)
    for ms in (10, 100, 1000):
        ev = check(minimum_note_length=ms)
        assert all(n[1] - n[0] >= ms / 1000.0 for n in ev)
    for f in (40, 80, 200, 2000):
        ev = check(minimum_frequency=f)
        assert all(n[2] >= np.round(hz_to_midi(f)) for n in ev)
        ev = check(maximum_frequency=f)
        assert all(n[2] <= np.round(hz_to_midi(f)) for n in ev)
    model.close()


def test_model_path_is_loaded_once_per_thread():
    """predict(path) with a model PATH (the reference's default call): the loaded model is kept and reused by later calls
    of the same thread, another thread gets its own (a handle is not for two threads at once) — which dies with that
    thread —, same results either way."""
    import gc
    import threading
    import weakref

    from basic_pitch_amd import inference as inf

    wav = os.path.join(GOLDEN, "vocadito_10.wav")
    inf.clear_model_cache()
    a, _, ev_a = inf.predict(wav)
    assert len(inf._MODEL_CACHE.models) == 1
    first = next(iter(inf._MODEL_CACHE.models.values()))
    b, _, ev_b = inf.predict(wav, inf.ICASSP_2022_MODEL_PATH)
    assert len(inf._MODEL_CACHE.models) == 1 and next(iter(inf._MODEL_CACHE.models.values())) is first
    got = {}

    def other():
        got["out"] = inf.predict(wav)
        (theirs,) = inf._MODEL_CACHE.models.values()
        got["is_own"] = theirs is not first
        got["ref"] = weakref.ref(theirs)

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert got["is_own"] and len(inf._MODEL_CACHE.models) == 1
    del t
    gc.collect()
    assert got["ref"]() is None  # the other thread's model went with the thread (0.8 GB of device buffers)
    for k in a:
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], got["out"][0][k]), k
    assert len(ev_a) == len(ev_b) == len(got["out"][2]) == 28
    with pytest.raises(ValueError):
        inf.predict(wav, os.path.join(GOLDEN, "no_such_model.onnx"))
    inf.clear_model_cache()
    assert len(inf._MODEL_CACHE.models) == 0


def _ab_env(**switches):
    """Environment of a subprocess that runs an A/B variant of a kernel: the switches (BP_ONSET, BP_CONV1, BP_RESAMPLE, ...)
    only exist in the A/B library (build.py: build_library(ab=True), -DBP_AB_KERNELS), which is built here if the tree
    does not hold a current one; with no switches, the product library."""
    e = dict(os.environ)
    for k in ("BASIC_PITCH_AMD_LIB", "BP_ONSET", "BP_NOTE", "BP_CONV1", "BP_CONV2", "BP_RIM", "BP_RESAMPLE", "BP_CONTOUR_PARTS"):
        e.pop(k, None)
    if switches:
        from basic_pitch_amd import build as B

        e["BASIC_PITCH_AMD_LIB"] = B.build_library(ab=True)
        e.update(switches)
    return e


def test_resample_kernels_agree_bit_for_bit(tmp_path):
    """The resampler's three kernels — one thread per output, the LDS-tiled one, the 2 : 1 register-window one — add the
    same products in the same order: identical float32 signals, at the signal's edges too (a 2 : 1 length that is not a
    multiple of the block's 1024 outputs, a signal shorter than the filter).  One process per kernel: the product library
    (automatic choice) and the A/B library with BP_RESAMPLE=plain / tiled."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "experiments", "resample_ab.py")
    got = {}
    for mode in ("plain", "tiled", "auto"):
        out = str(tmp_path / f"{mode}.npz")
        subprocess.run([sys.executable, tool, out], check=True, timeout=600,
                       env=_ab_env() if mode == "auto" else _ab_env(BP_RESAMPLE=mode))
        got[mode] = np.load(out)
    for k in got["plain"].files:
        for mode in ("tiled", "auto"):
            assert np.array_equal(got["plain"][k].view(np.uint32), got[mode][k].view(np.uint32)), (k, mode)


def test_predict_note_events_match_reference_golden(tmp_path):
    """BASELINE.json north star: MIDI note events identical to the reference's on its test clip.
    `predict()` end to end (WAV decode on the host; downmix, resampling, CQT + CNN on the MI355X; note decoding in C++)
    against the reference's golden `note_events.npz`: all 28 events, every discrete field exact (start / end
    time, pitch, pitch-bend list), amplitude within the reference's own tolerance (tests/test_inference.py:
    73-76, atol 1e-4)."""
    from basic_pitch_amd import inference as inf

    wav = os.path.join(GOLDEN, "vocadito_10.wav")
    model_output, midi, events = inf.predict(wav)
    g = np.load(os.path.join(GOLDEN, "vocadito_10_note_events.npz"))
    assert len(events) == len(g["pitch"]) == 28
    for i, e in enumerate(events):
        assert e[0] == g["start_s"][i] and e[1] == g["end_s"][i] and e[2] == g["pitch"][i], i
        assert abs(float(e[3]) - float(g["amplitude"][i])) <= 1e-4, i
        assert list(e[4]) == list(g["bend_values"][g["bend_offsets"][i] : g["bend_offsets"][i + 1]]), i
    assert set(model_output) == {"note", "onset", "contour"}
    assert len(midi.instruments) == 1 and len(midi.instruments[0].notes) == 28
    # the remaining assertions of the reference's test_predict (tests/test_inference.py:50-65): shapes, the supported
    # pitch range, and the model output's length against the audio's duration through model_frames_to_time
    from basic_pitch_amd import audio as A, note_creation as NC

    assert model_output["note"].shape == model_output["onset"].shape and isinstance(events, list)
    assert all(21 <= e[2] <= 21 + 88 for e in events)
    last_frame_s = NC.model_frames_to_time(model_output["note"].shape[0])[-1]
    assert abs(last_frame_s - A.get_duration(wav)) <= 2 * (256 / 22050)
    # predict_and_save writes the reference's four artefacts (inference.py:565-602)
    inf.predict_and_save([wav], tmp_path, True, True, True, True)
    stem = tmp_path / "vocadito_10_basic_pitch"
    assert stem.with_suffix(".mid").read_bytes() == open(os.path.join(GOLDEN, "midi", "clip_default.mid"), "rb").read()
    assert stem.with_suffix(".wav").read_bytes()[:4] == b"RIFF"  # the sonified MIDI (inference.py:588-594)
    saved = np.load(stem.with_suffix(".npz"), allow_pickle=True)["basic_pitch_model_output"].item()
    assert saved["note"].shape == (787, 88)
    assert len(stem.with_suffix(".csv").read_text().strip().splitlines()) == 29


def test_second_clip_note_events(weights):
    """A second recording (the reference's tests/resources/vocadito_14.wav, stored losslessly as FLAC): predict() on the
    MI355X — native FLAC decode, device resampling, CQT + CNN, C++ note decoding — against the note events the
    unmodified reference note_creation produces from the fp64 oracle chain (tools/make_second_clip_fixture.py; the fp32
    oracle decodes to the same events, so none of them sits on a threshold): every discrete field equal, amplitudes
    within 1e-4, posteriorgram statistics within 1e-4."""
    from basic_pitch_amd import inference as inf

    g = np.load(os.path.join(GOLDEN, "vocadito_14_expected.npz"))
    assert bool(g["fp32_oracle_agrees"][0])
    out, midi, events = inf.predict(os.path.join(GOLDEN, "vocadito_14.flac"))
    assert len(events) == len(g["pitch"]) == 26
    for i, e in enumerate(events):
        assert e[0] == g["start_s"][i] and e[1] == g["end_s"][i] and e[2] == g["pitch"][i], i
        assert abs(float(e[3]) - float(g["amplitude"][i])) <= 1e-4, i
        assert list(e[4]) == list(g["bend_values"][g["bend_offsets"][i] : g["bend_offsets"][i + 1]]), i
    for k in ("note", "onset", "contour"):
        assert np.abs(out[k].mean(axis=0) - g[f"{k}_colmean"]).max() <= 1e-4, k
        assert np.abs(out[k].max(axis=1) - g[f"{k}_rowmax"]).max() <= 1e-4, k


def test_predict_many_equals_per_file_predict(tmp_path):
    """predict_many (device resampling, windows packed across files, threaded note decoding) returns, in input order,
    exactly what predict() returns file by file — and so the reference's 28 golden note events for the golden clip."""
    import shutil

    from basic_pitch_amd import Model
    from basic_pitch_amd.inference import predict, predict_many

    clip = os.path.join(GOLDEN, "vocadito_10.wav")
    paths = []
    for i in range(3):
        q = tmp_path / f"clip_{i}.wav"
        shutil.copy(clip, q)
        paths.append(q)
    m = Model(max_windows=8)
    many = predict_many(paths, m, group=2, decode_threads=2)
    assert len(many) == 3
    ref_out, _, ref_events = predict(paths[0], m)
    assert len(ref_events) == 28
    for out, midi, events in many:
        for k in ref_out:
            assert np.array_equal(out[k], ref_out[k]), k
        assert len(events) == len(ref_events)
        for a, b in zip(events, ref_events):
            assert a[:4] == b[:4] and list(a[4] or []) == list(b[4] or [])
        assert midi.get_end_time() > 0
    m.close()


def test_predict_many_sharded_on_one_gpu_and_flac_input(tmp_path):
    """The product's multi-GPU entry point with world size 1 (this box has one GPU): same results as predict() file by
    file, a broken file reported in place; and a FLAC copy of the reference's clip (test-side encoder, native decoder)
    gives bit-identical posteriorgrams and the reference's 28 golden note events."""
    import shutil

    import flac_writer as FW
    from basic_pitch_amd import Model, audio as A, predict_many_sharded
    from basic_pitch_amd.inference import predict

    clip = os.path.join(GOLDEN, "vocadito_10.wav")
    pcm, sr = A.read_wav(clip)
    flac = tmp_path / "clip.flac"
    flac.write_bytes(FW.encode(np.round(pcm * 32768.0).astype(np.int64), sr, 16, blocksize=4096))
    wav2 = tmp_path / "clip_copy.wav"
    shutil.copy(clip, wav2)
    broken = tmp_path / "broken.wav"
    broken.write_bytes(b"RIFF\x00\x00\x00\x00WAVEjunk")
    paths = [clip, str(flac), str(broken), str(wav2)]
    res = predict_many_sharded(paths, gpus=1, group=2, decode_threads=2)
    assert len(res) == 4 and isinstance(res[2], Exception)
    m = Model(max_windows=8)
    ref_out, _, ref_events = predict(clip, m)
    assert len(ref_events) == 28
    for i in (0, 1, 3):
        out, midi, events = res[i]
        for k in ref_out:
            assert np.array_equal(out[k], ref_out[k]), (i, k)
        assert [(e[0], e[1], e[2], list(e[4] or [])) for e in events] == [(e[0], e[1], e[2], list(e[4] or [])) for e in ref_events]
    m.close()


def test_ort_shim_session_runs_reference_call_pattern(cases):
    """The reference's ONNX leg, verbatim call pattern (inference.py:134-136, 173-180), against the shim.  The session
    takes the reference's nmp.onnx (structure-checked, constants extracted at load — covered on CPU where the
    reference checkout exists, tests/test_host_cpu.py) or this package's pre-extracted blob; the GPU box has no
    reference checkout, so the blob stands in here.  Outputs come back in the requested order."""
    import basic_pitch_amd.ort_shim as ort
    from basic_pitch_amd.inference import ICASSP_2022_MODEL_PATH

    x, r32, r64 = cases
    with pytest.raises(ValueError):
        ort.InferenceSession(b"not a model at all", providers=ort.get_available_providers())
    sess = ort.InferenceSession(str(ICASSP_2022_MODEL_PATH), providers=ort.get_available_providers())
    res = sess.run(
        ["StatefulPartitionedCall:1", "StatefulPartitionedCall:2", "StatefulPartitionedCall:0"],
        {"serving_default_input_2:0": x[:, :, None]},
    )
    assert [r.shape for r in res] == [(4, 172, 88), (4, 172, 88), (4, 172, 264)]
    for got, k in zip(res, ("note", "onset", "contour")):
        assert np.abs(got[:3] - r64[k][:3]).max() <= 1e-4, k


@pytest.mark.gpu
def test_long_file_split_by_window_range_equals_the_unsplit_file(tmp_path):
    """SURVEY.md 8e on hardware: a 50-second stereo 44.1 kHz take beside the 9-second reference clip in a job of two workers
    (both on this box's one GPU, each its own process and handle): the long file outweighs an even share, `plan_units` cuts
    it into two window ranges, each worker resamples the file on the device and computes its own windows, the parent
    concatenates and decodes.  Posteriorgrams and note events are BIT-EQUAL to `predict()` on the whole file (windows are
    independent, the library's results do not depend on the batching), and so are the short file's."""
    import wave

    from basic_pitch_amd import predict, predict_many_sharded
    from basic_pitch_amd.sharding import _file_costs, plan_units

    rng = np.random.default_rng(77)
    n = 50 * 44100
    t = np.arange(n) / 44100.0
    tone = 0.3 * np.sin(2 * np.pi * 220.0 * t) * (np.sin(2 * np.pi * 0.7 * t) > 0) + 0.2 * np.sin(2 * np.pi * 523.25 * t) * (np.sin(2 * np.pi * 0.45 * t) > 0)
    x = np.stack([tone + 0.01 * rng.standard_normal(n), 0.8 * tone + 0.01 * rng.standard_normal(n)], axis=1)
    long_wav = tmp_path / "long.wav"
    with wave.open(str(long_wav), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(44100)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
    paths = [os.path.join(GOLDEN, "vocadito_10.wav"), str(long_wav)]
    units, _ = plan_units(_file_costs(paths), 2)
    assert (1, 0, 2) in units and (1, 1, 2) in units
    whole = [predict(p) for p in paths]
    res = predict_many_sharded(paths, gpus=1, workers_per_gpu=2, decode_threads=2)
    for (mo_a, _, ev_a), (mo_b, _, ev_b) in zip(whole, res):
        for k in ("note", "onset", "contour"):
            assert mo_a[k].shape == mo_b[k].shape and np.array_equal(mo_a[k], mo_b[k]), k
        assert len(ev_a) == len(ev_b) and all(a[:4] == b[:4] and a[4] == b[4] for a, b in zip(ev_a, ev_b))
    assert len(whole[1][2]) > 20


@pytest.mark.gpu
def test_predict_and_save_sharded_two_workers_on_one_gpu(tmp_path):
    """The batch-job entry point on hardware: two worker processes share the GPU, each predicts and writes its own files;
    the MIDI of the reference clip is byte-identical to the committed pretty_midi-layout fixture whichever worker got it,
    and equals the single-process `predict_and_save` output."""
    import shutil

    from basic_pitch_amd import predict_and_save, predict_and_save_sharded

    clips = []
    for i in range(4):
        p = tmp_path / f"clip_{i}.wav"
        shutil.copy(os.path.join(GOLDEN, "vocadito_10.wav"), p)
        clips.append(str(p))
    out2, out1 = tmp_path / "sharded", tmp_path / "serial"
    out2.mkdir()
    out1.mkdir()
    rep = predict_and_save_sharded(clips, out2, True, False, False, True, gpus=1, workers_per_gpu=2, group=2)
    assert [r["n_note_events"] for r in rep] == [28, 28, 28, 28]
    predict_and_save(clips[:1], out1, True, False, False, True)
    want = open(os.path.join(GOLDEN, "midi", "clip_default.mid"), "rb").read()
    serial = open(out1 / "clip_0_basic_pitch.mid", "rb").read()
    assert serial == want
    for r in rep:
        assert open(r["outputs"]["midi"], "rb").read() == want
        assert open(r["outputs"]["note_events"]).read() == open(out1 / "clip_0_basic_pitch.csv").read()


def test_onset_march_equals_workgroup_kernel(tmp_path):
    """The three onset kernels are the same operator in different decompositions.  The 32x32x16 march (BP_ONSET=march32,
    onset_march.hip) and the round-2 workgroup kernel (BP_ONSET=ring, also the fp8 mode's kernel) use the same k-step
    order: bit-identical maps.  The default since round 4 (onset_march16.hip, 16x16x32 with the weights in registers)
    sums conv1's 200 products in another order (4 taps per matrix instruction instead of 2) and the head's 32 channels in
    one instruction instead of two: fp32 accumulation-order differences only, 2e-6 on a sigmoid output.  Random stack
    images and note maps through the C ABI stage hook, one process per kernel (the choice is read once per process)."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "experiments", "onset_ab.py")
    outs = {}
    for name, env in (("march16", {}), ("march32", {"BP_ONSET": "march32"}), ("ring", {"BP_ONSET": "ring"})):
        out = str(tmp_path / f"{name}.npy")
        subprocess.run([sys.executable, tool, out], check=True, env=_ab_env(**env), timeout=600)
        outs[name] = np.load(out)
    assert np.isfinite(outs["march16"]).all()
    assert np.array_equal(outs["march32"], outs["ring"])
    d = np.abs(outs["march16"] - outs["march32"]).max()
    assert d <= 2e-6, d


def test_contour_march_equals_round_kernel(tmp_path):
    """The contour conv1 interior as a vertical march on 16x16x32 (conv_contour_march.hip, the default since round 4) and
    as 256-position rounds on 32x32x16 (BP_CONV1=rounds) are the same folded operator: they sum the same 528 products
    per output in another order (32 taps per matrix instruction instead of 16, frame taps interleaved row by row), so
    the contour maps agree to fp32 accumulation-order noise: 2e-6 on a sigmoid output.  Random z through the C ABI stage
    hook, one process per kernel (the choice is read once per process)."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "experiments", "contour_ab.py")
    outs = {}
    for name, env in (("march", {}), ("rounds", {"BP_CONV1": "rounds"})):
        out = str(tmp_path / f"{name}.npy")
        subprocess.run([sys.executable, tool, out], check=True, env=_ab_env(**env), timeout=600)
        outs[name] = np.load(out)
    assert np.isfinite(outs["march"]).all()
    d = np.abs(outs["march"] - outs["rounds"]).max()
    assert d <= 2e-6, d


def test_rim_march_equals_rim_gemm(tmp_path):
    """The rim of contour conv1 with the weights resident in registers (conv_contour_rim_march.hip, the default for the
    309-bin CQT since round 5: 16-row blocks on 16x16x32, 14 k-steps of 32, z rows by LDS-DMA) and as the round-3 GEMM
    (conv_contour_rim.hip, BP_RIM=gemm in the A/B library: 32-row blocks on 32x32x16, 27 k-steps of 16) multiply the same
    per-side matrix with the same z rows: the contour maps agree to fp32 accumulation-order noise, on the rim bins and —
    through conv2's 5 x 5 window — their neighbours.  Random z through the C ABI stage hook, one process per kernel."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "experiments", "contour_ab.py")
    outs = {}
    for name, env in (("march", {}), ("gemm", {"BP_RIM": "gemm"})):
        out = str(tmp_path / f"{name}.npy")
        subprocess.run([sys.executable, tool, out], check=True, env=_ab_env(**env), timeout=600)
        outs[name] = np.load(out)
    assert np.isfinite(outs["march"]).all()
    d = np.abs(outs["march"] - outs["gemm"])
    assert d.max() <= 2e-6, d.max()
    assert d[..., 24:240].max() == 0.0  # away from the rim nothing changed


def test_conv2_projection_equals_the_vector_kernel(tmp_path):
    """Contour conv2 as a tap projection on the matrix cores (conv_contour2.hip contour_conv2_proj_kernel, the default since
    round 6: per input pixel 25 taps x 8 channels on v_mfma_f32_32x32x8_f16 with split operands, 25 additions per output) and
    the round-2 vector kernel (200 fp32 FMAs per output, BP_CONV2=valu in the A/B library) compute the same sums in another
    order: the contour maps of the whole contour stage (same conv1 kernels in both runs) agree to accumulation-order noise,
    at the rim bins (c1's zero pad columns), the first / last frames and the slab cuts too."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "experiments", "contour_ab.py")
    outs = {}
    for name, env in (("proj", {}), ("valu", {"BP_CONV2": "valu"})):
        out = str(tmp_path / f"{name}.npy")
        subprocess.run([sys.executable, tool, out], check=True, env=_ab_env(**env), timeout=600)
        outs[name] = np.load(out)
    assert np.isfinite(outs["proj"]).all()
    d = np.abs(outs["proj"] - outs["valu"]).max()
    assert d <= 2e-6, d


def test_note_march16_equals_note_march32(tmp_path):
    """The note branch on 16x16x32 (note_march16.hip, the default since round 6: one accumulator at scale 2^11, the
    activations' lo parts from a residual matrix instruction, conv2's vertical sum in lane) and the round-2 march on 32x32x16
    (note_march.hip, BP_NOTE=march32 in the A/B library) evaluate the same split-precision products of the same operands in
    another fp32 order: the note maps agree to accumulation-order noise, at the map's rims and at every cut of the frames
    into wave shares (5 windows: the end-to-end cut; 256 windows would take the aligned cut — covered by
    test_batch_invariance_and_chunking, which compares a 256-window launch with 7-window launches bit for bit).  Random
    contour maps through the C ABI stage hook, one process per kernel."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "experiments", "note_ab.py")
    outs = {}
    for name, env in (("march16", {}), ("march32", {"BP_NOTE": "march32"})):
        out = str(tmp_path / f"{name}.npy")
        subprocess.run([sys.executable, tool, out], check=True, env=_ab_env(**env), timeout=600)
        outs[name] = np.load(out)
    assert np.isfinite(outs["march16"]).all()
    d = np.abs(outs["march16"] - outs["march32"]).max()
    assert d <= 2e-6, d


def test_device_note_candidates_give_the_host_decoders_events(tmp_path):
    """The dense half of note decoding on the device (csrc/note_device.hip: constrain_frequency, inferred onsets, peak
    picking + threshold as a bitmap, the pitch bends of every (frame, bin); note_creation.py:289-343, 394-402, 182-219)
    against its numpy restatement — bit-equal note map, bitmap and bend map on the 16 reference-generated cases and on
    seeded fuzz maps — and, decoded by the host half, the events of bp_notes_decode bit for bit.  NaN maps and thresholds
    <= 0 are reported back (status 1) for the map decoder."""
    import sys

    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import note_cases
    from basic_pitch_amd import Model, note_creation as NC
    from oracle import note_oracle as NO

    m = Model(max_windows=8)

    def check(out, args):
        pb = args.get("include_pitch_bends", True)
        prm = NC._note_params(args["onset_thresh"], args["frame_thresh"], args.get("min_note_len", 11),
                              args.get("infer_onsets", True), args.get("max_freq"), args.get("min_freq"),
                              args.get("melodia_trick", True), NC.ENERGY_TOLERANCE, pb)
        before = {k: np.array(v, dtype=np.float32, copy=True) for k, v in out.items()}
        note, bits, bend, status = m.note_candidates(out, prm)
        for k in before:  # the caller's maps are left alone
            assert np.array_equal(before[k], np.asarray(out[k], dtype=np.float32), equal_nan=True), k
        if status:
            return status
        r_note, r_bits, r_bend = NO.note_candidates(out, args["onset_thresh"], args.get("infer_onsets", True),
                                                    args.get("min_freq"), args.get("max_freq"), pb)
        assert np.array_equal(note, r_note) and np.array_equal(bits, r_bits)
        assert (bend is None and r_bend is None) or np.array_equal(bend, r_bend)
        a = {k: v.copy() for k, v in before.items()}
        raw, bends, n = NC._decode(a["note"], a["onset"], a["contour"], args["onset_thresh"], args["frame_thresh"],
                                   args.get("min_note_len", 11), args.get("infer_onsets", True), args.get("max_freq"),
                                   args.get("min_freq"), args.get("melodia_trick", True), NC.ENERGY_TOLERANCE, pb)
        full = [(float(raw[i].start_s), float(raw[i].end_s), int(raw[i].pitch_midi), np.float32(raw[i].amplitude),
                 bends[raw[i].bend_offset : raw[i].bend_offset + raw[i].n_bends].tolist() if pb else None) for i in range(n)]
        got = NC.decode_candidates(note, bits, bend, prm)
        assert len(got) == len(full)
        for x, y in zip(got, full):
            assert x[:3] == y[:3] and np.float32(x[3]).tobytes() == np.float32(y[3]).tobytes() and x[4] == y[4]
        return 0

    n_events = 0
    for name in note_cases.CASES:
        out, args = note_cases.case_args(name)
        args = {k: v for k, v in args.items() if k not in ("multiple_pitch_bends", "midi_tempo")}
        assert check(out, args) == 0, name
    rng = np.random.default_rng(5)
    for trial in range(12):
        T = int(rng.integers(3, 3000))
        out = {"note": rng.random((T, 88), dtype=np.float32) ** 3, "onset": rng.random((T, 88), dtype=np.float32) ** 4,
               "contour": rng.random((T, 264), dtype=np.float32)}
        if trial % 3 == 0:  # note-like structure: runs along time
            out["note"] = np.repeat(out["note"][:: 7], 7, axis=0)[:T].copy()
        args = dict(onset_thresh=float(rng.choice([0.2, 0.5, 0.9])), frame_thresh=float(rng.choice([0.1, 0.3])),
                    infer_onsets=bool(rng.integers(0, 2)), melodia_trick=bool(trial % 2), min_note_len=int(rng.choice([3, 11])),
                    include_pitch_bends=bool(rng.integers(0, 2)), min_freq=float(rng.choice([0, 100.0])) or None,
                    max_freq=float(rng.choice([0, 2000.0])) or None)
        assert check(out, args) == 0, trial
    # device-resident maps (what the track path hands over), a silent map, and the two cases the host must take
    import torch

    out, args = note_cases.case_args("clip_default")
    dev = {k: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).cuda() for k, v in out.items()}
    prm = NC._note_params(0.5, 0.3, 11, True, None, None, True, NC.ENERGY_TOLERANCE, True)
    n1, b1, e1, s1 = m.note_candidates(dev, prm)
    n2, b2, e2, s2 = m.note_candidates(out, prm)
    assert s1 == s2 == 0 and np.array_equal(n1, n2) and np.array_equal(b1, b2) and np.array_equal(e1, e2)
    zero = {"note": np.zeros((50, 88), np.float32), "onset": np.zeros((50, 88), np.float32), "contour": np.zeros((50, 264), np.float32)}
    assert check(zero, dict(onset_thresh=0.5, frame_thresh=0.3)) == 0
    # a contour map with NaN / +-inf entries (np.argmax: the first NaN wins, then the first maximum): the bend kernel's
    # block-uniform slow path, at a frame count that is no multiple of its 16-frame blocks
    odd = {"note": rng.random((203, 88), dtype=np.float32) ** 3, "onset": rng.random((203, 88), dtype=np.float32) ** 4,
           "contour": rng.random((203, 264), dtype=np.float32)}
    odd["contour"][5, 100] = np.nan
    odd["contour"][5, 130] = np.nan
    odd["contour"][40, 3] = np.nan
    odd["contour"][41, 260] = np.inf
    odd["contour"][60, :] = -np.inf
    odd["contour"][61, 20:90] = np.inf
    odd["contour"][202, 263] = np.nan
    assert check(odd, dict(onset_thresh=0.5, frame_thresh=0.3, min_note_len=3)) == 0
    bad = {k: v.copy() for k, v in zero.items()}
    bad["onset"][7, 3] = np.nan
    assert check(bad, dict(onset_thresh=0.5, frame_thresh=0.3)) == 1
    assert check(zero, dict(onset_thresh=0.0, frame_thresh=0.3)) == 1
    m.close()


def test_whole_path_is_run_to_run_deterministic():
    """The same batch through the whole path 150 times at three batch sizes: every output bit-identical to the first pass.
    The rim kernel of contour conv1 orders its LDS-DMA by hand (counted s_waitcnt vmcnt, one barrier per item): a missing
    wait or a barrier one item early shows up as a run-to-run difference long before it shows up as a parity error."""
    import subprocess
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "experiments", "determinism_stress.py")
    res = subprocess.run([sys.executable, tool, "150"], capture_output=True, text=True, timeout=600, env=_ab_env())
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.gpu
def test_track_maps_gives_the_nan_fallback_its_maps_without_a_second_pass():
    """when bp_infer_pcm_raw_candidates reports status 1 (a NaN in the maps, or an onset threshold <= 0 as
    here) the caller needs the maps themselves; bp_track_maps hands over the three posteriorgrams that call left on the device — bit for bit what
    bp_infer_pcm_raw returns for the same samples — and refuses once the handle's track buffer has been used again."""
    import ctypes as C

    from basic_pitch_amd import Model, _native
    from basic_pitch_amd.note_creation import _note_params

    rng = np.random.default_rng(5)
    n = 3 * 22050
    pcm = (0.2 * np.sin(2 * np.pi * 330.0 * np.arange(n) / 22050.0) + 0.01 * rng.standard_normal(n)).astype(np.float32)
    pcm = pcm.reshape(-1, 1)
    m = Model(max_windows=16)
    want = m.predict_pcm_raw(pcm, _native.BP_PCM_F32, n, 1, 22050)
    T = want["note"].shape[0]
    prm = _note_params(0.0, 0.3, 11, True, None, None, True, 11, True)
    note = np.empty((T, 88), np.float32)
    bits = np.empty((T, 12), np.uint8)
    bend = np.empty((T, 88), np.int8)
    status = C.c_int(0)
    rc = m._lib.bp_infer_pcm_raw_candidates(m._handle, pcm.ctypes.data, _native.BP_PCM_F32, n, 1, 22050, C.byref(prm),
                                            note.ctypes.data, bits.ctypes.data, bend.ctypes.data, C.byref(status))
    assert rc == 0 and status.value == 1
    got = {k: np.empty_like(want[k]) for k in ("note", "onset", "contour")}
    assert m._lib.bp_track_maps(m._handle, T + 1, got["note"].ctypes.data, got["onset"].ctypes.data, got["contour"].ctypes.data,
                                _native.BP_MEM_HOST) == -1
    rc = m._lib.bp_track_maps(m._handle, T, got["note"].ctypes.data, got["onset"].ctypes.data, got["contour"].ctypes.data,
                              _native.BP_MEM_HOST)
    assert rc == 0
    for k in got:
        assert np.array_equal(got[k], want[k], equal_nan=True), k
    m.predict_pcm_raw(pcm, _native.BP_PCM_F32, n, 1, 22050)  # the track buffer is used by another call: nothing to hand over
    assert m._lib.bp_track_maps(m._handle, T, got["note"].ctypes.data, got["onset"].ctypes.data, got["contour"].ctypes.data,
                                _native.BP_MEM_HOST) == -1
    m.close()
