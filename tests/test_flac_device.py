"""FLAC decoded ON THE DEVICE (csrc/flac_device.hip) against the host decoder (csrc/flac_decode.cpp,
itself pinned to RFC 9639's worked example and to WAV <-> FLAC pairs in tests/test_audio_decode.py): the same integers,
sample for sample, on the RFC's example file, on streams from the test-side encoder that exercise every subframe type /
stereo mode / residual coding / sample size, and on the reference's second recording; the MD5 of STREAMINFO is checked
here on the device's output; corrupt and truncated streams are errors; `predict()` on a .flac file goes through it."""
import hashlib
import os

import numpy as np
import pytest

import flac_writer as FW
from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model():
    from basic_pitch_amd import Model

    m = Model(max_windows=32)
    yield m
    m.close()


def _host_ints(data, bits):
    """the host decoder's floats back as the integers they were made of"""
    from basic_pitch_amd import _native

    import ctypes as C

    lib = _native.load_library()
    ch, sr, b, n = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    assert lib.bp_flac_info(data, len(data), C.byref(ch), C.byref(sr), C.byref(b), C.byref(n)) == 0
    pcm = np.empty((n.value, ch.value), np.float32)
    got = C.c_int64()
    rc = lib.bp_flac_decode(data, len(data), pcm.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(got))
    assert rc == 0 and got.value == n.value
    return np.round(pcm.astype(np.float64) * (1 << (bits - 1))).astype(np.int64), sr.value


def _md5_of(pcm, bits):
    bps = (bits + 7) // 8
    raw = b"".join(int(v).to_bytes(bps, "little", signed=True) for v in pcm.reshape(-1))
    return hashlib.md5(raw).digest()


def test_rfc9639_example_on_the_device(model):
    data = bytes.fromhex(
        "664c6143" "80000022" "10001000" "00000f00000f" "0ac442f000000001" "3e84b41807dc690307586a3dad1a2e0f"
        "fff869180000bf" "0358fd03128b" "aa9a"
    )
    pcm, sr = model.flac_decode_device(data)
    assert sr == 44100 and pcm.tolist() == [[25588, 10416]]
    assert _md5_of(pcm, 16) == data[26:42]
    for pos in (0x2C, 0x30, 0x33, 0x38):  # a flipped bit in the frame: CRC-8 / CRC-16 / chain
        bad = bytearray(data)
        bad[pos] ^= 0x04
        with pytest.raises(ValueError):
            model.flac_decode_device(bytes(bad))


@pytest.mark.parametrize(
    "bits,ch,sr,n,bs",
    [(16, 1, 44100, 5000, 1152), (16, 2, 44100, 9000, 1152), (24, 2, 48000, 4000, 576), (8, 1, 8000, 3000, 192),
     (16, 3, 22050, 2500, 1000), (16, 2, 12345, 700, 300), (20, 2, 96000, 2000, 4096), (12, 1, 16000, 1200, 256)],
)
def test_device_decoder_equals_host_decoder_on_the_round_trip_matrix(model, bits, ch, sr, n, bs):
    """tests/test_audio_decode.py::test_flac_round_trip's streams (the test-side encoder cycles through constant / verbatim /
    fixed 0..4 / LPC subframes, Rice and Rice2 with several partition orders, escaped partitions, all four stereo modes,
    wasted bits, an ID3 tag in front): device == source PCM == host decoder, and the MD5 of STREAMINFO holds."""
    rng = np.random.default_rng(bits * 100 + ch)
    t = np.arange(n) / sr
    x = np.stack([0.4 * np.sin(2 * np.pi * 220 * (c + 1) * t) + 0.05 * rng.standard_normal(n) for c in range(ch)], 1)
    full = 1 << (bits - 1)
    pcm = np.clip(np.round(x * full), -full, full - 1).astype(np.int64)
    pcm[100:400] = 0
    pcm[1200:1500] = (pcm[1200:1500] >> 3) << 3
    pcm[50] = -full
    data = FW.encode(pcm, sr, bits, blocksize=bs, id3=(ch == 3))
    got, sr2 = model.flac_decode_device(data)
    assert sr2 == sr and got.shape == (n, ch)
    assert np.array_equal(got, pcm)
    host, _ = _host_ints(data, bits)
    assert np.array_equal(got, host)
    lay = model.flac_layout(data)
    md5_at = data.index(b"fLaC") + 8 + 18
    assert _md5_of(got, bits) == data[md5_at : md5_at + 16] and lay["n_frames"] == n
    # truncation and corruption are errors, not silence
    with pytest.raises(ValueError):
        model.flac_decode_device(data[: len(data) - 7])
    bad = bytearray(data)
    bad[len(data) // 2] ^= 0x10
    with pytest.raises(ValueError):
        model.flac_decode_device(bytes(bad))


def test_second_recording_and_whole_path(model, tmp_path):
    """tests/golden/vocadito_14.flac (537,924 frames, 132 frames of 4096): device == host decoder on every sample, MD5 of
    STREAMINFO on the device's output; its posteriorgrams through bp_infer_flac equal those of the host-decoded samples
    through bp_infer_pcm bit for bit (the same ingest kernels run on the same integers); `predict()` takes the device path
    for a .flac file and returns the same events as for the host-decoded samples."""
    from basic_pitch_amd import audio, inference as inf

    path = os.path.join(GOLDEN, "vocadito_14.flac")
    data = open(path, "rb").read()
    got, sr = model.flac_decode_device(data)
    host, sr2 = _host_ints(data, 16)
    assert sr == sr2 == 44100 and got.shape == host.shape == (537924, 1) and np.array_equal(got, host)
    md5_at = data.index(b"fLaC") + 8 + 18
    assert _md5_of(got, 16) == data[md5_at : md5_at + 16]
    a = model.predict_flac(data)
    pcm, _ = audio.read_audio(path)
    b = model.predict_pcm(pcm, sr)
    for k in ("note", "onset", "contour"):
        assert a[k].shape == b[k].shape and np.array_equal(a[k], b[k]), k
    calls = []
    orig = model.predict_flac
    model.predict_flac = lambda blob: (calls.append(len(blob)), orig(blob))[1]
    try:
        mo, _, ev = inf.predict(path, model)
    finally:
        del model.predict_flac
    assert calls == [len(data)] and all(np.array_equal(mo[k], a[k]) for k in a) and len(ev) > 20
    # a stream without a sample count in STREAMINFO is left to the host decoder — predict() still works
    rng = np.random.default_rng(5)
    pcm16 = (3000 * np.sin(np.arange(30000) * 0.05) + rng.integers(-200, 200, 30000)).astype(np.int64)[:, None]
    nohdr = FW.encode(pcm16, 22050, 16, blocksize=1024, total_in_header=False)
    p = tmp_path / "nototal.flac"
    p.write_bytes(nohdr)
    from basic_pitch_amd._native import NativeLibraryError

    with pytest.raises((ValueError, NativeLibraryError)):
        model.flac_decode_device(nohdr)
    mo2 = inf.run_inference(str(p), model)
    assert mo2["note"].shape[0] == int(30000 / 36164 * 142)


def test_long_random_streams_with_every_coding_choice(model):
    """Fuzz: noise-like and tonal 16-bit stereo material, 20 frames each, block sizes 192..4608, the encoder's coding
    choices cycling per frame — device == host on every sample."""
    for seed, bs in ((1, 192), (2, 1024), (3, 4608), (4, 2304)):
        rng = np.random.default_rng(seed)
        n = bs * 20 + 77
        t = np.arange(n)
        x = np.stack([6000 * np.sin(t * 0.01 * (seed + 1)) + rng.integers(-3000, 3000, n),
                      5000 * np.sin(t * 0.013) + rng.integers(-30, 30, n)], 1).astype(np.int64)
        data = FW.encode(x, 44100, 16, blocksize=bs)
        got, _ = model.flac_decode_device(data)
        assert np.array_equal(got, x), (seed, bs)


def test_bench_corpus_encoder(model, tmp_path):
    """tools/flac_synth.c (the encoder `bench.py --workload files --native --flac` makes its corpus with: LPC order 8, Rice
    partitions, block size 4096, a short last block) writes streams both decoders accept: host (every CRC checked) == device
    == the WAV it was given; and the native pipeline transcribes the .flac through the device decoder to the bytes it
    writes for the .wav."""
    import subprocess
    import wave

    from basic_pitch_amd import audio, transcribe_files

    exe = str(tmp_path / "flac_synth")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["gcc", "-O2", "-o", exe, os.path.join(root, "tools", "flac_synth.c"), "-lm"], check=True)
    rng = np.random.default_rng(9)
    n = 44100 * 12 + 333
    t = np.arange(n) / 44100.0
    x = 0.3 * np.sin(2 * np.pi * 330.0 * t) * (np.sin(2 * np.pi * 1.1 * t) > 0) + 0.01 * rng.standard_normal(n)
    pcm = (np.clip(np.stack([x, x[::-1]], 1), -1, 1) * 32767).astype("<i2")
    wav, flac = tmp_path / "tone.wav", tmp_path / "tone.flac"
    with wave.open(str(wav), "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(44100)
        w.writeframes(pcm.tobytes())
    subprocess.run([exe, str(wav), str(flac)], check=True, stderr=subprocess.DEVNULL)
    data = flac.read_bytes()
    assert len(data) < 0.9 * pcm.nbytes
    y, sr = audio.read_audio(flac)
    assert sr == 44100 and np.array_equal(np.round(y * 32768).astype(np.int64), pcm.astype(np.int64))
    got, _ = model.flac_decode_device(data)
    assert np.array_equal(got, pcm.astype(np.int32))
    outs = {}
    before = model._lib.bp_files_direct_reads()
    for name, src, kw in (("wav", wav, {}), ("flac_dev", flac, {}), ("flac_host", flac, {"host_flac": True})):
        o = tmp_path / name
        o.mkdir()
        (rep,) = transcribe_files([str(src)], o, models=[model], threads=1, **kw)
        assert rep["status"] == 0 and rep["n_note_events"] > 5, rep
        outs[name] = {f: open(os.path.join(o, f), "rb").read() for f in sorted(os.listdir(o))}
    assert model._lib.bp_files_direct_reads() == before
    names = [sorted(v) for v in outs.values()]
    assert names[0] == names[1] == names[2] == ["tone_basic_pitch.csv", "tone_basic_pitch.mid"]
    for f in names[0]:
        assert outs["wav"][f] == outs["flac_dev"][f] == outs["flac_host"][f], f


def test_every_lpc_order_with_the_classes_mixed_in_one_wave(model):
    """LPC orders 1..32 cycling from frame to frame (the lanes of one wave are frames): the register predictors of 4 / 8 / 12
    products — the widest class present in the wave serves all its lanes —, the generic path of the orders above 12, partition
    orders 0..2, Rice and Rice2, escaped partitions; 16- and 24-bit.  device == source == host decoder."""
    orders = [1, 2, 4, 5, 8, 9, 12, 13, 20, 32, 3, 11, 7, 16, 6, 10]
    for bits, bs, seed in ((16, 256, 11), (24, 192, 12), (16, 1000, 13)):
        rng = np.random.default_rng(seed)
        n = bs * 70 + 31
        t = np.arange(n)
        full = 1 << (bits - 1)
        x = np.stack([0.3 * full * np.sin(t * 0.021) + rng.integers(-full // 64, full // 64, n),
                      0.2 * full * np.sin(t * 0.0057) + rng.integers(-full // 4096 - 2, full // 4096 + 2, n)], 1).astype(np.int64)
        plan = lambda fi: dict(kind="lpc", lpc_order=orders[fi % len(orders)], porder=fi % 3, rice2=bool(fi % 2),
                               escape=(fi % 7 == 6), stereo=("indep", "ms", "ls", "sr")[fi % 4])
        data = FW.encode(x, 48000, bits, blocksize=bs, plan=plan)
        got, _ = model.flac_decode_device(data)
        assert np.array_equal(got, x), (bits, bs)
        host, _ = _host_ints(data, bits)
        assert np.array_equal(got, host), (bits, bs)


def test_blocking_wait_lane_sleeps_while_the_device_decodes():
    """BP_FLAG_BLOCKING_WAIT (the lanes of a file job): the end of a call sleeps between event queries — a blocking-sync event
    wait polls on this runtime, user time == wall time (tools/experiments/host_cpu.py) — so a lane that waits for 20 decodes
    of a 20-second stream costs its core a fraction of the wall time, and decodes the same samples."""
    import ctypes as C
    import resource
    import time

    from basic_pitch_amd import Model

    rng = np.random.default_rng(21)
    n = 20 * 44100
    t = np.arange(n)
    x = np.stack([8000 * np.sin(t * 0.02) + rng.integers(-500, 500, n), 6000 * np.sin(t * 0.013) + rng.integers(-300, 300, n)],
                 1).astype(np.int64)
    data = FW.encode(x, 44100, 16, blocksize=4096, plan=lambda fi: dict(kind="lpc", porder=3, rice2=False,
                                                                                          escape=False, stereo="indep"))
    spin, sleep = Model(max_windows=16), Model(max_windows=16, blocking_wait=True)
    a, _ = spin.flac_decode_device(data)
    b, _ = sleep.flac_decode_device(data)
    assert np.array_equal(a, b) and np.array_equal(a, x)
    nf = C.c_int64()

    def cost(m):
        for _ in range(3):
            m._lib.bp_flac_decode_device(m._handle, data, len(data), None, 0, C.byref(nf))
        r0, t0 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
        for _ in range(20):
            assert m._lib.bp_flac_decode_device(m._handle, data, len(data), None, 0, C.byref(nf)) == 0
        r1, t1 = resource.getrusage(resource.RUSAGE_SELF), time.perf_counter()
        return (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime), t1 - t0

    cpu_spin, wall_spin = cost(spin)
    cpu_sleep, wall_sleep = cost(sleep)
    assert cpu_sleep < 0.6 * wall_sleep, (cpu_sleep, wall_sleep)
    assert wall_sleep < 2.0 * wall_spin + 0.01, (wall_sleep, wall_spin)
    spin.close()
    sleep.close()


def test_corrupt_small_files_are_errors_or_the_host_decoders_samples(model):
    """A lane that loses a corrupt stream runs ahead of its frame (long unary runs: kilobytes per burst) — it must stay inside
    the file's buffer (its block loads are clamped to the padding behind the file) and the call must end in an error or, if
    the damage happens to be decodable, in exactly what the host decoder makes of the same bytes.  Small files: the buffer's
    slack behind them is smallest."""
    rng = np.random.default_rng(99)
    n = 700
    x = np.stack([3000 * np.sin(np.arange(n) * 0.05) + rng.integers(-200, 200, n)], 1).astype(np.int64)
    good = FW.encode(x, 16000, 16, blocksize=256, plan=lambda fi: dict(kind="lpc", porder=1, rice2=False, escape=False))
    start = good.index(b"fLaC") + 42
    outcomes = {"error": 0, "decoded": 0}
    for trial in range(150):
        bad = bytearray(good)
        for _ in range(1 + trial % 4):
            pos = int(rng.integers(start, len(bad)))
            bad[pos] = 0 if trial % 3 == 0 else int(rng.integers(0, 256))  # zeros make long unary runs
        if trial % 5 == 0:
            del bad[int(rng.integers(start + 10, len(bad))):]
        try:
            got, _ = model.flac_decode_device(bytes(bad))
        except ValueError:
            outcomes["error"] += 1
            continue
        outcomes["decoded"] += 1
        host, _ = _host_ints(bytes(bad), 16)
        assert np.array_equal(got, host), trial
    assert outcomes["error"] > 100, outcomes
