"""Decode side of `librosa.load` (inference.py:239): the native FLAC decoder (csrc/flac_decode.cpp, host C++) and the
read_audio dispatch.  CPU only.

Pins: (1) the worked example of the FLAC specification itself (RFC 9639, appendix D.1: a complete one-frame file whose
decoded samples the RFC states) — a known answer from outside this repo, with live CRC-8 / CRC-16 / MD5 checks;
(2) WAV <-> FLAC pairs: files coded by the test-side encoder (tests/flac_writer.py, every subframe type / stereo mode /
residual coding) decode to exactly the PCM they were made from, including an excerpt of the reference's test clip."""
import os

import numpy as np
import pytest

import flac_writer as FW
from conftest import GOLDEN


def test_rfc9639_example_file(tmp_path):
    """RFC 9639 D.1 "Decoding Example 1": 16-bit stereo 44.1 kHz, one frame of one sample, verbatim subframes with
    wasted bits; the RFC gives the decoded samples as 25588 (left) and 10416 (right)."""
    from basic_pitch_amd import audio

    data = bytes.fromhex(
        "664c6143" "80000022" "10001000" "00000f00000f" "0ac442f000000001" "3e84b41807dc690307586a3dad1a2e0f"
        "fff869180000bf" "0358fd03128b" "aa9a"
    )
    p = tmp_path / "rfc.flac"
    p.write_bytes(data)
    x, sr = audio.read_audio(p)
    assert sr == 44100 and x.shape == (1, 2) and x.dtype == np.float32
    assert (x[0] * 32768).tolist() == [25588.0, 10416.0]
    # every integrity check is live: flip one bit anywhere in the frame or the MD5 and the file is rejected
    for pos in (0x1A, 0x2C, 0x30, 0x33, 0x38):
        bad = bytearray(data)
        bad[pos] ^= 0x04
        p.write_bytes(bytes(bad))
        with pytest.raises(ValueError):
            audio.read_audio(p)


@pytest.mark.parametrize(
    "bits,ch,sr,n,bs",
    [(16, 1, 44100, 5000, 1152), (16, 2, 44100, 9000, 1152), (24, 2, 48000, 4000, 576), (8, 1, 8000, 3000, 192),
     (16, 3, 22050, 2500, 1000), (16, 2, 12345, 700, 300), (20, 2, 96000, 2000, 4096), (12, 1, 16000, 1200, 256)],
)
def test_flac_round_trip(tmp_path, bits, ch, sr, n, bs):
    from basic_pitch_amd import audio

    rng = np.random.default_rng(bits * 100 + ch)
    t = np.arange(n) / sr
    x = np.stack([0.4 * np.sin(2 * np.pi * 220 * (c + 1) * t) + 0.05 * rng.standard_normal(n) for c in range(ch)], 1)
    full = 1 << (bits - 1)
    pcm = np.clip(np.round(x * full), -full, full - 1).astype(np.int64)
    pcm[100:400] = 0                                   # constant subframes
    pcm[1200:1500] = (pcm[1200:1500] >> 3) << 3        # wasted bits
    pcm[50] = -full                                    # extreme value
    data = FW.encode(pcm, sr, bits, blocksize=bs, id3=(ch == 3), total_in_header=(bits != 8))
    p = tmp_path / "t.flac"
    p.write_bytes(data)
    y, sr2 = audio.read_audio(p)
    assert sr2 == sr and y.shape == (n, ch)
    assert np.array_equal(y, (pcm / float(full)).astype(np.float32))
    assert abs(audio.get_duration(p) - n / sr) < 1e-12
    # truncation and corruption are errors, not silence
    p.write_bytes(data[: len(data) - 7])
    with pytest.raises(ValueError):
        audio.read_audio(p)
    bad = bytearray(data)
    bad[len(data) // 2] ^= 0x10
    p.write_bytes(bytes(bad))
    with pytest.raises(ValueError):
        audio.read_audio(p)


def test_wav_flac_pair_of_the_reference_clip():
    """tests/golden/vocadito_10_excerpt.flac (tools/make_flac_fixture.py) == seconds 2-3 of vocadito_10.wav."""
    from basic_pitch_amd import audio

    wav, sr = audio.read_wav(os.path.join(GOLDEN, "vocadito_10.wav"))
    fl, sr2 = audio.read_audio(os.path.join(GOLDEN, "vocadito_10_excerpt.flac"))
    assert sr2 == sr == 44100 and fl.shape == (44100, 1)
    assert np.array_equal(fl, wav[2 * sr : 3 * sr])
    a, _ = audio.load(os.path.join(GOLDEN, "vocadito_10_excerpt.flac"))
    assert a.shape == (22050,) and np.array_equal(a, audio.resample(wav[2 * sr : 3 * sr, 0], sr))


def test_unknown_container_is_a_clear_error(tmp_path):
    from basic_pitch_amd import audio

    p = tmp_path / "x.mp3"
    p.write_bytes(b"\xff\xfb\x90\x00" + b"\0" * 400)
    with pytest.raises((ValueError, RuntimeError)) as e:
        audio.read_audio(p)
    assert "mp3" in str(e.value) or "decoder" in str(e.value) or "Error" in str(e.value)


def test_second_clip_flac_decodes_and_has_the_expected_length():
    """tests/golden/vocadito_14.flac (the reference's second recording, LPC / fixed subframes from the test-side
    encoder): MD5-verified decode, 537,924 frames at 44.1 kHz -> ceil(n / 2) samples at 22.05 kHz."""
    from basic_pitch_amd import audio

    x, sr = audio.read_audio(os.path.join(GOLDEN, "vocadito_14.flac"))
    assert sr == 44100 and x.shape == (537924, 1) and np.abs(x).max() <= 1.0
    g = np.load(os.path.join(GOLDEN, "vocadito_14_expected.npz"))
    y, _ = audio.load(os.path.join(GOLDEN, "vocadito_14.flac"))
    assert y.shape == (int(g["n_samples_22k"][0]),) == (268962,)


def test_flac_every_truncation_point_near_the_end_is_an_error(tmp_path):
    """The decoder reads 64-bit windows while eight bytes are left and byte-wise after that: cutting the stream anywhere
    in its last 48 bytes is reported (never read past the end, never silence), the whole stream still decodes."""
    from basic_pitch_amd import audio

    rng = np.random.default_rng(12)
    pcm = rng.integers(-3000, 3000, size=(4096 + 37, 2)).astype(np.int64)
    pcm[100:300, 0] = (2000 * np.sin(np.arange(200) * 0.2)).astype(np.int64)
    data = FW.encode(pcm, 44100, 16, blocksize=1024)
    p = tmp_path / "t.flac"
    p.write_bytes(data)
    y, _ = audio.read_audio(p)
    assert np.array_equal(y, (pcm / 32768.0).astype(np.float32))
    for cut in range(1, 49):
        p.write_bytes(data[: len(data) - cut])
        with pytest.raises(ValueError):
            audio.read_audio(p)


def test_flac_rice_code_that_fills_the_bit_register_exactly(tmp_path):
    """A 64-bit Rice code (k = 0, residual -32: 63 zeros and the stop bit) at a byte-aligned position: the register-resident
    Rice reader then holds exactly the code (need == cnt == 64) and must not shift by the register's width (round-4
    advisor finding: `buf <<= 64` is undefined, on x86 a no-op, and every later residual of the partition was wrong)."""
    from basic_pitch_amd import audio

    n = 192
    r = np.zeros(n, np.int64)
    # after the 64-bit code sixty 1-bit codes, then 00001 across the next register's last bit: the stale stop bit that the
    # no-op shift leaves in the register would be OR-ed onto that 0
    r[0], r[61], r[100], r[150] = -32, 2, -32, 1
    pcm = (r * 64)[:, None]  # six wasted bits: 8 + 6 header bits + 10 residual-header bits put the first code on a byte
    data = FW.encode(pcm, 22050, 16, blocksize=n, plan=lambda fi: dict(kind="fixed0", porder=0, rice2=False, escape=False))
    p = tmp_path / "r64.flac"
    p.write_bytes(data)
    y, sr = audio.read_audio(p)
    assert sr == 22050 and np.array_equal(y[:, 0], (pcm[:, 0] / 32768.0).astype(np.float32))
