"""The native file pipeline's host pieces (csrc/file_pipeline.cpp) against the Python code they replace — no GPU:

* bp_notes_to_midi == `note_events_to_midi(...).write()` (basic_pitch_amd/midi.py, itself pinned byte for byte to
  pretty_midi + mido's layout by tests/golden/midi) on the 16 reference-generated note cases, and therefore == the
  committed golden .mid files;
* bp_notes_to_csv == `save_note_events` (inference.py:409-428: csv.writer rows, repr() floats, "\\r\\n");
* bp_wav_decode == `audio.read_wav` for every sample format the Python reader takes.
The whole pipeline (bp_transcribe_files) is a gpu test at the bottom.
"""
import ctypes as C
import os
import struct
import wave

import numpy as np
import pytest

import note_cases
from basic_pitch_amd import _native
from basic_pitch_amd import note_creation as NC
from basic_pitch_amd import inference as INF

GOLDEN = note_cases.GOLDEN


def _decode(out, args):
    a = dict(args)
    multi, tempo = a.pop("multiple_pitch_bends", False), a.pop("midi_tempo", 120)
    ev, bends, n = NC._decode(out["note"], out["onset"], out["contour"], a["onset_thresh"], a["frame_thresh"], a["min_note_len"],
                              a.get("infer_onsets", True), a.get("max_freq"), a.get("min_freq"), a.get("melodia_trick", True),
                              NC.ENERGY_TOLERANCE, a.get("include_pitch_bends", True))
    return ev, bends, n, multi, tempo, a.get("include_pitch_bends", True)


def _native_bytes(fn, *args):
    need = fn(*args, None, 0)
    assert need >= 0, need
    buf = (C.c_uint8 * max(1, need))()
    assert fn(*args, C.addressof(buf), need) == need
    return bytes(buf[:need])


@pytest.mark.parametrize("name", sorted(note_cases.CASES))
def test_native_midi_and_csv_equal_the_python_writers(tmp_path, name):
    lib = _native.load_library()
    out, args = note_cases.case_args(name)
    ev, bends, n, multi, tempo, with_bends = _decode({k: v.copy() for k, v in out.items()}, args)
    bp = bends.ctypes.data if with_bends else None
    got_mid = _native_bytes(lambda *a: lib.bp_notes_to_midi(C.addressof(ev), n, bp, int(multi), float(tempo), *a))
    got_csv = _native_bytes(lambda *a: lib.bp_notes_to_csv(C.addressof(ev), n, bp, *a))

    midi, events = NC.model_output_to_notes({k: v.copy() for k, v in out.items()}, **args)
    assert got_mid == midi.to_bytes()
    golden = os.path.join(GOLDEN, "midi", f"{name}.mid")
    if os.path.exists(golden):
        assert got_mid == open(golden, "rb").read()
    path = tmp_path / "n.csv"
    INF.save_note_events(events, path)
    assert got_csv == open(path, "rb").read()


def test_native_midi_of_no_events_is_the_timing_track_only():
    lib = _native.load_library()
    got = _native_bytes(lambda *a: lib.bp_notes_to_midi(None, 0, None, 0, 120.0, *a))
    midi, _ = NC.model_output_to_notes({"note": np.zeros((10, 88), np.float32), "onset": np.zeros((10, 88), np.float32),
                                        "contour": np.zeros((10, 264), np.float32)}, 0.5, 0.3)
    assert got == midi.to_bytes()


def test_python_float_repr_is_reproduced():
    """The CSV's times are repr(float): shortest round-trip digits, fixed notation for 1e-4 <= |x| < 1e16."""
    lib = _native.load_library()
    rng = np.random.default_rng(0)
    vals = [0.0, 1.0, 2.5, 0.1, 1e-4, 9.999e-5, 1e-5, 123456789.125, 1e15, 1e16, 1.5e16, 1e22, 3.0e-7, 0.3715192743764172,
            178.12345678901234, 5e-324, 1.7976931348623157e308] + list(rng.uniform(0, 200, 200)) + list(10.0 ** rng.uniform(-8, 18, 100))
    ev = (_native.bp_note_event * len(vals))()
    for i, v in enumerate(vals):
        ev[i].start_s, ev[i].end_s, ev[i].pitch_midi, ev[i].amplitude, ev[i].n_bends = v, -v, 60, 0.5, 0
    text = _native_bytes(lambda *a: lib.bp_notes_to_csv(C.addressof(ev), len(vals), None, *a)).decode()
    rows = text.split("\r\n")[1:-1]
    assert len(rows) == len(vals)
    for v, row in zip(vals, rows):
        s, e, pitch, vel = row.split(",")
        assert s == repr(float(v)) and e == repr(-float(v)), (v, row)
        assert pitch == "60" and vel == str(int(np.round(127 * np.float32(0.5))))


def _wav_bytes(fmt_tag, bits, channels, rate, payload, extensible=False):
    if extensible:
        fmt = struct.pack("<HHIIHHHHIH", 0xFFFE, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits, 22, bits, 3,
                          fmt_tag) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 5) + b"junk!\x00" + \
        b"data" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
    return b"RIFF" + struct.pack("<I", len(body)) + body


@pytest.mark.parametrize("kind", ["u8", "i16", "i24", "i32", "f32", "f64", "i16_ext", "i16_odd_tail"])
def test_native_wav_reader_equals_read_wav(tmp_path, kind):
    from basic_pitch_amd import audio

    lib = _native.load_library()
    rng = np.random.default_rng(3)
    n, ch = 1001, 2 if kind != "u8" else 1
    if kind == "u8":
        data = _wav_bytes(1, 8, ch, 8000, rng.integers(0, 256, n * ch, dtype=np.uint8).tobytes())
    elif kind in ("i16", "i16_ext", "i16_odd_tail"):
        raw = rng.integers(-32768, 32768, n * ch, dtype=np.int16).astype("<i2").tobytes()
        data = _wav_bytes(1, 16, ch, 44100, raw + (b"\x7f" if kind == "i16_odd_tail" else b""), extensible=kind == "i16_ext")
    elif kind == "i24":
        v = rng.integers(-(1 << 23), 1 << 23, n * ch)
        data = _wav_bytes(1, 24, ch, 48000, b"".join(int(x & 0xFFFFFF).to_bytes(3, "little") for x in v))
    elif kind == "i32":
        data = _wav_bytes(1, 32, ch, 96000, rng.integers(-(1 << 31), 1 << 31, n * ch, dtype=np.int64).astype("<i4").tobytes())
    elif kind == "f32":
        data = _wav_bytes(3, 32, ch, 22050, rng.uniform(-1, 1, n * ch).astype("<f4").tobytes())
    else:
        data = _wav_bytes(3, 64, ch, 22050, rng.uniform(-1, 1, n * ch).astype("<f8").tobytes())
    p = tmp_path / "x.wav"
    p.write_bytes(data)
    want, sr = audio.read_wav(p)
    c, r, b, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int64()
    assert lib.bp_wav_info(data, len(data), C.byref(c), C.byref(r), C.byref(b), C.byref(nf)) == 0
    assert (c.value, r.value, nf.value) == (want.shape[1], sr, want.shape[0])
    got = np.empty((nf.value, c.value), np.float32)
    k = C.c_int64()
    assert lib.bp_wav_decode(data, len(data), got.ctypes.data_as(C.POINTER(C.c_float)), nf.value, C.byref(k)) == 0
    assert k.value == nf.value and np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_native_wav_reader_rejects_what_read_wav_rejects():
    lib = _native.load_library()
    nf = C.c_int64()
    for data in (b"RIFFxxxxWAVE", b"not a wav file at all", _wav_bytes(2, 4, 1, 8000, b"\x00" * 64), _wav_bytes(1, 12, 1, 8000, b"\x00" * 64)):
        assert lib.bp_wav_info(data, len(data), None, None, None, C.byref(nf)) == -7  # BP_ERR_BAD_AUDIO
        assert lib.bp_files_last_error()


@pytest.mark.gpu
def test_transcribe_files_equals_predict_and_save(tmp_path):
    """bp_transcribe_files through basic_pitch_amd.transcribe_files: the same .mid and .csv bytes as predict_and_save
    (the Python pipeline over the same kernels) for WAV (stereo 44.1 kHz, mono 22.05 kHz) and FLAC input; a duplicate stem,
    an existing output and an unreadable file are reported per file and stop nothing."""
    import shutil

    from basic_pitch_amd import Model, predict_and_save, transcribe_files

    src = tmp_path / "in"
    src.mkdir()
    shutil.copy(os.path.join(GOLDEN, "vocadito_10.wav"), src / "clip.wav")
    rng = np.random.default_rng(5)
    t = np.arange(3 * 22050) / 22050.0
    x = (0.4 * np.sin(2 * np.pi * 220.0 * t) * (np.sin(2 * np.pi * 2.0 * t) > 0) + 0.01 * rng.standard_normal(t.size))
    with wave.open(str(src / "mono22k.wav"), "wb") as w:
        w.setnchannels(1), w.setsampwidth(2), w.setframerate(22050)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())
    flacs = [f for f in sorted(os.listdir(GOLDEN)) if f.endswith(".flac")]
    assert flacs, "the FLAC fixtures live next to the golden WAV"
    shutil.copy(os.path.join(GOLDEN, flacs[-1]), src / "f.flac")  # vocadito_14.flac: 12 s mono 44.1 kHz
    (src / "broken.wav").write_bytes(b"RIFF\x00\x00\x00\x00WAVEjunk")
    sub = src / "sub"
    sub.mkdir()
    shutil.copy(src / "mono22k.wav", sub / "clip.wav")  # same stem as the first input
    good = [src / "clip.wav", src / "mono22k.wav"] + ([src / "f.flac"] if flacs else [])
    paths = good + [src / "broken.wav", sub / "clip.wav"]

    ref_dir, out_dir = tmp_path / "ref", tmp_path / "out"
    ref_dir.mkdir(), out_dir.mkdir()
    model = Model(max_windows=64)
    predict_and_save(good, ref_dir, True, False, False, True, model)
    rep = transcribe_files(paths, out_dir, models=[model, Model(max_windows=64)], threads=3)
    assert [r["status"] for r in rep[: len(good)]] == [0] * len(good), rep
    assert rep[len(good)]["status"] != 0 and rep[len(good) + 1]["status"] != 0 and "same file stem" in rep[-1]["message"]
    for p in good:
        stem = os.path.splitext(os.path.basename(str(p)))[0]
        for ext in ("mid", "csv"):
            a, b = ref_dir / f"{stem}_basic_pitch.{ext}", out_dir / f"{stem}_basic_pitch.{ext}"
            assert a.read_bytes() == b.read_bytes(), (stem, ext)
    assert rep[0]["n_note_events"] == 28  # the reference's golden clip
    again = transcribe_files(good[:1], out_dir, models=[model])
    assert again[0]["status"] != 0 and "already exists" in again[0]["message"]
    # the workers' page-locked buffers are pooled between calls; releasing the pool changes nothing but memory
    assert all(set(r["ms"]) == {"read", "lane_wait", "device", "notes", "write"} for r in rep)
    # lanes built per device (here: the one GPU named twice = four lanes; on a node: one call over all its GPUs)
    out3 = tmp_path / "out3"
    out3.mkdir()
    rep3 = transcribe_files(good, out3, lanes=2, devices=[0, 0])
    assert [r["status"] for r in rep3] == [0] * len(good)
    assert (out3 / "clip_basic_pitch.csv").read_bytes() == (out_dir / "clip_basic_pitch.csv").read_bytes()
    _native.load_library().bp_files_release_buffers()
    out2 = tmp_path / "out2"
    out2.mkdir()
    rep2 = transcribe_files(good, out2, models=[model], threads=2, host_decode=True)  # the round-4 path: all three maps back
    assert [r["status"] for r in rep2] == [0] * len(good)
    for p in good:
        stem = os.path.splitext(os.path.basename(str(p)))[0]
        assert (out_dir / f"{stem}_basic_pitch.mid").read_bytes() == (out2 / f"{stem}_basic_pitch.mid").read_bytes()


@pytest.mark.gpu
def test_sharded_batch_job_through_the_native_pipeline(tmp_path):
    """predict_and_save_sharded(native=True): two worker processes on the one GPU, each running its LPT shard through
    bp_transcribe_files; same bytes as the single-process Python job, reports in input order."""
    import shutil

    from basic_pitch_amd import Model, predict_and_save, predict_and_save_sharded

    src = tmp_path / "in"
    src.mkdir()
    paths = []
    for i in range(4):
        p = src / f"c{i}.wav"
        shutil.copy(os.path.join(GOLDEN, "vocadito_10.wav"), p)
        paths.append(str(p))
    ref_dir, out_dir = tmp_path / "ref", tmp_path / "out"
    ref_dir.mkdir(), out_dir.mkdir()
    m = Model(max_windows=64)
    predict_and_save(paths[:1], ref_dir, True, False, False, True, m)
    m.close()
    rep = predict_and_save_sharded(paths, out_dir, True, False, False, True, gpus=1, workers_per_gpu=2, native=True,
                                   native_threads=2, native_lanes=1)
    assert [r["n_note_events"] for r in rep] == [28] * 4
    for i in range(4):
        for ext in ("mid", "csv"):
            assert (out_dir / f"c{i}_basic_pitch.{ext}").read_bytes() == (ref_dir / f"c0_basic_pitch.{ext}").read_bytes()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["u8", "i16", "i24", "i32", "f32", "f64"])
def test_raw_pcm_formats_equal_the_float_path(kind):
    """bp_infer_pcm_raw converts the file's sample format on the device: bit-identical posteriorgrams to converting on the
    host (bp_wav_decode's scaling = read_wav's) and calling bp_infer_pcm — stereo 44.1 kHz and mono 22.05 kHz, input and
    outputs in page-locked memory from bp_host_alloc, on a handle that waits asleep (BP_FLAG_BLOCKING_WAIT)."""
    from basic_pitch_amd import Model

    lib = _native.load_library()
    fmt, dt, tag, bits = {"u8": (_native.BP_PCM_U8, np.uint8, 1, 8), "i16": (_native.BP_PCM_S16, "<i2", 1, 16),
                          "i24": (_native.BP_PCM_S24, None, 1, 24), "i32": (_native.BP_PCM_S32, "<i4", 1, 32),
                          "f32": (_native.BP_PCM_F32, "<f4", 3, 32), "f64": (_native.BP_PCM_F64, "<f8", 3, 64)}[kind]
    rng = np.random.default_rng(11)
    model = Model(max_windows=8, blocking_wait=True)
    fp = C.POINTER(C.c_float)
    for ch, rate, n in ((2, 44100, 3 * 44100 + 17), (1, 22050, 2 * 22050 + 5)):
        t = np.arange(n) / rate
        x = 0.5 * np.sin(2 * np.pi * 330.0 * t)[:, None] * np.array([1.0, 0.6])[None, :ch] + 0.02 * rng.standard_normal((n, ch))
        x = np.clip(x, -0.99, 0.99)
        if kind == "u8":
            raw = np.round(x * 127 + 128).astype(np.uint8).tobytes()
        elif kind == "i24":
            v = np.round(x * (2**23 - 1)).astype(np.int64).ravel()
            raw = b"".join(int(s & 0xFFFFFF).to_bytes(3, "little") for s in v)
        elif tag == 1:
            raw = np.round(x * (2 ** (bits - 1) - 1)).astype(dt).tobytes()
        else:
            raw = x.astype(dt).tobytes()
        wav = _wav_bytes(tag, bits, ch, rate, raw)
        pcm = np.empty((n, ch), np.float32)
        k = C.c_int64()
        assert lib.bp_wav_decode(wav, len(wav), pcm.ctypes.data_as(fp), n, C.byref(k)) == 0 and k.value == n
        T = lib.bp_handle_track_n_frames(model._handle, lib.bp_handle_resampled_length(model._handle, n, rate))
        assert T > 0
        ref = [np.empty((T, w), np.float32) for w in (88, 88, 264)]
        rc = lib.bp_infer_pcm(model._handle, pcm.ctypes.data_as(fp), n, ch, rate, *[a.ctypes.data_as(fp) for a in ref], 0)
        assert rc == 0, model.last_error() if hasattr(model, "last_error") else rc
        # the raw samples at an odd offset inside a page-locked buffer, as the file pipeline hands them over
        n_out = T * 440 * 4
        host = lib.bp_host_alloc(len(raw) + 3 + n_out)
        assert host
        try:
            C.memmove(host + 3, raw, len(raw))
            out = host + 3 + len(raw)
            out += (-out) % 4
            host_out = lib.bp_host_alloc(n_out)
            assert host_out
            try:
                ptrs = [C.cast(host_out + o * 4, fp) for o in (0, T * 88, T * 176)]
                rc = lib.bp_infer_pcm_raw(model._handle, host + 3, fmt, n, ch, rate, *ptrs, 0)
                assert rc == 0
                got = np.ctypeslib.as_array(C.cast(host_out, fp), shape=(T * 440,)).copy()
            finally:
                lib.bp_host_free(host_out)
        finally:
            lib.bp_host_free(host)
        for a, o, w in zip(ref, (0, T * 88, T * 176), (88, 88, 264)):
            assert np.array_equal(a.view(np.uint32).ravel(), got[o : o + T * w].view(np.uint32)), (kind, ch, w)
    assert lib.bp_infer_pcm_raw(model._handle, None, 99, 10, 1, 22050, None, None, None, 0) == _native.BP_ERR_INVALID_ARG


@pytest.mark.gpu
def test_lanes_on_two_devices_in_one_process(tmp_path):
    """VERDICT r4 item 5c.  One native file job whose lanes live on different GPUs: with two or more devices on the box
    bp_transcribe_files gets handles created with different device ordinals and must write the bytes of the one-device
    job; on a one-GPU box the lane-to-device map of `lane_models` is asserted (device-major, `lanes` per device) and the job
    runs with the same device named twice."""
    import shutil

    from basic_pitch_amd import Model, transcribe_files
    from basic_pitch_amd.inference import lane_models

    lib = _native.load_library()
    n_dev = int(lib.bp_device_count())
    assert n_dev >= 1
    devs = [0, 1] if n_dev >= 2 else [0, 0]
    models = lane_models(devices=devs, lanes=2)
    try:
        assert [m.info()["device"] for m in models] == [devs[0], devs[0], devs[1], devs[1]]
        src = tmp_path / "in"
        src.mkdir()
        paths = []
        for i in range(6):
            shutil.copy(os.path.join(GOLDEN, "vocadito_10.wav"), src / f"c{i}.wav")
            paths.append(src / f"c{i}.wav")
        out_a, out_b = tmp_path / "a", tmp_path / "b"
        out_a.mkdir(), out_b.mkdir()
        rep = transcribe_files(paths, out_a, models=models, threads=4)
        assert [r["status"] for r in rep] == [0] * 6 and all(r["n_note_events"] == 28 for r in rep)
        one = Model(max_windows=64)
        rep1 = transcribe_files(paths[:1], out_b, models=[one])
        one.close()
        assert rep1[0]["status"] == 0
        ref_mid, ref_csv = (out_b / "c0_basic_pitch.mid").read_bytes(), (out_b / "c0_basic_pitch.csv").read_bytes()
        for i in range(6):
            assert (out_a / f"c{i}_basic_pitch.mid").read_bytes() == ref_mid and (out_a / f"c{i}_basic_pitch.csv").read_bytes() == ref_csv
        # handles of different modes in one job are refused (the buffers of a file are sized from the first lane)
        ext = Model(max_windows=8, ext_cqt_44k=True)
        with pytest.raises(ValueError):
            transcribe_files(paths[:1], tmp_path, models=[models[0], ext])
        ext.close()
    finally:
        for m in models:
            m.close()
