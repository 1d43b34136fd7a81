"""CPU-side tests: the C-ABI library loads and exports what the header declares, host logic mirrors
the reference (windowing, un-overlapping, audio ingest, shard planning).  No compute calls."""
import ctypes as C
import os
import re
import struct
import wave

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from oracle import bp_oracle as O


@pytest.fixture(scope="module")
def lib():
    from basic_pitch_amd import _native, build

    build.build_library()
    return _native.load_library()


def test_library_exports_every_header_symbol(lib):
    from basic_pitch_amd import _native

    header = open(os.path.join(ROOT, "include", "basic_pitch_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(bp_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_native.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert b"gfx950" in lib.bp_version()


def test_library_is_in_tree():
    from basic_pitch_amd import build

    assert os.path.realpath(build.LIB_PATH).startswith(os.path.realpath(ROOT))


def test_track_geometry_matches_reference_formulas(lib):
    """inference.py:207,242 (window count) and 277-279 (row count) for many lengths."""
    rng = np.random.default_rng(0)
    lengths = [1, 100, 3839, 3840, 3841, 32324, 32325, 36164, 36165, 43844, 200607, 3_969_000] + [
        int(v) for v in rng.integers(1, 5_000_000, 200)
    ]
    for n in lengths:
        padded = n + 3840
        n_win = len(range(0, padded, 36164))
        assert lib.bp_track_n_windows(n) == n_win, n
        rows = min(n_win * 142, int(n / 36164 * 142))
        assert lib.bp_track_n_frames(n) == rows, n
    assert lib.bp_track_n_windows(0) == 0 and lib.bp_track_n_frames(0) == 0
    assert lib.bp_track_n_windows(200607) == 6 and lib.bp_track_n_frames(200607) == 787


def test_pyramid_layout(lib):
    lens = [21922, 10961, 5480, 2740, 1370, 685, 342, 171]
    end = 0
    for k in range(1, 9):
        off, ln = C.c_int64(), C.c_int64()
        assert lib.bp_pyramid_layout(k, C.byref(off), C.byref(ln)) == 0
        assert ln.value == lens[k - 1] and off.value >= end and off.value % 4 == 0
        end = off.value + ln.value
    assert end <= 43712
    assert lib.bp_pyramid_layout(0, C.byref(off), C.byref(ln)) != 0


def test_create_rejects_bad_weights(lib):
    h = C.c_void_p()
    assert lib.bp_create(b"garbage", 7, 0, 0, 0, C.byref(h)) == -2
    assert b"magic" in lib.bp_last_error(None)
    blob = open(os.path.join(ROOT, "basic_pitch_amd", "assets", "nmp_weights.bin"), "rb").read()
    assert lib.bp_create(blob[:1000], 1000, 0, 0, 0, C.byref(h)) == -2  # truncated
    # a blob with a missing tensor
    n = struct.unpack_from("<I", blob, 12)[0]
    renamed = bytearray(blob)
    renamed[16 : 16 + 24] = b"not_a_tensor".ljust(24, b"\0")
    assert lib.bp_create(bytes(renamed), len(renamed), 0, 0, 0, C.byref(h)) == -2
    assert n == 19


def test_model_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from basic_pitch_amd import Model
    from basic_pitch_amd._native import NativeLibraryError

    with pytest.raises(NativeLibraryError, match="no HIP device"):
        Model()
    with pytest.raises(ValueError):
        Model("/nonexistent/model.bin")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "basic_pitch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "bp_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f


def test_window_and_unwrap_mirror_reference(clip_22k):
    from basic_pitch_amd import inference as inf

    hop = inf.AUDIO_N_SAMPLES - 30 * inf.FFT_HOP
    wins = list(inf.window_audio_file(clip_22k, hop))
    assert len(wins) == 6  # tests/test_inference.py:168
    for w, times in wins:
        assert w.shape == (43844, 1) and times["start"] <= times["end"]
    assert np.array_equal(clip_22k[:43844], wins[0][0][:, 0])
    got = list(inf.get_audio_input(os.path.join(GOLDEN, "vocadito_10.wav"), 30 * 256, hop))
    assert len(got) == 6 and got[0][2] == 200607 and got[0][0].shape == (1, 43844, 1)
    ow, n = O.window_track(clip_22k)
    assert np.array_equal(np.concatenate([g[0][:, :, 0] for g in got]), ow)
    # unwrap: same rows as the oracle's restatement
    rng = np.random.default_rng(0)
    out = rng.random((6, 172, 88)).astype(np.float32)
    a = inf.unwrap_output(out, 200607, 30, hop)
    b = O.unwrap_output(out, 200607)
    assert a.shape == (787, 88) and np.array_equal(a, b)
    assert inf.unwrap_output(out[0], 200607, 30, hop) is None  # rank != 3 (inference.py:264-265)


def _write_wav(path, data, sr, sampwidth):
    with wave.open(path, "wb") as w:
        w.setnchannels(data.shape[1])
        w.setsampwidth(sampwidth)
        w.setframerate(sr)
        w.writeframes(data.tobytes())


def test_wav_reader_and_resampler(tmp_path):
    from basic_pitch_amd import audio

    sr = 44100
    t = np.arange(sr) / sr
    sig = 0.5 * np.sin(2 * np.pi * 440 * t)
    st = np.stack([sig, 0.5 * sig], axis=1)
    p16 = str(tmp_path / "a16.wav")
    _write_wav(p16, (st * 32767).astype("<i2"), sr, 2)
    x, fs = audio.read_wav(p16)
    assert fs == sr and x.shape == (sr, 2) and np.abs(x[:, 0] - sig).max() < 1e-4
    y, fs2 = audio.load(p16)
    assert fs2 == 22050 and y.dtype == np.float32 and y.shape == (22050,)
    ref = 0.75 * 0.5 * np.sin(2 * np.pi * 440 * np.arange(22050) / 22050)
    assert np.abs(y[200:-200] - ref[200:-200]).max() < 2e-3
    p32 = str(tmp_path / "a32.wav")
    _write_wav(p32, (st * (2**31 - 1)).astype("<i4"), sr, 4)
    x32, _ = audio.read_wav(p32)
    assert np.abs(x32[:, 0] - sig).max() < 1e-6
    # 24-bit
    v = (st[:, :1] * (2**23 - 1)).astype(np.int32)
    b = np.zeros((v.shape[0], 3), np.uint8)
    b[:, 0], b[:, 1], b[:, 2] = v[:, 0] & 255, (v[:, 0] >> 8) & 255, (v[:, 0] >> 16) & 255
    p24 = str(tmp_path / "a24.wav")
    with wave.open(p24, "wb") as w:
        w.setnchannels(1), w.setsampwidth(3), w.setframerate(sr), w.writeframes(b.tobytes())
    x24, _ = audio.read_wav(p24)
    assert np.abs(x24[:, 0] - sig).max() < 1e-6
    assert abs(audio.get_duration(p16) - 1.0) < 1e-9
    with pytest.raises(ValueError):
        (tmp_path / "bad.wav").write_bytes(b"nope")
        audio.read_wav(str(tmp_path / "bad.wav"))


def test_golden_clip_ingest(clip_22k):
    assert clip_22k.dtype == np.float32 and np.abs(clip_22k).max() <= 1.0


def test_shard_planning():
    from basic_pitch_amd.sharding import plan_shards, shard_imbalance, split_windows

    costs = [3_969_000] * 1000  # BASELINE.json configs[2]: 1000 equal 3-minute tracks over 8 GPUs
    shards = plan_shards(costs, 8)
    assert sorted(i for s in shards for i in s) == list(range(1000))
    assert {len(s) for s in shards} == {125}
    rng = np.random.default_rng(1)
    ragged = rng.integers(10_000, 10_000_000, 333).tolist()
    sh = plan_shards(ragged, 8)
    assert sorted(i for s in sh for i in s) == list(range(333))
    assert shard_imbalance(ragged, sh) < 1.02
    assert plan_shards([], 4) == [[], [], [], []]
    assert plan_shards([5.0], 2) == [[0], []]
    assert split_windows(110, 8) == [(0, 14), (14, 28), (28, 42), (42, 56), (56, 70), (70, 84), (84, 97), (97, 110)]
    with pytest.raises(ValueError):
        plan_shards([1], 0)


def test_ort_shim_contract_cpu(tmp_path):
    """basic_pitch_amd.ort_shim presents the three onnxruntime entry points the reference uses (inference.py:134-136,
    173-180); without a GPU the session must fail loudly (no CPU fallback), and a foreign model file is a ValueError."""
    import basic_pitch_amd.ort_shim as shim
    from basic_pitch_amd._native import NativeLibraryError

    assert shim.get_available_providers() == ["MI355XExecutionProvider"]
    bad = tmp_path / "other.onnx"
    bad.write_bytes(b"not the reference model")
    with pytest.raises(ValueError):
        shim.InferenceSession(str(bad), providers=shim.get_available_providers())
    assert shim.OUTPUT_KEYS == {"StatefulPartitionedCall:1": "note", "StatefulPartitionedCall:2": "onset",
                                "StatefulPartitionedCall:0": "contour"}
    import sys

    assert "onnxruntime" not in sys.modules or sys.modules["onnxruntime"] is not shim  # never installed implicitly


REF_ONNX = "/root/reference/basic_pitch/saved_models/icassp_2022/nmp.onnx"


@pytest.mark.skipif(not os.path.exists(REF_ONNX), reason="needs the reference checkout (dev container only)")
def test_model_blob_is_extracted_from_the_reference_onnx_at_load(tmp_path):
    """`Model(path)` takes the serialized model like the reference (inference.py:78-154): nmp.onnx is structure-checked
    and its 18 constants extracted at load (basic_pitch_amd/weights.py) — byte-identical to the shipped blob; the other
    artifacts of the same model resolve to the nmp.onnx next to them; anything else is a ValueError."""
    import hashlib
    import shutil

    from basic_pitch_amd import weights
    from basic_pitch_amd.inference import ICASSP_2022_MODEL_PATH

    shipped = open(ICASSP_2022_MODEL_PATH, "rb").read()
    assert hashlib.sha256(open(REF_ONNX, "rb").read()).hexdigest() == weights.NMP_ONNX_SHA256
    assert weights.load_model_blob(REF_ONNX) == shipped
    assert weights.load_model_blob(ICASSP_2022_MODEL_PATH) == shipped
    d = os.path.dirname(REF_ONNX)
    for other in ("nmp", "nmp.tflite", "nmp.mlpackage"):  # what basic_pitch.ICASSP_2022_MODEL_PATH may point at
        assert weights.load_model_blob(os.path.join(d, other)) == shipped, other
    bad = tmp_path / "x.onnx"
    bad.write_bytes(b"\x08\x07\x12\x04abcd")
    with pytest.raises(ValueError):
        weights.load_model_blob(bad)
    with pytest.raises(ValueError):
        weights.load_model_blob(tmp_path / "missing.onnx")
    (tmp_path / "lonely.tflite").write_bytes(b"x")
    with pytest.raises(ValueError):
        weights.load_model_blob(tmp_path / "lonely.tflite")
    # a model with the right structure but other weights (fine-tuned) is accepted; a truncated file is not
    data = bytearray(open(REF_ONNX, "rb").read())
    trunc = tmp_path / "t.onnx"
    trunc.write_bytes(bytes(data[: len(data) // 2]))
    with pytest.raises(ValueError):
        weights.load_model_blob(trunc)
    shutil.copy(REF_ONNX, tmp_path / "nmp.onnx")
    assert weights.load_model_blob(tmp_path / "nmp.onnx") == shipped


def test_no_kernel_of_the_product_library_uses_scratch():
    """Every kernel the product library carries fits its registers: no spill, `.private_segment_fixed_size` 0 (round-4
    review: the two CQT kernels were the only default-path kernels with scratch).  Read from the code objects inside the
    built library — what will actually run — not from a recompile."""
    import sys

    from basic_pitch_amd import build

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import kernel_resources

    rows = kernel_resources(build.build_library())
    assert len(rows) >= 40, len(rows)
    names = " ".join(r["name"] for r in rows)
    for must in ("pl_pyramid_window_kernel", "cqt_filterbank_planes_kernel", "contour_conv1_march_kernel",
                 "contour_conv1_rim_kernel", "contour_conv1_rim_march_kernel", "contour_conv2_proj_kernel", "note_march16_kernel", "onset_march16_kernel", "flac_decode_kernel"):
        assert must in names, must
    # kernels that left the product library (the A/B library carries them): the 32x32x16 marches, the workgroup branch
    # kernel and the fp8-corrections mode's kernels
    for gone in ("note_march_kernel", "branch_kernel", "contour_conv1_fold_mx_kernel", "onset_march_kernel", "contour_conv2_kernel"):
        assert gone + "<" not in names and gone + "(" not in names, gone
    bad = [(r["name"], r["scratch"], r["vgpr_spill"], r["sgpr_spill"]) for r in rows
           if r["scratch"] or r["vgpr_spill"] or r["sgpr_spill"]]
    assert not bad, bad
    for r in rows:
        assert r["lds"] <= 160 * 1024, r


def test_product_library_reads_no_behaviour_switch_from_the_environment():
    """The BP_* switches and the superseded kernels they select live in the A/B library only (build_library(ab=True),
    -DBP_AB_KERNELS): the product library does not even import getenv, and its sources call it in one place, the A/B gate."""
    import subprocess

    from basic_pitch_amd import build

    und = subprocess.run(["nm", "-D", "--undefined-only", build.build_library()], capture_output=True, text=True).stdout
    assert "getenv" not in und
    n = 0
    for f in os.listdir(build.CSRC):
        n += len(re.findall(r"\bgetenv\s*\(", open(os.path.join(build.CSRC, f)).read()))
    assert n <= 2, n
    for src in build.AB_SOURCES:
        assert src not in build.SOURCES


def test_native_file_reader_direct_io_reads_the_same_bytes(tmp_path):
    """bp_transcribe_files' reader (csrc/file_pipeline.cpp read_file_into) with and without O_DIRECT: the same bytes as
    Python's read for lengths around the 4 KiB block size (an O_DIRECT read is issued in whole blocks and stops at the end of
    the file), an empty file included; a file system that refuses O_DIRECT is read through the page cache and says so; a missing file is an error, not a crash."""
    import ctypes as C

    from basic_pitch_amd import _native

    lib = _native.load_library()

    def fnv(b):
        h = 1469598103934665603
        for x in b:
            h = ((h ^ x) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        return h

    rng = np.random.default_rng(3)
    used_any = False
    for n in (0, 1, 4095, 4096, 4097, 8192, 100_000, 1_048_576 + 17):
        p = tmp_path / f"f{n}.bin"
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        p.write_bytes(data)
        for direct in (0, 1):
            h, used = C.c_uint64(), C.c_int(-1)
            got = lib.bp_files_read_probe(os.fsencode(str(p)), direct, C.byref(h), C.byref(used))
            assert got == n and h.value == fnv(data), (n, direct, got)
            assert used.value in (0, 1) and (direct or used.value == 0)
            used_any = used_any or used.value == 1
    before = lib.bp_files_direct_reads()
    h, used = C.c_uint64(), C.c_int(-1)
    assert lib.bp_files_read_probe(os.fsencode(str(tmp_path / "missing.bin")), 1, C.byref(h), C.byref(used)) < 0
    assert lib.bp_files_direct_reads() == before
    if os.path.isdir("/dev/shm"):  # tmpfs: refuses O_DIRECT before Linux 6.6 (buffered fallback), takes it as a no-op since
        q = "/dev/shm/bp_direct_probe_%d.bin" % os.getpid()
        try:
            open(q, "wb").write(b"x" * 5000)
            assert lib.bp_files_read_probe(os.fsencode(q), 1, C.byref(h), C.byref(used)) == 5000 and h.value == fnv(b"x" * 5000)
        finally:
            os.unlink(q)
    print("O_DIRECT used on", tmp_path, ":", used_any)
