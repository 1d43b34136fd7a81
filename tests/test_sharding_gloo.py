"""N > 1 path on CPU: two processes over gloo run the file-sharded driver.

The per-file work here is the ORACLE on one tiny window (tests may use the oracle as the compute
stand-in; on the GPU box the same driver wraps Model.predict_track).  What is under test is the
protocol: deterministic LPT plan on every rank, disjoint coverage, per-item error isolation, host-side
gather to rank 0, no device collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, out_path: str) -> None:
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from basic_pitch_amd.sharding import plan_shards, run_sharded
    from oracle import bp_oracle as O

    W = O.load_weights()
    lengths = [50_000, 7_000, 43_000, 20_000, 90_000, 12_345, 64_000]
    seen = []

    def process(i):
        seen.append(i)
        if i == 3:
            raise RuntimeError("decode failed")  # one bad file must not sink the job
        rng = np.random.default_rng(100 + i)
        x = rng.uniform(-1, 1, (1, O.AUDIO_N_SAMPLES)).astype(np.float32)
        r = O.forward(x, W, np.float32)
        return {"rank": rank, "checksum": float(r["note"].sum() + r["onset"].sum() + r["contour"].sum())}

    merged = run_sharded(list(range(len(lengths))), lengths, process)
    plan = plan_shards(lengths, world)
    assert sorted(seen) == plan[rank]
    if rank == 0:
        assert sorted(merged.keys()) == list(range(len(lengths)))
        assert isinstance(merged[3], RuntimeError)
        for r_, shard in enumerate(plan):
            for i in shard:
                if i != 3:
                    assert merged[i]["rank"] == r_
        torch.save({k: (v if not isinstance(v, Exception) else str(v)) for k, v in merged.items()}, out_path)
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_file_sharding(tmp_path):
    port = _free_port()
    out_path = str(tmp_path / "merged.pt")
    mp.spawn(_worker, args=(2, port, out_path), nprocs=2, join=True)
    merged = torch.load(out_path)
    # single-process run gives the same per-file results (sharding does not change values)
    sys.path.insert(0, ROOT)
    from oracle import bp_oracle as O

    W = O.load_weights()
    for i in (0, 6):
        x = np.random.default_rng(100 + i).uniform(-1, 1, (1, O.AUDIO_N_SAMPLES)).astype(np.float32)
        r = O.forward(x, W, np.float32)
        ref = float(r["note"].sum() + r["onset"].sum() + r["contour"].sum())
        assert abs(merged[i]["checksum"] - ref) <= 1e-3 * abs(ref)


# ---- the product entry point: predict_many_sharded ------------------------------------------------------------------
class FakeModel:
    """Compute stand-in for a rank without a GPU: the same interface `predict_many` drives (`resample`,
    `predict_tracks`), posteriorgrams = a deterministic function of the samples with note-like ridges, so the REAL host
    code around it runs: file reads, shard plan, packing order, C++ note decoding, MIDI assembly, gathers."""

    def __init__(self, device):
        self.device = device

    def resample(self, pcm, sr):
        from basic_pitch_amd import audio

        return audio.resample(np.ascontiguousarray(audio.to_mono(pcm)), sr)

    def predict_tracks(self, signals):
        outs = []
        for y in signals:
            T = int(len(y) / 36164 * 142)
            seed = int(np.abs(y[:1000]).sum() * 1e6) % (2**31)
            rng = np.random.default_rng(seed)
            note = rng.uniform(0, 0.2, (T, 88)).astype(np.float32)
            onset = rng.uniform(0, 0.2, (T, 88)).astype(np.float32)
            contour = rng.uniform(0, 0.2, (T, 264)).astype(np.float32)
            for _ in range(6):
                f, t0 = int(rng.integers(5, 80)), int(rng.integers(0, max(1, T - 40)))
                note[t0 : t0 + 30, f] = 0.8
                onset[t0, f] = 0.9
            outs.append({"note": note, "onset": onset, "contour": contour, "device": self.device})
        return outs


def fake_factory(device):
    return FakeModel(device)


def _write_clips(tmp_path, n):
    import wave

    paths = []
    for i in range(n):
        rng = np.random.default_rng(50 + i)
        x = (rng.uniform(-0.5, 0.5, 22050 * (2 + 3 * (i % 4))) * 32767).astype("<i2")
        p = tmp_path / f"clip_{i}.wav"
        with wave.open(str(p), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(22050)
            w.writeframes(x.tobytes())
        paths.append(str(p))
    return paths


def _summary(results):
    out = []
    for r in results:
        if isinstance(r, Exception):
            out.append(("error", type(r).__name__))
        else:
            mo, midi, ev = r
            out.append((mo["device"], mo["note"].shape[0], float(mo["note"].sum()), len(ev),
                        [(e[0], e[1], e[2]) for e in ev], len(midi.instruments)))
    return out


def _sharded_worker(rank, world, port, paths, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basic_pitch_amd import predict_many_sharded

    res = predict_many_sharded(paths, model_factory=fake_factory, group=2, decode_threads=2)
    if rank == 0:
        torch.save(_summary(res), out_path)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_predict_many_sharded_two_ranks_gloo_and_spawn(tmp_path):
    """The product's multi-GPU entry point with the compute stubbed (FakeModel): inside a 2-rank gloo job, as 2 spawned
    workers from a plain process, and in one process — all three return the same per-file results in input order, each
    file computed on the rank the LPT plan gives it, a missing file reported in place without sinking the job."""
    from basic_pitch_amd import predict_many_sharded
    from basic_pitch_amd.sharding import _file_costs, plan_shards

    paths = _write_clips(tmp_path, 7)
    paths.insert(3, str(tmp_path / "missing.wav"))
    single = _summary(predict_many_sharded(paths, gpus=1, model_factory=fake_factory, group=3, decode_threads=2))
    assert single[3] == ("error", "ValueError") and all(s[0] == 0 for i, s in enumerate(single) if i != 3)
    assert all(s[3] > 0 for i, s in enumerate(single) if i != 3), "the stand-in posteriorgrams must decode to notes"

    out_path = str(tmp_path / "sharded.pt")
    mp.spawn(_sharded_worker, args=(2, _free_port(), paths, out_path), nprocs=2, join=True)
    ranked = torch.load(out_path)
    spawned = _summary(predict_many_sharded(paths, gpus=2, model_factory=fake_factory, group=2, decode_threads=2))
    plan = plan_shards(_file_costs(paths), 2)
    owner = {i: r for r, shard in enumerate(plan) for i in shard}
    assert all(len(s) >= 2 for s in plan)
    for got in (ranked, spawned):
        assert len(got) == len(paths)
        for i, (g, s) in enumerate(zip(got, single)):
            if i == 3:
                assert g == s
            else:
                assert g[0] == owner[i] and g[1:] == s[1:], i


def test_predict_and_save_sharded_writes_per_worker(tmp_path):
    """The batch job entry point with the compute stubbed: 2 spawned workers predict AND write their own shares (MIDI +
    note CSV + .npz), only small reports come back; the files are byte-identical to the single-process job's, a missing
    input is reported in place."""
    from basic_pitch_amd import predict_and_save_many, predict_and_save_sharded

    paths = _write_clips(tmp_path, 6)
    paths.insert(2, str(tmp_path / "missing.wav"))
    one, two = tmp_path / "one", tmp_path / "two"
    one.mkdir()
    two.mkdir()
    ref = predict_and_save_many(paths, one, True, False, True, True, model_or_model_path=FakeModel(0), group=4,
                                decode_threads=2, return_exceptions=True)
    got = predict_and_save_sharded(paths, two, True, False, True, True, gpus=2, model_factory=fake_factory, group=2,
                                   decode_threads=2)
    assert len(got) == len(ref) == len(paths)
    assert isinstance(got[2], ValueError) and isinstance(ref[2], ValueError)
    for i, (g, r) in enumerate(zip(got, ref)):
        if i == 2:
            continue
        assert g["n_note_events"] == r["n_note_events"] > 0
        assert sorted(g["outputs"]) == ["midi", "model_output", "note_events"]
        for kind in ("midi", "note_events"):
            a, b = open(g["outputs"][kind], "rb").read(), open(r["outputs"][kind], "rb").read()
            assert a == b and len(a) > 0, (i, kind)
        a = np.load(g["outputs"]["model_output"], allow_pickle=True)["basic_pitch_model_output"].item()
        b = np.load(r["outputs"]["model_output"], allow_pickle=True)["basic_pitch_model_output"].item()
        assert all(np.array_equal(a[k], b[k]) for k in ("note", "onset", "contour"))
    # the reference's behaviour without return_exceptions: the first failure propagates (inference.py:603-604)
    with pytest.raises(ValueError):
        predict_and_save_many(paths, one, True, False, False, False, model_or_model_path=FakeModel(0))


def dying_factory(device):
    """A worker whose process ends without a Python exception (what a HIP abort, a segfault or the OOM killer look like
    from the parent): worker 1 exits hard before it has posted anything."""
    if device == 1:
        os._exit(17)
    return FakeModel(device)


class _Unpicklable(FakeModel):
    def predict_tracks(self, signals):
        outs = super().predict_tracks(signals)
        for o in outs:
            o["device"] = lambda: None  # a result the queue cannot pickle
        return outs


def unpicklable_factory(device):
    return _Unpicklable(device)


def test_sharded_parent_notices_a_worker_that_died_natively(tmp_path):
    """the parent polled `queue.get()` without a timeout, so a worker that died without posting (or whose
    result could not be pickled in the queue's feeder thread) hung the job forever.  Both now surface as RuntimeError."""
    import time

    from basic_pitch_amd import predict_and_save_sharded, predict_many_sharded

    paths = _write_clips(tmp_path, 4)
    t0 = time.time()
    with pytest.raises(RuntimeError, match="exited with code 17"):
        predict_many_sharded(paths, gpus=2, model_factory=dying_factory, group=2, decode_threads=1)
    out = tmp_path / "out"
    out.mkdir()
    with pytest.raises(RuntimeError, match="exited with code 17"):
        predict_and_save_sharded(paths, out, True, False, False, False, gpus=2, model_factory=dying_factory, group=2,
                                 decode_threads=1)
    with pytest.raises(RuntimeError, match="worker failed"):
        predict_many_sharded(paths, gpus=2, model_factory=unpicklable_factory, group=2, decode_threads=1)
    assert time.time() - t0 < 120


def test_predict_and_save_many_is_one_pipeline_and_resolves_duplicate_stems(tmp_path):
    """`predict_and_save_many` used to call `predict_many` once per group (no read-ahead, no overlap
    between groups), and two inputs with the same stem could race on the exists-check.  Now it is one pipeline over all
    files (every group but the first is read while its predecessor computes) and a later input with an already-claimed
    stem gets the reference's IOError in place — in the single-process job and in the sharded one."""
    import shutil

    from basic_pitch_amd import predict_and_save_many, predict_and_save_sharded

    paths = _write_clips(tmp_path, 5)
    other = tmp_path / "elsewhere"
    other.mkdir()
    shutil.copy(paths[1], other / "clip_1.wav")  # same stem as paths[1]
    paths.append(str(other / "clip_1.wav"))
    calls = []

    class Recording(FakeModel):
        def predict_tracks(self, signals):
            calls.append(len(signals))
            return super().predict_tracks(signals)

    out = tmp_path / "out"
    out.mkdir()
    rep = predict_and_save_many(paths, out, True, False, False, True, model_or_model_path=Recording(0), group=2,
                                decode_threads=2, return_exceptions=True)
    assert calls == [2, 2, 1]  # 5 unique files in groups of 2 through ONE predict_many call
    assert isinstance(rep[5], IOError) and "clip_1" in str(rep[5])
    assert all(r["n_note_events"] > 0 and os.path.exists(r["outputs"]["midi"]) for r in rep[:5])
    # return_exceptions=False follows the reference's sequential loop (inference.py:548-604): everything in
    # front of the duplicate is predicted and written, THEN its IOError is raised
    out2 = tmp_path / "out2"
    out2.mkdir()
    with pytest.raises(IOError, match="clip_1"):
        predict_and_save_many(paths, out2, True, False, False, False, model_or_model_path=FakeModel(0))
    assert sorted(os.listdir(out2)) == [f"clip_{i}_basic_pitch.mid" for i in range(5)]
    # nothing is saved: nothing can collide
    rep_ns = predict_and_save_many(paths, tmp_path, False, False, False, False, model_or_model_path=FakeModel(0))
    assert all(isinstance(r, dict) and r["outputs"] == {} for r in rep_ns)
    # the earlier namesake could not be read and wrote nothing: the later file is an ordinary file after all
    broken = tmp_path / "broken"
    broken.mkdir()
    (broken / "clip_1.wav").write_bytes(b"not audio")
    out4 = tmp_path / "out4"
    out4.mkdir()
    rep4 = predict_and_save_many([str(broken / "clip_1.wav"), paths[1]], out4, True, False, False, False,
                                 model_or_model_path=FakeModel(0), return_exceptions=True)
    assert isinstance(rep4[0], Exception) and isinstance(rep4[1], dict) and os.path.exists(rep4[1]["outputs"]["midi"])
    out3 = tmp_path / "out3"
    out3.mkdir()
    rep3 = predict_and_save_sharded(paths, out3, True, False, False, True, gpus=2, model_factory=fake_factory, group=2,
                                    decode_threads=1)
    assert isinstance(rep3[5], IOError)
    assert [r["n_note_events"] for r in rep3[:5]] == [r["n_note_events"] for r in rep[:5]]


def test_save_shard_native_routes_through_transcribe_files(tmp_path, monkeypatch):
    """`predict_and_save_sharded(..., native=True)`: a worker's share goes through `inference.transcribe_files` (the native
    pipeline; stubbed here — it needs a GPU, tests/test_file_pipeline.py covers it there) and its per-file statuses come
    back as the reports of the Python pipeline: dicts with the output paths, or the exception of the file."""
    from basic_pitch_amd import inference, sharding

    paths = _write_clips(tmp_path, 3)
    seen = {}

    def fake_transcribe(audio_paths, out_dir, save_midi, save_notes, models=None, threads=0, **kw):
        seen.update(paths=list(audio_paths), threads=threads, kw=kw, lanes=len(models))
        return [{"status": 0, "n_note_events": 7, "n_frames": 100, "message": ""},
                {"status": -7, "n_note_events": 0, "n_frames": 0, "message": "x: not a WAV or FLAC file"},
                {"status": -1, "n_note_events": 0, "n_frames": 0, "message": "y already exists and would be overwritten."}]

    monkeypatch.setattr(inference, "transcribe_files", fake_transcribe)
    out = tmp_path / "o"
    out.mkdir()
    rep = sharding._save_shard(paths, [0, 1, 2], 0, None, fake_factory, out,
                               {"save_midi": True, "sonify_midi": False, "save_model_outputs": False, "save_notes": True},
                               {"native": True, "native_threads": 4, "onset_threshold": 0.6, "group": 8, "decode_threads": 2})
    assert seen["paths"] == paths and seen["threads"] == 4 and seen["kw"] == {"onset_threshold": 0.6} and seen["lanes"] == 1
    assert rep[0] == {"n_note_events": 7, "outputs": {"midi": str(out / "clip_0_basic_pitch.mid"),
                                                       "note_events": str(out / "clip_0_basic_pitch.csv")}}
    assert isinstance(rep[1], ValueError) and isinstance(rep[2], IOError)
    with pytest.raises(ValueError, match="native=True"):
        sharding._save_shard(paths, [0], 0, None, fake_factory, out,
                             {"save_midi": True, "sonify_midi": True, "save_model_outputs": False, "save_notes": True}, {"native": True})


# ---- the window-range fallback: one long file among short ones is cut into pieces ------------------------
class WindowFake(FakeModel):
    """A stand-in whose posteriorgrams are a function of each WINDOW's samples (as the real graph's are), with both entry
    points `predict_many_sharded` drives: `predict_tracks` (whole files: host windowing -> predict -> unwrap, the
    reference's structure) and `predict` (a piece's windows).  Split and unsplit results can then be compared bit for bit."""

    max_windows = 4

    def predict(self, x):
        x = np.asarray(x, np.float32).reshape(len(x), -1)
        note = np.zeros((len(x), 172, 88), np.float32)
        onset = np.zeros((len(x), 172, 88), np.float32)
        contour = np.zeros((len(x), 172, 264), np.float32)
        for i, w in enumerate(x):
            rng = np.random.default_rng(int(np.abs(w[4000:9000]).sum() * 1e5) % (2**31))
            note[i] = rng.uniform(0, 0.2, (172, 88))
            onset[i] = rng.uniform(0, 0.2, (172, 88))
            contour[i] = rng.uniform(0, 0.2, (172, 264))
            f, t0 = int(rng.integers(5, 80)), int(rng.integers(20, 100))
            note[i, t0 : t0 + 40, f] = 0.8
            onset[i, t0, f] = 0.9
        return {"note": note, "onset": onset, "contour": contour}

    def predict_tracks(self, signals):
        from basic_pitch_amd import inference as inf

        outs = []
        for y in signals:
            padded = np.concatenate([np.zeros(3840, np.float32), np.asarray(y, np.float32)])
            wins = [w[:, 0] for w, _ in inf.window_audio_file(padded, 36164)]
            parts = {"note": [], "onset": [], "contour": []}
            for a in range(0, len(wins), 3):  # another batching than predict_window_range's
                for k, v in self.predict(np.stack(wins[a : a + 3])).items():
                    parts[k].append(v)
            outs.append({k: inf.unwrap_output(np.concatenate(v), len(y), 30, 36164) for k, v in parts.items()})
        return outs


def window_fake_factory(device):
    return WindowFake(device)


def _write_long_and_short(tmp_path):
    import wave

    paths = []
    for i, seconds in enumerate((2, 61, 3, 2)):
        rng = np.random.default_rng(70 + i)
        x = (rng.uniform(-0.5, 0.5, 22050 * seconds) * 32767).astype("<i2")
        p = tmp_path / f"take_{i}.wav"
        with wave.open(str(p), "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(22050)
            w.writeframes(x.tobytes())
        paths.append(str(p))
    return paths


def _digest(results):
    import hashlib

    out = []
    for r in results:
        if isinstance(r, Exception):
            out.append(("error", type(r).__name__, str(r)[:40]))
        else:
            mo, midi, ev = r
            out.append((mo["note"].shape, hashlib.sha256(b"".join(np.ascontiguousarray(mo[k]).tobytes() for k in ("note", "onset", "contour"))).hexdigest(),
                        [(float(e[0]), float(e[1]), int(e[2]), float(e[3])) for e in ev]))
    return out


def _split_worker(rank, world, port, paths, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from basic_pitch_amd import predict_many_sharded

    res = predict_many_sharded(paths, model_factory=window_fake_factory, group=2, decode_threads=2)
    if rank == 0:
        torch.save(_digest(res), out_path)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_one_long_file_is_split_by_window_range_over_two_ranks(tmp_path):
    """SURVEY.md 8e / VERDICT r5 missing #3: `split_windows` is part of the product's plan.  A 61-second take among three
    short ones outweighs an even share of a 2-rank job, so `plan_units` cuts it into two window ranges; each rank decodes the
    file, computes its own windows (`predict_window_range`), rank 0 concatenates the rows and decodes the notes over the whole
    file.  Inside a 2-rank gloo job and as 2 spawned workers the result is BIT-EQUAL to the unsplit single-process result
    (posteriorgram bytes and note events) — windows are independent — and a long file that cannot be read is reported in
    place."""
    from basic_pitch_amd import predict_many_sharded
    from basic_pitch_amd.sharding import _file_costs, plan_units, split_windows

    paths = _write_long_and_short(tmp_path)
    units, shards = plan_units(_file_costs(paths), 2)
    assert (1, 0, 2) in units and (1, 1, 2) in units and len(units) == 5
    owners = {units[u]: r for r, sh in enumerate(shards) for u in sh}
    assert owners[(1, 0, 2)] != owners[(1, 1, 2)]  # the two halves of the long file run on different ranks
    single = _digest(predict_many_sharded(paths, gpus=1, model_factory=window_fake_factory, group=2, decode_threads=2))
    assert single[1][0][0] == int(61 * 22050 / 36164 * 142) and len(single[1][2]) > 10
    out_path = str(tmp_path / "split.pt")
    mp.spawn(_split_worker, args=(2, _free_port(), paths, out_path), nprocs=2, join=True)
    ranked = torch.load(out_path)
    spawned = _digest(predict_many_sharded(paths, gpus=2, model_factory=window_fake_factory, group=2, decode_threads=2))
    assert ranked == single and spawned == single
    # a job of ONE file: every rank takes a window range of it
    solo = _digest(predict_many_sharded(paths[1:2], gpus=2, model_factory=window_fake_factory, decode_threads=1))
    assert solo == single[1:2]
    n_win = len(range(0, 61 * 22050 + 3840, 36164))
    assert split_windows(n_win, 2) == [(0, 19), (19, 38)] and n_win == 38
    # a long file that is not audio: its pieces fail on both ranks, the file's entry is the exception, the others are there
    bad = tmp_path / "long_bad.wav"
    bad.write_bytes(b"RIFF" + bytes(3_000_000))
    res = predict_many_sharded([paths[0], str(bad), paths[2]], gpus=2, model_factory=window_fake_factory, decode_threads=1)
    assert isinstance(res[1], Exception) and not isinstance(res[0], Exception) and not isinstance(res[2], Exception)
