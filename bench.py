#!/usr/bin/env python
"""bench.py — BASELINE.json's metric on MI355X: audio windows/sec, end-to-end CQT + CNN.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (bp_infer_async: pyramid -> filterbank + normalise -> 3 branches of the CNN
-> three posteriorgrams) over one batch of 256 synthetic 2-second 22.05 kHz windows that is already
resident in HBM (BASELINE.json configs[1]: "Batch=256 synthetic 2 s @ 22.05 kHz mono windows,
1xMI355X, fp32"); outputs stay in HBM.  Every rank owns its own batch (windows are independent
units: file/window sharding, no collective on the data path — SURVEY.md §8e), so scaling is weak
and `value` = windows processed by all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel (contour_conv1_march_kernel: Conv2D 8->8 3x39, 65 % of the path's FLOPs): algorithmic FLOP
                per launch / mean launch duration measured with HIP events on the kernel's stream over the
                timed steps, against the dense f16 MFMA peak (2.5 PFLOP/s; the kernel spends three f16 MFMAs per
                product — hi*hi + lo*hi + hi*lo, the fp32-class split — on a folded operator with 528 instead of 936
                products per output, so frac <= 0.59 by construction — executed_frac is the matrix-pipe occupancy)
  cpu_baseline  the oracle's C restatement of the frozen graph (fp32, AVX2, OpenMP over windows, all host
                cores) timed on rank 0 on a bounded sample of the same synthetic windows ("port": the
                reference's own ONNX/TF runtimes are not installable here)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 256
FLOP_PER_WINDOW = 1_048_159_296
BYTES_PER_WINDOW = 478_096               # fp32 I/O: 175,376 in + 302,720 out
C1_FLOP_PER_WINDOW = 680_030_208         # contour conv1: 2*8*8*3*39*172*264 (models.py:241-250)
F32_MFMA_PEAK_TFLOPS = 157.3             # MI355X_MICROARCH.md: dense f32 matrix peak (= f32 vector peak)
F16_MFMA_PEAK_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense f16/bf16 matrix peak
# contour_conv1_folded_kernel (conv_contour_direct.hip) computes the 56 interior groups of every frame (bins 20..243,
# 56/66 of conv1's products): 172 x 56 positions = 38 rounds of 256; per round each of the 8 waves issues 36 k-steps x 3
# MFMAs (hi*hi, lo*hi, hi*lo).  BP_CONV1=full: the exact kernel over all 66 groups, 45 rounds x 8 waves x 63 x 3.
F1_SHARE = 56.0 / 66.0
F1_EXECUTED_FLOP_PER_WINDOW = 38 * 8 * 36 * 3 * (2 * 32 * 32 * 16)
F1_BYTES_PER_WINDOW = 174 * 448 * 4 + 172 * 224 * 8 * 4  # zp read + interior c1 written
# contour_conv1_march_kernel (conv_contour_march.hip, the default since round 4): 7 strips of 32 bins x 4 frame chunks per
# window; a chunk of 43 frames marches over 45 z rows (2 rows of warm-up), 54 v_mfma_f32_16x16x32_f16 per row (3 frame taps
# x 6 k-steps x 3 split products)
M1_EXECUTED_FLOP_PER_WINDOW = 7 * 4 * (43 + 2) * 54 * (2 * 16 * 16 * 32)
D1_EXECUTED_FLOP_PER_WINDOW = 45 * 8 * 63 * 3 * (2 * 32 * 32 * 16)
D1_BYTES_PER_WINDOW = 174 * 448 * 4 + 172 * 264 * 8 * 4  # zp read + c1 written
HBM_PEAK_GBS = 8000.0
DTYPE_F16 = "f32 I/O + accumulate, split-f16 (22-bit hi+lo) MFMA operands, all three products of every layer on f16 MFMA"
DTYPE_DEFAULT = DTYPE_F16  # since round 3 the default IS the all-f16 split (fp32-class)
PMC_PROFILE = "r06_b"  # the committed rocprofv3 --pmc profile `roofline.traffic` is read from


def pmc_traffic(kernel_key: str, batch: int):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes, gfx950 x2 correction on the read side — profiles/r01_m_pmc.md).  PMC
    counters cannot be collected from inside the timed run, so this is the per-launch figure of the same
    command at the same batch, or None when no profile for this batch is committed."""
    path = os.path.join(ROOT, "profiles", PMC_PROFILE + "_pmc.json")
    if batch != 256 or not os.path.exists(path):
        return None
    with open(path) as f:
        prof = json.load(f)
    for k, v in prof.items():
        if kernel_key in k:
            return v["hbm_read_bytes"] + v["hbm_write_bytes"]
    return None


def cpu_baseline(seconds_budget: float = 12.0) -> dict:
    """The oracle's C restatement of the frozen graph (oracle/bp_oracle.c: fp32, AVX2, OpenMP over windows)
    on all host cores, bounded sample (checker code, timed as the CPU baseline).  Falls back to the torch-CPU
    oracle when the C library has not been built."""
    from oracle import bp_oracle as O

    rng = np.random.default_rng(0)
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:  # a cgroup CPU quota (the GPU box: 16 of 256 hardware threads) is the number of cores we really have
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            threads = max(1, min(threads, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    try:
        O.c_library()
        chunk = 2 * threads
        x = rng.uniform(-1, 1, (chunk, O.AUDIO_N_SAMPLES)).astype(np.float32)
        run = lambda: O.forward_c(x, threads)  # noqa: E731
        what = "C/OpenMP fp32 oracle (oracle/bp_oracle.c)"
    except OSError:
        import torch

        W = O.load_weights()
        threads = torch.get_num_threads()
        chunk = 8
        x = rng.uniform(-1, 1, (chunk, O.AUDIO_N_SAMPLES)).astype(np.float32)
        run = lambda: O.forward(x, W, np.float32)  # noqa: E731
        what = "torch-CPU fp32 oracle"
    run()  # warm-up (thread pool, page faults)
    done = 0
    t0 = time.perf_counter()
    while True:
        run()
        done += chunk
        el = time.perf_counter() - t0
        if el >= seconds_budget or done >= 65536:
            break
    return {
        "value": done / el,
        "unit": "windows/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} uniform[-1,1) windows in batches of {chunk}, {el:.1f} s, {what}",
    }


def run_tracks(args, torch, dist, world, rank, local_rank) -> None:
    """BASELINE.json configs[2]: N synthetic 3-minute tracks (180 s @ 22.05 kHz = 3,969,000 samples -> 110 windows,
    15,584 output frames each; lengths jittered by up to 2 % so the shard plan has something to balance), sharded by
    file over the ranks with the LPT plan of basic_pitch_amd/sharding.py, each through bp_infer_track with device
    input and output: windowing with the 3840-sample lead-in, CQT + CNN, un-overlapping.  One "step" = one pass over
    the rank's shard; no collective on the data path."""
    from basic_pitch_amd.inference import Model
    from basic_pitch_amd.sharding import plan_shards, shard_imbalance

    dev = torch.device("cuda", local_rank)
    rng = np.random.default_rng(2024)
    base = 180 * 22050
    lengths = [int(base * (1.0 - 0.02 * rng.random())) for _ in range(args.tracks)]
    shards = plan_shards(lengths, world)
    mine = shards[rank]
    model = Model(device=local_rank, max_windows=256)
    lib = model._lib
    g = torch.Generator(device=dev)
    g.manual_seed(99 + rank)
    pool = [(torch.rand((base,), generator=g, device=dev) * 2.0 - 1.0).contiguous() for _ in range(4)]
    n_windows = sum(int(lib.bp_track_n_windows(lengths[i])) for i in mine)

    group = 64  # tracks per bp_infer_tracks call (windows are packed across track boundaries into full batches)

    def step():
        for g0 in range(0, len(mine), group):
            ids = mine[g0 : g0 + group]
            model.predict_tracks([pool[j % len(pool)][: lengths[i]] for j, i in enumerate(ids)])

    for _ in range(args.warmup if args.warmup < 2 else 1):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    total = args.reduce_over_ranks(float(n_windows), dist.ReduceOp.SUM)
    elapsed = args.reduce_over_ranks(elapsed, dist.ReduceOp.MAX)
    if rank == 0:
        line = {
            "metric": "audio windows/sec (2 s @ 22.05 kHz) end-to-end CQT+CNN",
            "value": total * args.steps / elapsed,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": DTYPE_DEFAULT,
            "data": "synthetic",
            "config": {
                "workload": f"{args.tracks} synthetic 3-minute tracks (110 windows each) through bp_infer_tracks (64 tracks per call), "
                "device-resident in/out, file-sharded (BASELINE.json configs[2])",
                "tracks_per_s": args.tracks * args.steps / elapsed,
                "shard_imbalance": shard_imbalance(lengths, shards),
                "sharding": "LPT by sample count, no collective",
            },
        }
        print(json.dumps(line), flush=True)


def run_files(args, torch=None, dist=None, world: int = 1, rank: int = 0, local_rank: int = 0) -> None:
    """End-to-end `predict()` over FILES (not the headline metric: host decode, H2D of the PCM, device
    resampling, CQT + CNN, D2H of what note decoding needs and note decoding on host threads are all inside the timed
    region): `--files` synthetic 16-bit stereo 44.1 kHz WAV files of `--file-seconds` each in a temporary directory.
    Reports files/s and audio-seconds per second.  With `--gpus N --native` (BASELINE.json configs[2] as a FILE job) the
    files are sharded over the ranks by the product's own plan (`sharding.plan_shards`: LPT over the files' sizes; no collective on the data path),
    every rank runs the native pipeline on its own GPU with its share of the host cores, the timed region is bracketed by
    barriers and closed by the slowest rank; rank 0 reports the whole-job rate, the per-rank rates and their imbalance."""
    import tempfile
    import wave

    from basic_pitch_amd.inference import Model, predict_many

    if world > 1 and not args.native:
        raise SystemExit("--workload files on several GPUs runs the native pipeline: add --native")
    total_files = args.files
    from basic_pitch_amd.sharding import plan_shards

    # the product's plan (what predict_and_save_sharded computes from os.path.getsize): the synthetic files are equally long
    file_bytes = 44 + int(args.file_seconds * 44100) * 4
    mine = plan_shards([float(file_bytes)] * total_files, world)[rank]
    args.files = len(mine)
    rng = np.random.default_rng(7 + rank)
    n = int(args.file_seconds * 44100)
    t = np.arange(n) / 44100.0
    with tempfile.TemporaryDirectory() as d:
        paths = []
        distinct = min(args.files, 32)  # the rest are hard links to these (same bytes, another name): 0.4 s of numpy per file
        synth = None
        if args.flac:
            # a FLAC copy of the corpus: tools/flac_synth.c (LPC order 8 + Rice partitions, like a real encoder's output) is
            # compiled here; the product itself has no encoder
            import subprocess

            synth = os.path.join(d, "flac_synth")
            subprocess.run(["gcc", "-O2", "-o", synth, os.path.join(ROOT, "tools", "flac_synth.c"), "-lm"], check=True)
        for i in range(args.files):
            p = os.path.join(d, f"f{i}.wav")
            if i >= distinct:
                os.link(paths[i % distinct], p)
                paths.append(p)
                continue
            f0 = 110.0 * 2 ** (rng.integers(0, 36) / 12.0)
            x = 0.3 * np.sin(2 * np.pi * f0 * t) * (np.sin(2 * np.pi * 1.5 * t) > 0) + 0.01 * rng.standard_normal(n)
            pcm = (np.clip(np.stack([x, x[::-1]], 1), -1, 1) * 32767).astype("<i2")
            with wave.open(p, "wb") as w:
                w.setnchannels(2)
                w.setsampwidth(2)
                w.setframerate(44100)
                w.writeframes(pcm.tobytes())
            paths.append(p)
        if synth:
            import subprocess

            flac_paths = []
            for i, p in enumerate(paths):
                q = p[:-4] + ".flac"
                if i >= distinct:
                    os.link(flac_paths[i % distinct], q)
                else:
                    subprocess.run([synth, p, q], check=True, stderr=subprocess.DEVNULL)
                flac_paths.append(q)
            flac_bytes = os.path.getsize(flac_paths[0])
            for p in paths:
                os.unlink(p)
            paths = flac_paths
        model = Model(device=local_rank, max_windows=256)
        windows = sum(int(model._lib.bp_track_n_windows(int(np.ceil(n / 2)))) for _ in paths)
        per_rank = None
        if args.native:
            # the native pipeline: one bp_transcribe_files call, C++ worker threads from the file's bytes to its .mid + .csv
            from basic_pitch_amd import transcribe_files
            from basic_pitch_amd.sharding import usable_cpus

            model.close()
            threads = args.native_threads or (max(2, usable_cpus() // world) if world > 1 else 0)
            lanes = [Model(device=local_rank, max_windows=128, blocking_wait=True) for _ in range(args.lanes)]
            out_dir, warm_dir = os.path.join(d, "out"), os.path.join(d, "warm")
            os.mkdir(out_dir)
            os.mkdir(warm_dir)
            kw = dict(models=lanes, threads=threads, host_decode=args.host_decode, direct_io=args.direct_io, host_flac=args.host_flac)
            transcribe_files(paths[: min(8, len(paths))], warm_dir, **kw)  # warm-up
            if args.cold_files:
                # the corpus does not sit in the page cache: written back, then dropped from it (the distinct files; the rest
                # are hard links to them).  Buffered reads then come from the storage device THROUGH the page cache,
                # O_DIRECT reads from the device straight into the page-locked buffers.
                os.sync()
                for pth in paths[:distinct]:
                    fd = os.open(pth, os.O_RDONLY)
                    try:
                        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                    finally:
                        os.close(fd)
            direct_before = int(lanes[0]._lib.bp_files_direct_reads())
            import resource

            ru0 = resource.getrusage(resource.RUSAGE_SELF)
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            rep = transcribe_files(paths, out_dir, **kw)
            el = time.perf_counter() - t0
            if world > 1:
                dist.barrier()
                mine_t = torch.tensor([el, float(len(paths))], dtype=torch.float64)
                allt = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
                dist.all_gather(allt, mine_t)
                per_rank = [(float(a[0]), int(a[1])) for a in allt]
                el = max(e for e, _ in per_rank)
            bad = [r for r in rep if r["status"] != 0]
            if bad:
                raise SystemExit(f"native pipeline: {len(bad)} files failed: {bad[0]}")
            n_events = sum(r["n_note_events"] for r in rep)
            stage_ms = {k: float(np.mean([r["ms"][k] for r in rep])) for k in rep[0]["ms"]}
            ru1 = resource.getrusage(resource.RUSAGE_SELF)
            io_note = {"host_cpu_ms_per_file": {"user": (ru1.ru_utime - ru0.ru_utime) * 1e3 / max(1, len(paths)),
                                                "system": (ru1.ru_stime - ru0.ru_stime) * 1e3 / max(1, len(paths))},
                       "direct_io_requested": bool(args.direct_io),
                       "files_read_with_o_direct": int(lanes[0]._lib.bp_files_direct_reads()) - direct_before,
                       "cold_files": bool(args.cold_files), "distinct_files": distinct, "tmp_dir": d,
                       "lanes": args.lanes, "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")}
            if args.flac:
                io_note["flac"] = {"decoder": "host (bp_flac_decode)" if args.host_flac else "device (flac_device.hip)",
                                   "bytes_per_file": flac_bytes, "wav_bytes_per_file": file_bytes}
            for m in lanes:
                m.close()
            back = ("all three posteriorgrams back (27.6 MB per file), note decoding on the host" if args.host_decode else
                    "inferred onsets, peak picking and the pitch bends on the device, the note map + onset-peak bitmap + bend map "
                    "back (7.1 MB per file), the note tracker on the host")
            how = (f"bp_transcribe_files: {args.lanes} GPU lanes (handles) per GPU, "
                   f"{threads or 'one per usable core'} C++ worker threads per process, each file from its bytes to its "
                   ".mid + .csv without Python (file read into page-locked memory, the 16-bit samples over PCIe as stored, "
                   f"conversion + downmix + resampling on the device, CQT + CNN, {back}, MIDI / CSV encoding, file writes)")
        elif args.save_workers > 0:
            # the batch job: predict_and_save_sharded, `--save-workers` host processes on the one GPU, every worker writes
            # its own MIDI + note CSV (nothing but small reports crosses process boundaries)
            from basic_pitch_amd import predict_and_save_sharded

            model.close()
            # eight names per file (symlinks): a job long enough to amortise the workers' start-up (~2 s of imports each)
            real = list(paths)
            for k in range(1, 8):
                for i, src in enumerate(real):
                    link = os.path.join(d, f"f{i}_{k}.wav")
                    os.symlink(src, link)
                    paths.append(link)
            windows *= 8
            out_dir, warm_dir = os.path.join(d, "out"), os.path.join(d, "warm")
            os.mkdir(out_dir)
            os.mkdir(warm_dir)  # an existing output file is an IOError, as in the reference (inference.py:401-404)
            kw = dict(gpus=1, workers_per_gpu=args.save_workers, group=32,
                      decode_threads=max(2, (os.cpu_count() or 2) // args.save_workers))
            predict_and_save_sharded(paths[: args.save_workers * 2], warm_dir, True, False, False, True, **kw)  # warm-up
            t0 = time.perf_counter()
            rep = predict_and_save_sharded(paths, out_dir, True, False, False, True, **kw)
            el = time.perf_counter() - t0
            for r in rep:
                if isinstance(r, BaseException):
                    raise r
            n_events = sum(r["n_note_events"] for r in rep)
            how = (f"predict_and_save_sharded, {args.save_workers} worker processes on the GPU, each writing its own .mid + "
                   ".csv (process start-up and model load inside the timed region)")
        else:
            predict_many(paths[: min(4, len(paths))], model)  # warm-up
            t0 = time.perf_counter()
            res = predict_many(paths, model, group=32)
            el = time.perf_counter() - t0
            n_events = sum(len(r[2]) for r in res)
            how = ("predict_many (host WAV read on a thread pool, PCM over PCIe, device resampling, note decoding on host "
                   "threads)")
    if rank != 0:
        return
    n_all = total_files if per_rank else len(paths)
    extra = {}
    if per_rank:
        rates = [k / e for e, k in per_rank]
        extra = {"scaling": "strong", "per_rank_files_per_s": rates, "per_rank_elapsed_s": [e for e, _ in per_rank],
                 "imbalance_max_over_min_elapsed": max(e for e, _ in per_rank) / min(e for e, _ in per_rank)}
        windows = windows * n_all / max(1, len(paths))
    print(json.dumps({
        "metric": f"files/sec end-to-end predict() (decode + resample + CQT + CNN + note decoding), {world} MI355X",
        "value": n_all / el, "unit": "files/s", "n_gpus": world, "higher_is_better": True, "data": "synthetic",
        "audio_seconds_per_s": n_all * args.file_seconds / el, "windows_per_s": windows / el,
        "config": {"workload": f"{n_all} synthetic 16-bit stereo 44.1 kHz {'FLAC' if args.flac else 'WAV'} files ({min(len(paths), 32)} distinct signals per rank) of {args.file_seconds:g} s through "
                   + how, "host_threads": min(16, os.cpu_count() or 1), "note_events": n_events,
                   "sharding": "sharding.plan_shards (LPT over the files' sizes), no collective on the data path"},
        **extra,
        **({"worker_ms_per_file": stage_ms, "file_io": io_note} if args.native else {}),
    }), flush=True)

STEP_ALGORITHMIC_BYTES_PER_WINDOW = BYTES_PER_WINDOW  # fp32 audio in + three fp32 posteriorgrams out


def pmc_step_traffic(batch: int):
    """Whole-step HBM bytes (every kernel of one step: PMC read x 2 + written, the committed profile) and their ratio to
    the step's algorithmic bytes (audio in + posteriorgrams out): what the implementation moves beyond what the path
    has to.  None when no profile for this batch is committed."""
    path = os.path.join(ROOT, "profiles", PMC_PROFILE + "_pmc.json")
    if batch != 256 or not os.path.exists(path):
        return None
    with open(path) as f:
        prof = json.load(f)
    total = sum((v.get("hbm_read_bytes", 0.0) + v.get("hbm_write_bytes", 0.0)) * v.get("launches_per_step", 1)
                for v in prof.values() if isinstance(v, dict))
    alg = STEP_ALGORITHMIC_BYTES_PER_WINDOW * batch
    return {"bytes_per_step": total, "algorithmic_bytes_per_step": alg, "ratio": total / alg,
            "source": "profiles/%s_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)" % PMC_PROFILE}


def config_extras(torch, local_rank: int, steps: int = 12) -> dict:
    """The other BASELINE.json configs and the reference's own call pattern, AFTER the timed region (rank 0 only, the GPU
    still warm from it): beside `value`, never instead of it.  Every entry: 3 warm-up steps, then >= 10 timed steps
    (round 5 timed 4 steps behind 2: B = 1024 read slower than B = 256 there); windows/s and ms per step (per call)."""
    from basic_pitch_amd.inference import Model

    dev = torch.device("cuda", local_rank)
    out = {}

    def windows_mode(key, B, note, **kw):
        g = torch.Generator(device=dev)
        g.manual_seed(4321)
        win = 87688 if kw.get("ext_cqt_44k") else 43844
        audio = (torch.rand((B, win), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()
        o = {"note": torch.empty((B, 172, 88), device=dev), "onset": torch.empty((B, 172, 88), device=dev),
             "contour": torch.empty((B, 172, 264), device=dev)}
        m = Model(device=local_rank, max_windows=B, **kw)
        for _ in range(3):
            m._predict_device(audio, out=o, sync=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            m._predict_device(audio, out=o, sync=False)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        m.close()
        out[key] = {"windows_per_s": B * steps / el, "ms_per_step": el / steps * 1e3, "batch": B, "steps": steps, "warmup": 3,
                    "config": note}
        del audio, o

    windows_mode("b1024", 1024, "fp32, batch 1024 (configs[1] at 4 x the batch)")
    windows_mode("bf16_b1024", 1024, "bf16 CNN weights + fp32 CQT, batch 1024 (configs[3])", bf16_weights=True)
    windows_mode("ext44k_b512", 512, "44.1 kHz windows, extended 345-bin CQT, batch 512 (configs[4]; one GPU's share)",
                 ext_cqt_44k=True)
    windows_mode("b16", 16, "fp32, batch 16 (small-batch latency)")

    # configs[2] on one GPU: 256 synthetic 3-minute tracks (110 windows each), 64 per bp_infer_tracks call, device in/out
    m = Model(device=local_rank, max_windows=256)
    n_samples = 180 * 22050
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    pool = [(torch.rand((n_samples,), generator=g, device=dev) * 2.0 - 1.0).contiguous() for _ in range(4)]
    n_win = int(m._lib.bp_track_n_windows(n_samples))
    m.predict_tracks([pool[j % 4] for j in range(16)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4):
        m.predict_tracks([pool[j % 4] for j in range(64)])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out["tracks_256x3min"] = {"windows_per_s": 256 * n_win / el, "tracks_per_s": 256 / el, "windows_per_track": n_win,
                              "config": "256 synthetic 3-minute tracks through bp_infer_tracks, 64 per call, device "
                                        "in/out (configs[2], one GPU's share)"}
    m.close()
    del pool
    return out


def seam_b1_host(calls: int = 30) -> dict:
    """The reference's real call pattern (basic_pitch/inference.py:308-310, 173-180): one host window per
    `session.run([...], {input: x[n:n+1]})` through the onnxruntime-shaped seam — H2D, all kernels at batch 1, D2H, three
    fresh numpy arrays per call."""
    from basic_pitch_amd import ort_shim
    from basic_pitch_amd.inference import ICASSP_2022_MODEL_PATH

    sess = ort_shim.InferenceSession(str(ICASSP_2022_MODEL_PATH), max_windows=1)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, (1, 43844, 1)).astype(np.float32)
    names = ["StatefulPartitionedCall:1", "StatefulPartitionedCall:2", "StatefulPartitionedCall:0"]
    for _ in range(3):
        sess.run(names, {ort_shim.INPUT_NAME: x})
    t0 = time.perf_counter()
    for _ in range(calls):
        sess.run(names, {ort_shim.INPUT_NAME: x})
    el = time.perf_counter() - t0
    return {"windows_per_s": calls / el, "ms_per_call": el / calls * 1e3, "calls": calls,
            "config": "ort_shim.InferenceSession.run on ONE host window per call (the reference's batch-1 loop, "
                      "inference.py:308-310): PCIe in, batch-1 kernels, PCIe out"}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sustained-s", type=float, default=2.0, help="length of the extra steady-state measurement (0 = skip)")
    ap.add_argument("--no-exact-f32", action="store_true", help="skip the extra exact-f32 A/B rate (profiling runs)")
    ap.add_argument("--exact-f32", action="store_true", help="contour conv1 on the exact-f32 MFMA kernel (A/B)")
    ap.add_argument("--files", type=int, default=64, help="files in the job (--workload files)")
    ap.add_argument("--file-seconds", type=float, default=180.0, help="length of each file (--workload files)")
    ap.add_argument("--native", action="store_true",
                    help="--workload files: the native pipeline (bp_transcribe_files: C++ worker threads, no Python in the loop)")
    ap.add_argument("--lanes", type=int, default=3, help="--native: GPU lanes (handles) the workers queue for")
    ap.add_argument("--flac", action="store_true",
                    help="--workload files --native: the corpus as FLAC (encoded here by tools/flac_synth.c), decoded on the device")
    ap.add_argument("--host-flac", action="store_true", help="--flac: decode on the host (bp_flac_decode) instead")
    ap.add_argument("--direct-io", action="store_true",
                    help="--native: read the files with O_DIRECT straight into the page-locked buffers (no page-cache copy)")
    ap.add_argument("--cold-files", action="store_true",
                    help="--native: drop the files from the page cache before the timed job (a corpus larger than host memory)")
    ap.add_argument("--host-decode", action="store_true",
                    help="--native: bring all three posteriorgrams back and decode on the host (the round-4 path) instead of "
                         "extracting the onset peaks and pitch bends on the device")
    ap.add_argument("--native-threads", type=int, default=0, help="--native: C++ worker threads (0: one per hardware thread)")
    ap.add_argument("--save-workers", type=int, default=0,
                    help="--workload files: run the job as predict_and_save_sharded with this many host processes on the GPU")
    ap.add_argument("--workload", choices=["windows", "tracks", "files"], default="windows",
                    help="windows: BASELINE.json configs[1] (the headline line); tracks: configs[2], whole synthetic "
                    "3-minute tracks through bp_infer_track, file-sharded over the ranks")
    ap.add_argument("--tracks", type=int, default=1000, help="tracks in the whole job (--workload tracks)")
    ap.add_argument("--ext-cqt-44k", action="store_true",
                    help="BASELINE.json configs[4]: 44.1 kHz windows (87,688 samples), 10-octave / 345-bin CQT (use with "
                    "--batch 512); not the headline line")
    ap.add_argument("--f16-corrections", action="store_true",
                    help="accepted for compatibility: all three split-precision products on f16 MFMA is the default since round 3")
    ap.add_argument("--no-fp8-extra", action="store_true",
                    help="accepted and ignored (the fp8-corrections mode left the product library in round 6; old tool scripts pass it)")
    ap.add_argument("--no-config-extras", action="store_true",
                    help="skip the side rates of the other BASELINE.json configs and of the batch-1 host seam (profiling runs)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="all ranks use device 0 (exercising the N > 1 path on a one-GPU box; the numbers mean nothing)")
    ap.add_argument("--control-backend", choices=["gloo", "nccl"], default="gloo",
                    help="process group of the barrier / max-over-ranks reduction around the timed region: the data path "
                         "has no collective, so the control path runs on CPU tensors over gloo by default")
    ap.add_argument("--bf16-weights", action="store_true",
                    help="BASELINE.json configs[3]: bf16 CNN weights + fp32 CQT (use with --batch 1024); not the headline line")
    args = ap.parse_args()

    if args.workload == "files" and args.native and args.lanes > 4:
        # the HIP runtime maps a process's streams onto FOUR hardware queues by default; kernels of lanes that share a queue
        # run one after the other (a FLAC file's 2.7 ms decode kernel holds its queue that long: profiles/r06_flac_device.md).
        # One queue per lane, set before the runtime initialises; reported on the line.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(args.lanes, 16)))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, RCCL rendezvous on
        # 127.0.0.1) and relay rank 0's JSON line
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # Control path only (barrier + max over ranks around the timed region): windows are independent units and the
        # data path has no collective, so the default control group is gloo on CPU tensors; nccl (= RCCL)
        # is kept behind --control-backend nccl.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.control_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to 1 peer ranks ..."): the contract is
            # ONE JSON line there, so the group is brought up (first collective included) with fd 1 pointing at stderr
            sys.stdout.flush()
            saved_fd = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo")
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved_fd, 1)
                os.close(saved_fd)

    from basic_pitch_amd.inference import Model

    ctl_dev = torch.device("cuda", local_rank) if args.control_backend == "nccl" else torch.device("cpu")

    def reduce_over_ranks(x: float, op) -> float:
        if world == 1:
            return x
        t = torch.tensor([x], device=ctl_dev, dtype=torch.float64)
        dist.all_reduce(t, op=op)
        return float(t.item())

    args.reduce_over_ranks = reduce_over_ranks
    if args.workload == "files":
        run_files(args, torch, dist, world, rank, local_rank)
        if world > 1:
            dist.destroy_process_group()
        return
    if args.workload == "tracks":
        run_tracks(args, torch, dist, world, rank, local_rank)
        if world > 1:
            dist.destroy_process_group()
        return

    B = args.batch
    dev = torch.device("cuda", local_rank)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    win = 87688 if args.ext_cqt_44k else 43844
    audio = (torch.rand((B, win), generator=g, device=dev, dtype=torch.float32) * 2.0 - 1.0).contiguous()
    out = {
        "note": torch.empty((B, 172, 88), device=dev),
        "onset": torch.empty((B, 172, 88), device=dev),
        "contour": torch.empty((B, 172, 264), device=dev),
    }
    # The timed steps carry HIP events around the dominant kernel only (roofline.achieved is measured live there); the
    # full per-stage table costs 16 event records = ~25 us (2.6 %) per step and comes from a second, untimed pass.
    # The exact-f32 A/B path has no dominant-only mode: it is timed with the full set.
    model = Model(device=local_rank, max_windows=B, stage_timing=args.exact_f32, time_dominant=not args.exact_f32,
                  exact_f32_mfma=args.exact_f32, bf16_weights=args.bf16_weights, ext_cqt_44k=args.ext_cqt_44k)

    def step():
        model._predict_device(audio, out=out, sync=False)

    # Extra key beside the contract's K-step number: `sustained` — the same step back to back for >= 2 s on a handle
    # without event records, all ranks together.  It runs FIRST: K = 20..30 steps (~25 ms) starting on an idle GPU end
    # before the clock governor has settled (rounds 3 - 4 reported 0.786 ms per step in the K steps against 0.711
    # sustained), so the W warm-up + K timed steps below start right behind two seconds of the same work, on a GPU in the
    # state a batch job keeps it in.
    extras = {}
    # `value_cold`: the contract's W warm-up + K timed steps started on an idle GPU, as rounds 1 - 4 measured `value` — run
    # FIRST, on a handle of its own, so that both protocols are on every line and a round-to-round difference can be told
    # from a protocol difference.  `value` itself is measured behind the sustained pass since round 5.
    if not args.exact_f32 and args.sustained_s > 0:
        cold_model = Model(device=local_rank, max_windows=B, bf16_weights=args.bf16_weights, ext_cqt_44k=args.ext_cqt_44k)
        for _ in range(args.warmup):
            cold_model._predict_device(audio, out=out, sync=False)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cold_model._predict_device(audio, out=out, sync=False)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t_cold = reduce_over_ranks(time.perf_counter() - t0, dist.ReduceOp.MAX)
        cold_model.close()
        extras["value_cold"] = B * args.steps * world / t_cold
        extras["ms_per_step_cold"] = t_cold / args.steps * 1e3
        extras["value_note"] = ("`value` / `ms_per_step`: W warm-up + K timed steps run directly behind `sustained` (>= 2 s of the "
                                "same step; since round 5).  `value_cold` / `ms_per_step_cold`: the same W + K steps started on "
                                "an idle GPU before anything else ran (the protocol of rounds 1 - 4)")
    sus_model = None
    if not args.exact_f32 and args.sustained_s > 0:
        sus_model = Model(device=local_rank, max_windows=B, bf16_weights=args.bf16_weights, ext_cqt_44k=args.ext_cqt_44k)

        def sus_step():
            sus_model._predict_device(audio, out=out, sync=False)

        for _ in range(3):
            sus_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            sus_step()
        torch.cuda.synchronize()
        probe = (time.perf_counter() - t0) / 10
        n_sus = max(args.steps, int(args.sustained_s / probe) + 1)
        if world > 1:
            n_sus = int(reduce_over_ranks(float(n_sus), dist.ReduceOp.MAX))
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            sus_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t_sus = time.perf_counter() - t0
        t_sus = reduce_over_ranks(t_sus, dist.ReduceOp.MAX)
        extras["sustained"] = {"windows_per_s": B * n_sus * world / t_sus, "steps": n_sus, "seconds": t_sus,
                               "ms_per_step": t_sus / n_sus * 1e3,
                               "note": "same step, back to back, no event records, run directly before the W warm-up + K "
                                       "timed steps; beside `value`, not instead of it"}

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if args.warmup:
        model.stage_ms()  # reset the per-stage accumulation

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stage = model.stage_ms()  # mean per-launch ms over the timed steps (HIP events on the kernels' stream)
    if sus_model is not None:
        sus_model.close()  # after the timed steps: freeing a gigabyte of device buffers idles the GPU for milliseconds
    if not args.exact_f32 and rank == 0:
        # untimed second pass with events around every stage (same inputs, same kernels); the dominant kernel's entry
        # stays the one measured inside the timed region
        dom = {k: v for k, v in stage.items() if v > 0.0}
        model.close()
        model = Model(device=local_rank, max_windows=B, stage_timing=True, bf16_weights=args.bf16_weights,
                      ext_cqt_44k=args.ext_cqt_44k)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        model.stage_ms()
        for _ in range(min(args.steps, 20)):
            step()
        torch.cuda.synchronize()
        stage = model.stage_ms()
        stage_all_pass = dict(stage)
        stage.update(dom)
    else:
        stage_all_pass = None

    elapsed = reduce_over_ranks(elapsed, dist.ReduceOp.MAX)

    ok = bool(torch.isfinite(out["note"]).all() and torch.isfinite(out["onset"]).all() and torch.isfinite(out["contour"]).all())

    # (2) rank 0 only, the exact-f32 A/B path's rate on the same batch.
    if not args.exact_f32 and not args.no_exact_f32 and rank == 0 and not (args.bf16_weights or args.ext_cqt_44k):
        ex_model = Model(device=local_rank, max_windows=B, exact_f32_mfma=True)
        for _ in range(2):
            ex_model._predict_device(audio, out=out, sync=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(8):
            ex_model._predict_device(audio, out=out, sync=False)
        torch.cuda.synchronize()
        extras["exact_f32_windows_per_s"] = B * 8 / (time.perf_counter() - t0)
        ex_model.close()
    if (not args.exact_f32 and not args.no_config_extras and rank == 0 and world == 1
            and not (args.bf16_weights or args.ext_cqt_44k)):
        # beside the headline: two handles on two streams taking alternate batches — what a service with two lanes gets
        # (the kernels of one batch run under the launch gaps and tails of the other's; per-kernel times are no longer
        # those of a kernel alone on the chip, which is why `value` and `roofline` stay on one stream)
        lanes = [Model(device=local_rank, max_windows=B) for _ in range(2)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
        outs = [out, {k: torch.empty_like(v) for k, v in out.items()}]

        def lane_steps(n):
            for i in range(n):
                with torch.cuda.stream(streams[i & 1]):
                    lanes[i & 1]._predict_device(audio, out=outs[i & 1], sync=False)

        lane_steps(6)
        torch.cuda.synchronize()
        n_two = max(2 * args.steps, 200)
        t0 = time.perf_counter()
        lane_steps(n_two)
        torch.cuda.synchronize()
        t_two = time.perf_counter() - t0
        extras["two_lanes"] = {"windows_per_s": B * n_two / t_two, "ms_per_step": t_two / n_two * 1e3, "steps": n_two,
                               "note": "two handles, two streams, alternate batches of the same 256 windows; beside `value`"}
        for m2 in lanes:
            m2.close()

    if rank == 0:
        total_windows = B * args.steps * world
        value = total_windows / elapsed
        if args.exact_f32:
            c1_ms = stage["contour1"]
            c1_flop = C1_FLOP_PER_WINDOW
            c1_kernel = "contour1_kernel (Conv2D 8->8 3x39 + norm/BN/stack; exact-f32 MFMA 32x32x2)"
            c1_peak = F32_MFMA_PEAK_TFLOPS
            c1_exec = 355 * 504 * (2 * 32 * 32 * 2) * B / (c1_ms * 1e-3) / 1e12
            c1_bytes = None
            c1_key = None
        else:
            c1_ms = stage["contour_conv1"]
            folded = stage.get("contour_conv1_edge", 0.0) > 0.0
            mf = (2 / 3 if args.bf16_weights else 1)
            if folded and os.environ.get("BP_CONV1") != "rounds":
                c1_flop = C1_FLOP_PER_WINDOW * F1_SHARE
                c1_kernel = ("contour_conv1_march_kernel (interior 56/66 of harmonic stack + Conv2D 8->8 3x39 + ReLU with the 8 "
                             "shifted channels folded into one 176-tap kernel; wave-private vertical march on f16 MFMA 16x16x32: "
                             "hi/lo-split weights resident in registers, each z fragment read once from LDS for all three frame "
                             "taps, fp32 accumulate; algorithmic FLOPs = the reference's 8-channel products it replaces)")
                c1_exec = M1_EXECUTED_FLOP_PER_WINDOW * mf * B / (c1_ms * 1e-3) / 1e12
                c1_bytes = F1_BYTES_PER_WINDOW * B
                c1_key = "contour_conv1_march_kernel"
            elif folded:
                c1_flop = C1_FLOP_PER_WINDOW * F1_SHARE
                c1_kernel = ("contour_conv1_folded_kernel (interior 56/66 of harmonic stack + Conv2D 8->8 3x39 + ReLU with the 8 "
                             "shifted channels folded into one 176-tap kernel; f16 MFMA 32x32x16 on hi/lo-split operands from "
                             "LDS, fp32 accumulate; algorithmic FLOPs = the reference's 8-channel products it replaces)")
                c1_exec = F1_EXECUTED_FLOP_PER_WINDOW * mf * B / (c1_ms * 1e-3) / 1e12
                c1_bytes = F1_BYTES_PER_WINDOW * B
                c1_key = "contour_conv1_folded_kernel"
            else:
                c1_flop = C1_FLOP_PER_WINDOW
                c1_kernel = ("contour_conv1_kernel<FullGeo> (harmonic stack + Conv2D 8->8 3x39 + ReLU; f16 MFMA 32x32x16 on "
                             "hi/lo-split operands from LDS, fp32 accumulate, no K split)")
                c1_exec = D1_EXECUTED_FLOP_PER_WINDOW * mf * B / (c1_ms * 1e-3) / 1e12
                c1_bytes = D1_BYTES_PER_WINDOW * B
                c1_key = "contour_conv1_kernel"
            c1_peak = F16_MFMA_PEAK_TFLOPS
        achieved = c1_flop * B / (c1_ms * 1e-3) / 1e12
        line = {
            "metric": "audio windows/sec (2 s @ 44.1 kHz) end-to-end CQT+CNN" if args.ext_cqt_44k
            else "audio windows/sec (2 s @ 22.05 kHz) end-to-end CQT+CNN",
            "value": value,
            "unit": "windows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # fp32 data and accumulation; matrix products on f16 hi+lo operand pairs (22 bits); --bf16-weights: conv
            # weights rounded to bf16 (one f16 operand), activations and CQT unchanged
            "dtype": ("f32 I/O + accumulate, bf16-rounded weights as single f16 MFMA operands, split-f16 activations" if args.bf16_weights
                      else "exact f32 MFMA (A/B path)" if args.exact_f32
                      else DTYPE_DEFAULT),
            "data": "synthetic",
            "config": {
                "workload": f"batch={B} synthetic uniform[-1,1) 2 s @ 22.05 kHz mono windows per GPU, "
                + ("44.1 kHz windows of 87,688 samples, extended 10-octave / 345-bin CQT, HBM-resident in/out "
                   "(BASELINE.json configs[4])" if args.ext_cqt_44k else
                   "bf16 CNN weights + fp32 CQT, HBM-resident in/out (BASELINE.json configs[3])" if args.bf16_weights
                   else "fp32, HBM-resident in/out (BASELINE.json configs[1])"),
                "windows_per_step_per_gpu": B,
                "sharding": "independent windows per rank, no collective",
                **({"share_gpu": "all ranks on device 0 (functional check of the N > 1 path, not a scaling number)"}
                   if args.share_gpu else {}),
            },
            "roofline": {
                "kernel": c1_kernel,
                "bound": "mfma",
                "achieved": achieved,
                "peak": c1_peak,
                "unit": "TFLOP/s",
                "frac": achieved / c1_peak,
                "traffic": pmc_traffic(c1_key, B) if c1_key else None,
                "traffic_unit": "bytes per launch (PMC, profiles/%s_pmc.md)" % PMC_PROFILE,
                # the kernel's own ratio above is ~1; the STEP moves far more than the path's algorithmic I/O (c1 round trip,
                # pyramid planes, lp): that ratio belongs on the line too
                "step_traffic": pmc_step_traffic(B) if c1_key else None,
                # what THIS kernel reads and writes by construction (zp in, the materialised 8-channel c1 out): an
                # implementation choice, not the algorithm's I/O ...
                "kernel_io_bytes_per_launch": c1_bytes,
                # ... which for the whole contour branch (conv1 + conv2; c1 never needs to exist) is the 172 x 309 fp32 CQT
                # in and the 172 x 264 fp32 contour map out
                "branch_algorithmic_bytes_per_launch": (172 * 309 * 4 + 172 * 264 * 4) * B,
                "executed_mfma_tflops": c1_exec,  # incl. the 3-product split and Toeplitz padding
                "executed_frac": c1_exec / c1_peak,
                "launch_ms": c1_ms,
                "algorithmic_flop_per_launch": c1_flop * B,
            },
            "path_roofline": {
                "flop_frac_f16_peak": FLOP_PER_WINDOW * value / world / (F16_MFMA_PEAK_TFLOPS * 1e12),
                "hbm_frac_algorithmic": BYTES_PER_WINDOW * value / world / (HBM_PEAK_GBS * 1e9),
            },
            "stage_ms": stage,
            "stage_ms_note": "per-stage HIP-event means from a second untimed pass (16 event records per step cost ~2.6 %); "
                             "the dominant kernel's entry is the one measured inside the timed steps"
                             + ("" if stage_all_pass is None or c1_key is None else
                                f" (second pass: {stage_all_pass.get('contour_conv1', stage_all_pass.get('contour', 0.0)):.4f} ms)"),
            "outputs_finite": ok,
        }
        if not args.ext_cqt_44k:
            # the CQT stage (a7-a9: pyramid + filterbank + NormalizedLog; since round 3 the filterbank
            # launch also normalises, BatchNorms and splits the windows it owns — `zpack` is a stage of its own only when a
            # launch has fewer windows than CUs) against both of its rooflines; algorithmic 79,425,024 FLOP and 387,968 B
            # (audio in + 172x309 fp32 out) per window
            cqt_ms = stage["pyramid"] + stage["filterbank"] + stage.get("zpack", 0.0)
            cqt_rate = B / (cqt_ms * 1e-3)
            # ceilings per window: HBM 387,968 B at 8 TB/s; split-f16 MFMA = 3 products x the dense matrices the two
            # kernels execute (decimators 22.4 MFLOP, filterbank 57.1 MFLOP clipped to the kernels' support) at the
            # 2.5 PFLOP/s f16 peak.  The matrix ceiling is the lower (binding) one.
            cqt_mfma_flop = 3 * (22_361_088 + 57_063_936)
            line["cqt_stage"] = {
                "ms": cqt_ms,
                "includes": "pyramid (planes) + filterbank + normalise / BatchNorm / split (zpack, fused into the filterbank "
                            "launch at this batch)",
                "windows_per_s": cqt_rate,
                "hbm_fraction": 387_968 * cqt_rate / (HBM_PEAK_GBS * 1e9),
                "mfma_split_f16_fraction": cqt_mfma_flop * cqt_rate / (F16_MFMA_PEAK_TFLOPS * 1e12),
                "binding": "split-f16 MFMA (3 products per multiply): its ceiling, 10.5 M windows/s, lies below the HBM "
                           "ceiling of 20.6 M windows/s",
                "algorithmic_bytes_per_window": 387_968,
                "algorithmic_flop_per_window": 79_425_024,
            }
        line.update(extras)
        if (not args.no_config_extras and not args.exact_f32
                and not (args.bf16_weights or args.ext_cqt_44k) and B == BATCH):
            model.close()
            del audio, out
            torch.cuda.empty_cache()
            line["configs"] = config_extras(torch, local_rank)
            line["seam_b1_host"] = seam_b1_host()
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()  # rank 0's host cores, after the timed region, for any N
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
