"""File-level sharding of transcription jobs across the GPUs of one node.

The hot path has no exchange step: every 2-second window is an independent unit (per-window
normalisation, basic_pitch/layers/signal.py:177-183; stitching is a concat/trim,
basic_pitch/inference.py:267-279).  So N GPUs = N processes, each with its own handle, each owning a
disjoint set of files; the only "communication" is handing per-file results (or their paths) back to
rank 0 on the host.  No RCCL collective touches the data path.

`plan_shards` is the longest-processing-time-first assignment by sample count; `run_sharded` is the generic
per-rank driver (it only uses object gathers on the host); `predict_many_sharded` is the product entry point:
`predict()` over a list of files on all GPUs of the node — the per-file loop of the reference's `predict_and_save`
(inference.py:548-604) spread by file.  It runs either inside an existing `torch.distributed` job (one rank per GPU,
`python -m torch.distributed.run ...`; results gathered on rank 0) or, from a plain single process, spawns one worker
process per GPU itself.
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple


def usable_cpus() -> int:
    """Cores this process may really use: the affinity mask, capped by a cgroup CPU quota (a container on a 256-thread
    host may own 16) — what csrc/file_pipeline.cpp::default_threads counts for the native pipeline's workers."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def plan_shards(costs: Sequence[float], world_size: int) -> List[List[int]]:
    """Greedy LPT: heaviest item first onto the least-loaded rank.  Deterministic on every rank."""
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    order = sorted(range(len(costs)), key=lambda i: (-float(costs[i]), i))
    loads = [0.0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += float(costs[i])
    for s in shards:
        s.sort()
    return shards


def shard_imbalance(costs: Sequence[float], shards: List[List[int]]) -> float:
    """max load / mean load (1.0 = perfect)."""
    loads = [sum(float(costs[i]) for i in s) for s in shards]
    mean = sum(loads) / max(1, len(loads))
    return max(loads) / mean if mean > 0 else 1.0


def split_windows(n_windows: int, world_size: int) -> List[Tuple[int, int]]:
    """Window-range fallback for ONE very long file: contiguous [start, stop) per rank.

    Valid because a window's samples are determined by its index alone (start = w*36164 - 3840,
    inference.py:207,242): no rank needs a neighbour's data.
    """
    base, rem = divmod(n_windows, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < rem else 0)
        out.append((start, start + n))
        start += n
    return out


def plan_units(costs: Sequence[float], world_size: int) -> Tuple[List[Tuple[int, int, int]], List[List[int]]]:
    """The plan of a file-sharded job with the window-range fallback: work units `(file, piece, n_pieces)`
    and, per rank, the indices of its units (LPT over the units' costs; deterministic on every rank).

    A file is one unit — unless it alone outweighs an even share of the job (cost > total / world_size): no assignment of
    whole files can balance it, so it is cut into n_pieces = min(world_size, ceil(cost / share)) contiguous window ranges of
    equal cost (`split_windows`; a window needs nothing from its neighbours).  With world_size 1, or when no file is that
    heavy, the units are the files and the plan is `plan_shards`'s."""
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    total = sum(float(c) for c in costs)
    share = total / world_size
    units: List[Tuple[int, int, int]] = []
    unit_costs: List[float] = []
    for i, c in enumerate(costs):
        c = float(c)
        k = 1
        if world_size > 1 and share > 0 and c > share:
            k = min(world_size, int(-(-c // share)))
        for j in range(k):
            units.append((i, j, k))
            unit_costs.append(c / k)
    return units, plan_shards(unit_costs, world_size)


def run_sharded(
    items: Sequence[Any],
    costs: Sequence[float],
    process: Callable[[Any], Any],
    rank: Optional[int] = None,
    world_size: Optional[int] = None,
    gather: bool = True,
) -> Optional[Dict[int, Any]]:
    """Process this rank's shard with `process(item)`; gather {index: result} on rank 0.

    Uses torch.distributed only for the host-side object gather (no device collective).  A failure
    on one item is isolated (recorded as the exception) like the per-file try/except of the
    reference's predict_and_save (inference.py:548-604).
    """
    dist = None
    if world_size is None or rank is None:
        import torch.distributed as dist_mod

        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
            rank, world_size = dist.get_rank(), dist.get_world_size()
        else:
            rank, world_size = 0, 1
    else:
        import torch.distributed as dist_mod

        if world_size > 1 and dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
    shards = plan_shards(costs, world_size)
    mine: Dict[int, Any] = {}
    for idx in shards[rank]:
        try:
            mine[idx] = process(items[idx])
        except Exception as e:  # per-item isolation
            mine[idx] = e
    if not gather:
        return mine
    if dist is None or world_size == 1:
        return mine
    bucket: Optional[List[Optional[Dict[int, Any]]]] = [None] * world_size if rank == 0 else None
    dist.gather_object(mine, bucket, dst=0)
    if rank != 0:
        return None
    merged: Dict[int, Any] = {}
    for part in bucket or []:
        merged.update(part or {})
    return merged


def _file_costs(paths: Sequence[Any]) -> List[float]:
    """Work estimate per file without decoding it: its size in bytes (0 for a missing file, which then fails in
    `predict_many` on the rank that owns it and is reported per file)."""
    out = []
    for p in paths:
        try:
            out.append(float(os.path.getsize(p)))
        except OSError:
            out.append(0.0)
    return out


def _predict_shard(paths: Sequence[Any], indices: Sequence[int], device: int, model_or_model_path: Any,
                   model_factory: Optional[Callable[[int], Any]], kwargs: Dict[str, Any],
                   pieces: Sequence[Tuple[int, int, int]] = ()) -> Dict[Any, Any]:
    """One rank's share: a Model on its own GPU, `predict_many` over its whole files -> {input index: result or
    exception}, and `predict_window_range` for its pieces of split files -> {(index, piece, n_pieces): rows or exception}."""
    from . import inference

    if not indices and not pieces:
        return {}
    if model_factory is not None:
        model = model_factory(device)
    elif isinstance(model_or_model_path, inference.Model):
        model = model_or_model_path
    else:
        model = inference.Model(model_or_model_path, device=device)
    out: Dict[Any, Any] = {}
    if indices:
        res = inference.predict_many([paths[i] for i in indices], model, return_exceptions=True, **kwargs)
        out.update(zip(indices, res))
    for i, j, k in pieces:
        try:
            out[(i, j, k)] = inference.predict_window_range(paths[i], model, j, k)
        except Exception as e:  # per-file isolation, like the whole files'
            out[(i, j, k)] = e
    return out


def _units_of_rank(paths: Sequence[Any], world: int, rank: int) -> Tuple[List[int], List[Tuple[int, int, int]]]:
    units, shards = plan_units(_file_costs(paths), world)
    mine = [units[u] for u in shards[rank]]
    return [i for i, j, k in mine if k == 1], [(i, j, k) for i, j, k in mine if k > 1]


def _merge_units(merged: Dict[Any, Any], n_files: int, kwargs: Dict[str, Any]) -> List[Any]:
    """Per-file results in input order: whole files as they are, split files assembled from their pieces (the first piece's
    exception stands for the file: a file that cannot be read fails on every rank that touched it)."""
    from . import inference

    split: Dict[int, List[Any]] = {}
    for key, val in merged.items():
        if isinstance(key, tuple):
            split.setdefault(key[0], []).append((key[1], val))
    decode_kw = {k: v for k, v in kwargs.items() if k not in ("group", "decode_threads", "return_exceptions")}
    out: List[Any] = []
    for i in range(n_files):
        if i in split:
            parts = [v for _, v in sorted(split[i], key=lambda t: t[0])]
            bad = [v for v in parts if isinstance(v, BaseException)]
            if bad:
                out.append(bad[0])
                continue
            try:
                out.append(inference.assemble_window_ranges(parts, **decode_kw))
            except Exception as e:
                out.append(e)
        else:
            out.append(merged[i])
    return out


def _post(queue, rank: int, produce: Callable[[], Any]) -> None:
    """Run `produce` in a worker and post `(rank, pickled result)`; anything it raises — including a result that cannot
    be pickled, which inside the queue's feeder thread would be lost without a trace — is posted as the exception."""
    import pickle

    try:
        payload = pickle.dumps(produce())
    except BaseException as e:  # the parent must not wait forever for a rank that died early
        try:
            payload = pickle.dumps(e)
            pickle.loads(payload)  # an exception class with a multi-argument __init__ dumps fine and fails to LOAD
        except Exception:
            payload = pickle.dumps(RuntimeError(f"rank {rank}: {type(e).__name__}: {e}"))
    queue.put((rank, payload))


def _collect(procs, queue, what: str, poll_s: float = 0.5) -> Dict[int, Any]:
    """One posted result per worker, polling so that a worker that died NATIVELY (HIP abort, segfault, OOM kill — no
    Python exception, nothing posted) is noticed: its exit code is reported instead of waiting for it forever."""
    import pickle
    import queue as queue_mod

    pending = set(range(len(procs)))
    merged: Dict[int, Any] = {}
    failed: Optional[BaseException] = None
    grace: Dict[int, int] = {}
    while pending:
        try:
            rank, payload = queue.get(timeout=poll_s)
        except queue_mod.Empty:
            for r in sorted(pending):
                if not procs[r].is_alive():
                    # a worker that posted just before exiting: give the pipe a few polls to deliver
                    grace[r] = grace.get(r, 0) + 1
                    if grace[r] >= 4:
                        pending.discard(r)
                        failed = failed or RuntimeError(
                            f"{what}: worker {r} exited with code {procs[r].exitcode} without posting a result")
            continue
        pending.discard(rank)
        try:
            part = pickle.loads(payload)
        except Exception as e:  # never abort the collection: the other workers are still running and must be joined
            part = RuntimeError(f"{what}: the result of worker {rank} could not be unpickled: {type(e).__name__}: {e}")
        if isinstance(part, BaseException):
            failed = failed or part
        else:
            merged.update(part)
    for p in procs:
        p.join(timeout=30)
        if p.is_alive():
            p.terminate()
    if failed is not None:
        raise RuntimeError(f"{what}: a worker failed: {failed!r}") from failed
    return merged


def _spawned_worker(rank: int, world: int, device: int, paths, model_path, model_factory, kwargs, queue) -> None:
    def produce():
        files, pieces = _units_of_rank(paths, world, rank)
        return _predict_shard(paths, files, device, model_path, model_factory, kwargs, pieces)

    _post(queue, rank, produce)


def predict_many_sharded(
    audio_paths: Sequence[Any],
    model_or_model_path: Any = None,
    gpus: Optional[int] = None,
    model_factory: Optional[Callable[[int], Any]] = None,
    workers_per_gpu: int = 1,
    **predict_kwargs: Any,
) -> Optional[List[Any]]:
    """`predict()` (inference.py:431-506) for every file of `audio_paths`, file-sharded over the GPUs of one node.

    Returns, in input order, the `(model_output, midi_data, note_events)` tuple of each file — or the exception that
    file raised (per-file isolation, like the try / except of inference.py:548-604) — identical to what
    `predict_many` returns on one GPU.  Files are assigned by the LPT plan over their sizes (`plan_units`), every
    rank owns a `Model` on its own GPU and nothing but finished per-file results crosses rank boundaries (host side).
    A file that alone outweighs an even share of the job (one very long recording among short ones, or a job of one file)
    is cut into window ranges (`split_windows`), one per rank that takes a piece: every piece's rank decodes the file and
    computes its own windows (`inference.predict_window_range`), rank 0 / the parent concatenates the rows and decodes the
    notes over the whole file (`inference.assemble_window_ranges`) — bit-identical to the unsplit result.

      * inside a `torch.distributed` job (launched one rank per GPU): every rank calls this with the same list; the
        rank's GPU is LOCAL_RANK; results are gathered with `gather_object` and returned on rank 0 (None elsewhere);
      * otherwise `gpus` (default: all visible) x `workers_per_gpu` worker processes are spawned from here and the
        merged list is returned; `gpus=1, workers_per_gpu=1` runs in this process.  (`workers_per_gpu` > 1 puts several
        processes, each with its own handle, on one GPU.  One host process sustains ~33 three-minute files per second
        — file read, PCIe copies and the Python half of the MIDI assembly, against 0.4 ms of GPU time per file — but
        handing the posteriorgrams (27 MB per file) back through pipes costs more than the extra processes gain:
        measured 15-17 files/s with 4-8 workers.  It pays only when the workers keep their results.)

    `model_factory(device_ordinal)` overrides how a rank builds its model (tests use it to stub the compute);
    `predict_kwargs` are `predict_many`'s (thresholds, `group`, `decode_threads`, ...).
    """
    from . import inference

    paths = [os.fspath(p) for p in audio_paths]
    if model_or_model_path is None:
        model_or_model_path = inference.ICASSP_2022_MODEL_PATH
    dist = None
    try:
        import torch.distributed as dist_mod

        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
    except ImportError:
        pass
    if dist is not None:
        rank, world = dist.get_rank(), dist.get_world_size()
        files, pieces = _units_of_rank(paths, world, rank)
        device = int(os.environ.get("LOCAL_RANK", rank))
        mine = _predict_shard(paths, files, device, model_or_model_path, model_factory, predict_kwargs, pieces)
        bucket: Optional[List[Any]] = [None] * world if rank == 0 else None
        dist.gather_object(mine, bucket, dst=0)
        if rank != 0:
            return None
        merged: Dict[Any, Any] = {}
        for part in bucket or []:
            merged.update(part or {})
        return _merge_units(merged, len(paths), predict_kwargs)

    if gpus is None:
        import torch

        gpus = max(1, torch.cuda.device_count())
    if gpus < 1 or workers_per_gpu < 1:
        raise ValueError("gpus and workers_per_gpu must be >= 1")
    world = gpus * workers_per_gpu
    if world == 1:
        mine = _predict_shard(paths, list(range(len(paths))), 0, model_or_model_path, model_factory, predict_kwargs)
        return [mine[i] for i in range(len(paths))]
    if isinstance(model_or_model_path, inference.Model):
        raise ValueError("pass a model path (not a Model bound to one GPU) when spawning one worker per GPU")
    import multiprocessing as mp

    ctx = mp.get_context("spawn")  # HIP contexts do not survive fork
    queue = ctx.Queue()
    procs = [ctx.Process(target=_spawned_worker, args=(r, world, r % gpus, paths, os.fspath(model_or_model_path),
                                                        model_factory, predict_kwargs, queue), daemon=True)
             for r in range(world)]
    for p in procs:
        p.start()
    merged = _collect(procs, queue, "predict_many_sharded")
    return _merge_units(merged, len(paths), predict_kwargs)


def _save_shard(paths: Sequence[Any], indices: Sequence[int], device: int, model_or_model_path: Any,
                model_factory: Optional[Callable[[int], Any]], output_directory: Any, save_flags: Dict[str, Any],
                kwargs: Dict[str, Any]) -> Dict[int, Any]:
    """One worker's share of a `predict_and_save` job: predict AND write its files; only small reports leave the worker."""
    from . import inference

    if not indices:
        return {}
    kwargs = dict(kwargs)
    native = kwargs.pop("native", False)
    native_lanes, native_threads = int(kwargs.pop("native_lanes", 3)), int(kwargs.pop("native_threads", 0))
    if model_factory is not None:
        model = model_factory(device)
    elif isinstance(model_or_model_path, inference.Model):
        model = model_or_model_path
    else:  # the native pipeline's lanes wait asleep: its workers share the cores with the lanes' waiting threads
        model = inference.Model(model_or_model_path, device=device, blocking_wait=bool(native))
    if native:
        # the native pipeline (bp_transcribe_files: C++ worker threads, no Python in the loop) for this worker's share
        if save_flags.get("sonify_midi") or save_flags.get("save_model_outputs"):
            raise ValueError("native=True writes .mid and .csv only (no sonification, no .npz model outputs)")
        for k in ("group", "decode_threads", "sonification_samplerate"):
            kwargs.pop(k, None)
        lanes = [model]
        if model_factory is None and not isinstance(model_or_model_path, inference.Model):
            lanes += [inference.Model(model_or_model_path, device=device, blocking_wait=True) for _ in range(max(0, native_lanes - 1))]
        sel = [paths[i] for i in indices]
        raw = inference.transcribe_files(sel, output_directory, save_flags.get("save_midi", True), save_flags.get("save_notes", True),
                                         models=lanes, threads=native_threads, **kwargs)
        rep = []
        for pth, r in zip(sel, raw):
            if r["status"] != 0:
                rep.append(IOError(r["message"]) if "already exists" in r["message"] or "same file stem" in r["message"]
                           else ValueError(r["message"]))
                continue
            stem = os.path.splitext(os.path.basename(str(pth)))[0]
            outs = {}
            if save_flags.get("save_midi", True):
                outs["midi"] = os.path.join(os.fspath(output_directory), f"{stem}_basic_pitch.mid")
            if save_flags.get("save_notes", True):
                outs["note_events"] = os.path.join(os.fspath(output_directory), f"{stem}_basic_pitch.csv")
            rep.append({"n_note_events": r["n_note_events"], "outputs": outs})
        for m in lanes[1:]:
            m.close()
        return dict(zip(indices, rep))
    rep = inference.predict_and_save_many([paths[i] for i in indices], output_directory, model_or_model_path=model,
                                          return_exceptions=True, **save_flags, **kwargs)
    return dict(zip(indices, rep))


def _spawned_saver(rank: int, world: int, device: int, paths, model_path, model_factory, output_directory, save_flags,
                   kwargs, queue) -> None:
    def produce():
        shards = plan_shards(_file_costs(paths), world)
        return _save_shard(paths, shards[rank], device, model_path, model_factory, output_directory, save_flags, kwargs)

    _post(queue, rank, produce)


def predict_and_save_sharded(
    audio_paths: Sequence[Any],
    output_directory: Any,
    save_midi: bool,
    sonify_midi: bool,
    save_model_outputs: bool,
    save_notes: bool,
    model_or_model_path: Any = None,
    gpus: Optional[int] = None,
    workers_per_gpu: int = 1,
    model_factory: Optional[Callable[[int], Any]] = None,
    **predict_kwargs: Any,
) -> Optional[List[Any]]:
    """`predict_and_save` (inference.py:509-618) for a list of files on all GPUs of a node — the batch transcription
    job of the north star.  Files are LPT-sharded by size over `gpus x workers_per_gpu` processes (or over the ranks of an
    existing `torch.distributed` job); every worker owns a handle on its GPU, runs `predict_and_save_many` over its
    share and WRITES its outputs itself: nothing but a per-file report `{"n_note_events": k, "outputs": {kind: path}}`
    (or the file's exception) travels back, so — unlike `predict_many_sharded`, whose 27 MB of posteriorgrams per
    three-minute file cost more in the pipes than extra processes gain — several host processes per GPU do scale: a
    file costs 0.4 ms of GPU time and ~30 ms of host time (read, PCIe, the Python half of note decoding and the writers).

    Returns the reports in input order (on rank 0 inside a distributed job, None elsewhere).  `predict_kwargs` are
    `predict_and_save_many`'s (thresholds, `sonification_samplerate`, `midi_tempo`, `group`, `decode_threads`).

    `native=True` (round 4; with `native_lanes`, `native_threads`): every worker runs its share through the native
    pipeline (`inference.transcribe_files` -> `bp_transcribe_files`: C++ worker threads from a file's bytes to its
    `.mid` / `.csv`) instead of the Python one: 299 instead of 85 three-minute files per second and GPU with ONE process per
    GPU; give each worker `native_threads` = usable cores / GPUs.  MIDI and note events only."""
    from . import inference

    paths = [os.fspath(p) for p in audio_paths]
    if model_or_model_path is None:
        model_or_model_path = inference.ICASSP_2022_MODEL_PATH
    inference.verify_output_dir(output_directory)
    # two inputs with the same stem would race on the exists-check of build_output_path when they land on different
    # workers: decided here, before sharding (the first wins, later ones get the reference's IOError in place)
    dup = inference.duplicate_output_stems(paths)
    if dup:
        keep = [i for i in range(len(paths)) if i not in dup]
        sub = predict_and_save_sharded([paths[i] for i in keep], output_directory, save_midi, sonify_midi,
                                       save_model_outputs, save_notes, model_or_model_path, gpus, workers_per_gpu,
                                       model_factory, **predict_kwargs)
        if sub is None:
            return None
        merged_all: List[Any] = [None] * len(paths)
        for i, r in zip(keep, sub):
            merged_all[i] = r
        for i, e in dup.items():
            merged_all[i] = e
        return merged_all
    flags = {"save_midi": save_midi, "sonify_midi": sonify_midi, "save_model_outputs": save_model_outputs,
             "save_notes": save_notes}
    dist = None
    try:
        import torch.distributed as dist_mod

        if dist_mod.is_available() and dist_mod.is_initialized():
            dist = dist_mod
    except ImportError:
        pass
    if dist is not None:
        rank, world = dist.get_rank(), dist.get_world_size()
        shards = plan_shards(_file_costs(paths), world)
        device = int(os.environ.get("LOCAL_RANK", rank))
        mine = _save_shard(paths, shards[rank], device, model_or_model_path, model_factory, output_directory, flags,
                           predict_kwargs)
        bucket: Optional[List[Any]] = [None] * world if rank == 0 else None
        dist.gather_object(mine, bucket, dst=0)
        if rank != 0:
            return None
        merged: Dict[int, Any] = {}
        for part in bucket or []:
            merged.update(part or {})
        return [merged[i] for i in range(len(paths))]

    if gpus is None:
        import torch

        gpus = max(1, torch.cuda.device_count())
    if gpus < 1 or workers_per_gpu < 1:
        raise ValueError("gpus and workers_per_gpu must be >= 1")
    world = gpus * workers_per_gpu
    if world == 1:
        mine = _save_shard(paths, list(range(len(paths))), 0, model_or_model_path, model_factory, output_directory, flags,
                           predict_kwargs)
        return [mine[i] for i in range(len(paths))]
    if isinstance(model_or_model_path, inference.Model):
        raise ValueError("pass a model path (not a Model bound to one GPU) when spawning worker processes")
    import multiprocessing as mp

    ctx = mp.get_context("spawn")  # HIP contexts do not survive fork
    queue = ctx.Queue()
    procs = [ctx.Process(target=_spawned_saver, args=(r, world, r % gpus, paths, os.fspath(model_or_model_path),
                                                       model_factory, os.fspath(output_directory), flags, predict_kwargs,
                                                       queue), daemon=True)
             for r in range(world)]
    for p in procs:
        p.start()
    merged = _collect(procs, queue, "predict_and_save_sharded")
    return [merged[i] for i in range(len(paths))]
