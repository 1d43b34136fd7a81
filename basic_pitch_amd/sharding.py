"""File-level sharding of transcription jobs across the GPUs of one node.

The hot path has no exchange step: every 2-second window is an independent unit (per-window
normalisation, basic_pitch/layers/signal.py:177-183; stitching is a concat/trim,
basic_pitch/inference.py:267-279).  So N GPUs = N processes, each with its own handle, each owning a
disjoint set of files; the only "communication" is handing per-file results (or their paths) back to
rank 0 on the host.  No RCCL collective touches the data path.

`plan_shards` is the longest-processing-time-first assignment by sample count; `run_sharded` is the
per-rank driver used under `python -m torch.distributed.run` (backend gloo or nccl: it only uses
object gathers on the host).
"""
from __future__ import annotations

from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple


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
