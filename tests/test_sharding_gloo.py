"""N > 1 path on CPU: two processes over gloo run the file-sharded driver.

The per-file work here is the ORACLE on one tiny window (tests may use the oracle as the compute
stand-in; on the GPU box the same driver wraps Model.predict_track).  What is under test is the
protocol: deterministic LPT plan on every rank, disjoint coverage, per-item error isolation, host-side
gather to rank 0, no device collective."""
import os
import socket
import sys

import numpy as np
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
