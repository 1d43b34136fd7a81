#!/usr/bin/env python
"""GPU diagnostic: every HIP stage vs the oracle, stage by stage and end to end (prints, no asserts).

Run on the GPU box:  python tools/gpu_stage_report.py [n_windows]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import make_windows  # noqa: E402
from oracle import bp_oracle as O  # noqa: E402
from stage_harness import StageRunner, ord_decode, ord_encode, pyr_pack, pyr_unpack  # noqa: E402

F32 = torch.float32


def err(name, got, ref, ref64=None):
    got = np.asarray(got, dtype=np.float64)
    d = np.abs(got - ref)
    nan = int(np.isnan(got).sum())
    msg = f"  {name:10s} max|d|={np.nanmax(d):.3e} mean={np.nanmean(d):.3e} ref_max={np.abs(ref).max():.3e} nan={nan}"
    if ref64 is not None:
        msg += f"  | vs fp64: ours {np.nanmax(np.abs(got - ref64)):.3e}  oracle32 {np.abs(ref - ref64).max():.3e}"
    print(msg, flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    W = O.load_weights()
    x = np.concatenate([make_windows("uniform", 1, 0), make_windows("normal", 1, 1), make_windows("tones", max(1, n - 2), 2)])[:n]
    n = x.shape[0]
    t0 = time.time()
    r32 = O.forward(x, W, np.float32, intermediates=True)
    r64 = O.forward(x, W, np.float64, intermediates=True)
    print(f"oracle fp32+fp64 for {n} windows: {time.time()-t0:.1f}s")

    sr = StageRunner()
    print("device:", sr.model.info())
    lib = sr.lib

    print("[pyramid] input: audio")
    out = sr.run("pyramid", n, {"audio": x}, {"pyr": ((n, 43712), F32)})
    lv = pyr_unpack(out["pyr"], lib)
    for k in range(1, 9):
        err(f"level{k}", lv[k], r32["levels"][k], r64["levels"][k])

    print("[filterbank] input: oracle fp32 pyramid")
    pyr = pyr_pack(r32["levels"], lib)
    out = sr.run("filterbank", n, {"audio": x, "pyr": pyr}, {"lp": ((n, 172, 309), F32), "mm": ((n, 2), torch.int32)})
    mag = np.sqrt(np.maximum(10.0 ** (out["lp"].astype(np.float64) / 10.0) - 1e-10, 0))
    err("mag", mag, r32["mag"], r64["mag"])
    err("lp", out["lp"], r32["lp"], r64["lp"])
    mm = ord_decode(out["mm"])
    print("  mm ours", mm.tolist(), "\n  mm ref ", r32["minmax"].tolist())
    for lvl in range(9):
        lo = max(0, (8 - lvl) * 36 - 15)
        hi = (8 - lvl) * 36 + 21
        d = np.abs(mag[:, :, lo:hi] - r64["mag"][:, :, lo:hi]).max()
        print(f"    level {lvl} bins [{lo},{hi}) max|mag-mag64|={d:.3e}  (mag max {r64['mag'][:, :, lo:hi].max():.3e})")

    mm_in = ord_encode(r32["minmax"])
    print("[contour1] input: oracle lp/minmax")
    out = sr.run("contour1", n, {"lp": r32["lp"], "mm": mm_in}, {"c1": ((n, 8, 172, 264), F32)})
    err("c1", out["c1"], r32["c1"], r64["c1"])
    for c in range(8):
        print(f"    out ch {c}: max|d|={np.abs(out['c1'][:, c] - r32['c1'][:, c]).max():.3e}")

    print("[contour2] input: oracle c1")
    out = sr.run("contour2", n, {"c1": r32["c1"]}, {"contour": ((n, 172, 264), F32)})
    err("contour", out["contour"], r32["contour"], r64["contour"])

    print("[note1] input: oracle contour")
    out = sr.run("note1", n, {"contour": r32["contour"]}, {"n1": ((n, 32, 172, 88), F32)})
    err("n1", out["n1"], r32["n1"], r64["n1"])

    print("[note2] input: oracle n1")
    out = sr.run("note2", n, {"n1": r32["n1"]}, {"note": ((n, 172, 88), F32)})
    err("note", out["note"], r32["note"], r64["note"])

    print("[onset1] input: oracle lp/minmax")
    out = sr.run("onset1", n, {"lp": r32["lp"], "mm": mm_in}, {"o1": ((n, 32, 172, 88), F32)})
    err("o1", out["o1"], r32["o1"], r64["o1"])

    print("[onset2] input: oracle note, o1")
    out = sr.run("onset2", n, {"note": r32["note"], "o1": r32["o1"]}, {"onset": ((n, 172, 88), F32)})
    err("onset", out["onset"], r32["onset"], r64["onset"])

    print("[end-to-end] Model.predict (host buffers)")
    res = sr.model.predict(x)
    for k in ("contour", "note", "onset"):
        err(k, res[k], r32[k], r64[k])
    print("[end-to-end] Model.predict (device tensors)")
    xt = torch.from_numpy(x).cuda()
    res_d = sr.model.predict(xt)
    for k in ("contour", "note", "onset"):
        print(f"  {k}: host-vs-device identical: {np.array_equal(res_d[k].cpu().numpy(), res[k])}")
    for w in range(n):
        print(f"  window {w}: " + " ".join(
            f"{k}: ours-64 {np.abs(res[k][w]-r64[k][w]).max():.2e} o32-64 {np.abs(r32[k][w]-r64[k][w]).max():.2e}" for k in ("contour", "note", "onset")))


if __name__ == "__main__":
    main()
