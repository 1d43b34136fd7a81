"""Run single HIP stages through the C ABI test hook (bp_run_stage) on torch CUDA buffers.

torch is only the device allocator here.  Inputs for each stage are the ORACLE's tensors for that
stage, so a defect in one kernel cannot hide behind (or be blamed on) the kernels upstream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from basic_pitch_amd import _native
from basic_pitch_amd.inference import Model

STAGE = {n: i for i, n in enumerate(_native.STAGE_NAMES)}


def ord_encode(x: np.ndarray) -> np.ndarray:
    """float32 -> order-preserving int32 (bp_common.h f2ord)."""
    i = np.ascontiguousarray(x, dtype=np.float32).view(np.int32)
    return np.where(i >= 0, i, i ^ np.int32(0x7FFFFFFF)).astype(np.int32)


def ord_decode(i: np.ndarray) -> np.ndarray:
    i = np.ascontiguousarray(i, dtype=np.int32)
    return np.where(i >= 0, i, i ^ np.int32(0x7FFFFFFF)).astype(np.int32).view(np.float32)


def zp_pack(z: np.ndarray) -> np.ndarray:
    """z (n,172,309) fp32 -> the library's pre-split, zero-padded words (n,174,448): f16 hi | f16 lo << 16 with
    lo = (z - hi) * 2^11; frame t, bin g at [t + 1][56 + g] (include/basic_pitch_amd.h: BP_Z_ROWS / ROW / PAD)."""
    z = np.ascontiguousarray(z, dtype=np.float32)
    hi = z.astype(np.float16)
    lo = ((z - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)  # kLoScale (bp_common.h)
    u = hi.view(np.uint16).astype(np.uint32) | (lo.view(np.uint16).astype(np.uint32) << 16)
    out = np.zeros((z.shape[0], _native.BP_Z_ROWS, _native.BP_Z_ROW), dtype=np.uint32)
    out[:, 1 : 1 + z.shape[1], _native.BP_Z_PAD : _native.BP_Z_PAD + z.shape[2]] = u
    return out


def zp_unpack(zp: np.ndarray) -> np.ndarray:
    """inverse of zp_pack (hi + lo in fp32), padding dropped"""
    zp = np.ascontiguousarray(zp).view(np.uint32)[:, 1:173, _native.BP_Z_PAD : _native.BP_Z_PAD + 309]
    hi = (zp & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float32)
    lo = (zp >> 16).astype(np.uint16).view(np.float16).astype(np.float32)
    return hi + lo / np.float32(2048.0)


def pyr_pack(levels, lib) -> np.ndarray:
    """Oracle pyramid levels [1..8] -> the library's pyr row layout."""
    n = levels[1].shape[0]
    out = np.zeros((n, _native.BP_PYR_STRIDE), dtype=np.float32)
    for k in range(1, 9):
        off, ln = C.c_int64(), C.c_int64()
        assert lib.bp_pyramid_layout(k, C.byref(off), C.byref(ln)) == 0
        assert levels[k].shape[1] == ln.value
        out[:, off.value : off.value + ln.value] = levels[k]
    return out


def pyr_unpack(pyr: np.ndarray, lib):
    levels = [None]
    for k in range(1, 9):
        off, ln = C.c_int64(), C.c_int64()
        assert lib.bp_pyramid_layout(k, C.byref(off), C.byref(ln)) == 0
        levels.append(pyr[:, off.value : off.value + ln.value])
    return levels


class StageRunner:
    def __init__(self, model: Optional[Model] = None):
        self.model = model or Model()
        self.lib = self.model._lib
        self.dev = torch.device("cuda", self.model.device)

    def _t(self, a: Optional[np.ndarray]):
        return None if a is None else torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def run(self, stage: str, n: int, inputs: Dict[str, np.ndarray], outputs: Dict[str, tuple]) -> Dict[str, np.ndarray]:
        """inputs: name -> array; outputs: name -> (shape, dtype).  Names are bp_stage_buffers fields."""
        bufs = _native.bp_stage_buffers()
        keep = {}
        for k, v in inputs.items():
            keep[k] = self._t(v)
            setattr(bufs, k, keep[k].data_ptr())
        for k, (shape, dtype) in outputs.items():
            t = torch.full(shape, float("nan") if dtype == torch.float32 else 0, dtype=dtype, device=self.dev)
            keep[k] = t
            setattr(bufs, k, t.data_ptr())
        torch.cuda.synchronize()
        rc = self.lib.bp_run_stage(self.model._handle, STAGE[stage], C.byref(bufs), n)
        _native.check(self.lib, self.model._handle, rc, f"bp_run_stage({stage})")
        return {k: keep[k].cpu().numpy() for k in outputs}
