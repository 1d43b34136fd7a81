#!/usr/bin/env python
"""Dev-time asset builder: frozen-graph constants -> basic_pitch_amd/assets/nmp_weights.bin.

Reads the reference's serialized model (`basic_pitch/saved_models/icassp_2022/nmp.onnx`,
SURVEY.md App. A) with the wire reader in tools/onnx_wire.py and writes the 18 tensors the hot
path needs into ONE flat little-endian file the C library and the oracle both parse
(format: include/basic_pitch_amd.h, "weights blob").  The script also regenerates the four CQT
constants from the formulas in `basic_pitch/layers/nnaudio.py:45-76,158-213,530-593` with scipy and
asserts that they are bit-identical to the ONNX initializers, so the blob is pinned to the
reference's source as well as to its artifact.

Run here (needs /root/reference); the output is committed, the GPU box never runs this.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from onnx_wire import load_graph  # noqa: E402

MAGIC = b"BPAMDW01"

# order of CNN Conv nodes in the graph
CNN_ORDER = ["onset1", "contour1", "contour2", "note1", "note2", "onset2"]
CNN_SHAPES = {
    "onset1": (32, 8, 5, 5),
    "contour1": (8, 8, 3, 39),
    "contour2": (1, 8, 5, 5),
    "note1": (32, 1, 7, 7),
    "note2": (1, 32, 7, 3),
    "onset2": (1, 33, 3, 3),
}


def regenerate_cqt_constants():
    """nnaudio.py formulas (build(): 530-593; create_cqt_kernels: 158-213; lowpass: 45-76)."""
    import scipy.signal

    sr, fmin, n_bins, bpo = 22050.0, 27.5, 309, 36
    Q = 1.0 / (2 ** (1 / bpo) - 1)
    lowpass = scipy.signal.firwin2(256, [0.0, 0.5 / 1.001, 0.5 * 1.001, 1.0], [1.0, 1.0, 0.0, 0.0])
    n_octaves = int(np.ceil(float(n_bins) / bpo))
    fmin_t = fmin * 2 ** (n_octaves - 1)
    remainder = n_bins % bpo
    fmax_t = fmin_t * 2 ** (((remainder if remainder else bpo) - 1) / bpo)
    fmin_t = fmax_t / 2 ** (1 - 1 / bpo)
    fft_len = 2 ** int(np.ceil(np.log2(np.ceil(Q * sr / fmin_t))))
    freqs = fmin_t * 2.0 ** (np.r_[0:bpo] / float(bpo))
    kern = np.zeros((bpo, int(fft_len)), dtype=np.complex64)
    for k in range(bpo):
        freq = freqs[k]
        _l = np.ceil(Q * sr / freq)
        start = int(np.ceil(fft_len / 2.0 - _l / 2.0)) - int(_l % 2)
        sig = (
            scipy.signal.get_window("hann", int(_l), fftbins=True)
            * np.exp(np.r_[-_l // 2 : _l // 2] * 1j * 2 * np.pi * freq / sr)
            / _l
        )
        kern[k, start : start + int(_l)] = sig / np.linalg.norm(sig, 1)
    all_freqs = fmin * 2.0 ** (np.r_[0:n_bins] / float(bpo))
    lengths = np.ceil(Q * sr / all_freqs)
    return (
        kern.real.astype(np.float32),
        kern.imag.astype(np.float32),
        lowpass.astype(np.float32),
        np.sqrt(lengths.astype(np.float32)).astype(np.float32),
    )


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--onnx", default="/root/reference/basic_pitch/saved_models/icassp_2022/nmp.onnx")
    ap.add_argument(
        "--out",
        default=os.path.join(os.path.dirname(__file__), "..", "basic_pitch_amd", "assets", "nmp_weights.bin"),
    )
    args = ap.parse_args()

    nodes, inits = load_graph(args.onnx)
    convs = [n for n in nodes if n["op_type"] == "Conv"]
    cqt_convs = [n for n in convs if n["attr"]["kernel_shape"] == [1, 256]]
    cnn_convs = [n for n in convs if n["attr"]["kernel_shape"] != [1, 256]]
    assert len(cqt_convs) == 26 and len(cnn_convs) == 6, (len(cqt_convs), len(cnn_convs))

    tensors = {}

    # --- CQT constants: the two 36-filter banks and the 1-filter decimator.
    banks = {}
    for n in cqt_convs:
        w = inits[n["input"][1]]
        b = inits[n["input"][2]]
        assert np.all(b == 0.0), "CQT conv bias must be zero"
        banks[n["input"][1]] = w
    bank36 = {k: v for k, v in banks.items() if v.shape[0] == 36}
    low = [v for v in banks.values() if v.shape[0] == 1]
    assert len(bank36) == 2 and len(low) == 1
    re_f, im_f, lp_f, sq_f = regenerate_cqt_constants()
    re = im = None
    for name, w in bank36.items():
        w2 = w.reshape(36, 256)
        if np.array_equal(w2, re_f):
            re = w2
        elif np.array_equal(w2, im_f):
            im = w2
    assert re is not None and im is not None, "ONNX CQT banks do not match the nnaudio.py formulas bit-exactly"
    lowpass = low[0].reshape(256)
    assert np.array_equal(lowpass, lp_f), "lowpass differs from firwin2 regeneration"
    sq = [v for k, v in inits.items() if v.dtype == np.float32 and v.size == 309]
    assert len(sq) == 1
    sqrt_len = sq[0].reshape(309)
    assert np.array_equal(sqrt_len, sq_f), "sqrt(lengths) differs from regeneration"
    tensors["cqt_kernel_re"] = re
    tensors["cqt_kernel_im"] = im
    tensors["cqt_lowpass"] = lowpass
    tensors["cqt_sqrt_len"] = sqrt_len

    # --- the imag conv output is negated in the graph (nnaudio.py:246); check a Neg node follows.
    assert sum(1 for n in nodes if n["op_type"] == "Neg") == 9

    # --- NormalizedLog constants and the folded BatchNorm affine (nodes Add/Mul/Mul and Mul/Add).
    by_out = {o: n for n in nodes for o in n["output"]}
    log_node = [n for n in nodes if n["op_type"] == "Log"]
    assert len(log_node) == 1
    add_node = by_out[log_node[0]["input"][0]]
    eps = [inits[i] for i in add_node["input"] if i in inits]
    assert len(eps) == 1
    tensors["log_eps"] = eps[0].reshape(1)
    consumers = lambda name: [n for n in nodes if name in n["input"]]  # noqa: E731
    mul1 = consumers(log_node[0]["output"][0])[0]
    c1 = [inits[i] for i in mul1["input"] if i in inits][0]
    mul2 = consumers(mul1["output"][0])[0]
    c2 = [inits[i] for i in mul2["input"] if i in inits][0]
    tensors["log_scale"] = np.asarray([c1.reshape(()), c2.reshape(())], dtype=np.float32)  # 1/ln10, 10

    bn_mul = [n for n in nodes if n["op_type"] == "Mul" and "batch_normalization/FusedBatchNormV3" in n["output"][0]]
    assert len(bn_mul) == 1
    bn_scale = [inits[i] for i in bn_mul[0]["input"] if i in inits][0]
    bn_add = consumers(bn_mul[0]["output"][0])[0]
    assert bn_add["op_type"] == "Add"
    bn_shift = [inits[i] for i in bn_add["input"] if i in inits][0]
    tensors["bn_affine"] = np.asarray([bn_scale.reshape(()), bn_shift.reshape(())], dtype=np.float32)

    # --- CNN: six Conv nodes in graph order.
    for name, n in zip(CNN_ORDER, cnn_convs):
        w = inits[n["input"][1]]
        b = inits[n["input"][2]]
        assert tuple(w.shape) == CNN_SHAPES[name], (name, w.shape)
        assert b.shape == (w.shape[0],)
        tensors[name + "_w"] = w.astype(np.float32)
        tensors[name + "_b"] = b.astype(np.float32)

    # --- serialise.
    names = list(tensors.keys())
    entries = b""
    data = b""
    off = 0
    for nm in names:
        arr = np.ascontiguousarray(tensors[nm], dtype="<f4")
        dims = list(arr.shape) + [1] * (4 - arr.ndim)
        entries += struct.pack("<24sI4III", nm.encode(), arr.ndim, *dims, off, arr.size)
        data += arr.tobytes()
        off += arr.size
    blob = MAGIC + struct.pack("<II", 1, len(names)) + entries + data
    out = os.path.abspath(args.out)
    with open(out, "wb") as f:
        f.write(blob)
    manifest = {
        "source": "basic_pitch/saved_models/icassp_2022/nmp.onnx (reference v0.4.0)",
        "sha256": hashlib.sha256(blob).hexdigest(),
        "tensors": {nm: list(tensors[nm].shape) for nm in names},
        "scalars": {
            "log_eps": float(tensors["log_eps"][0]),
            "log_scale": [float(x) for x in tensors["log_scale"]],
            "bn_affine": [float(x) for x in tensors["bn_affine"]],
        },
    }
    with open(out.replace(".bin", ".json"), "w") as f:
        json.dump(manifest, f, indent=1)
    print(json.dumps(manifest, indent=1))
    print("wrote", out, len(blob), "bytes")


if __name__ == "__main__":
    main()
