"""Serialized model -> the weights blob `bp_create` takes (format: include/basic_pitch_amd.h, "weights blob").

`basic_pitch.inference.Model(model_path)` (inference.py:78-154) is given the path of a serialized model; the only
in-tree carrier of the trained weights readable without TensorFlow is `saved_models/icassp_2022/nmp.onnx`
(SURVEY.md §8a row a16).  `load_model_blob(path)` accepts
  * a weights blob (`assets/nmp_weights.bin`, magic BPAMDW01) — returned as is;
  * an ONNX file of the Basic Pitch graph: its 18 constants are extracted here with the wire reader in onnx_wire.py
    (structure checked: 26 CQT convolutions sharing two 36-filter banks and one decimator, 6 CNN convolutions of the
    published shapes, the NormalizedLog constants and the folded BatchNorm affine), no `onnx` package needed;
  * the reference's other artifacts of the same model (`nmp/` SavedModel directory, `nmp.tflite`, `nmp.mlpackage`) —
    `basic_pitch.ICASSP_2022_MODEL_PATH` points at whichever matches the installed runtime — resolved to the
    `nmp.onnx` next to them, which holds the same weights.
Anything else raises ValueError like an unloadable model in the reference (inference.py:148-154).
"""
from __future__ import annotations

import os
import pathlib
import struct
from typing import Dict, Union

import numpy as np

from .onnx_wire import load_graph

MAGIC = b"BPAMDW01"
NMP_ONNX_SHA256 = "2c3c1d144bfa61ad236e92e169c13535c880469a12a047d4e73451f2c059a0ec"  # reference v0.4.0 artifact

# order of the CNN Conv nodes in the graph and their published shapes (models.py:241-318)
CNN_ORDER = ["onset1", "contour1", "contour2", "note1", "note2", "onset2"]
CNN_SHAPES = {
    "onset1": (32, 8, 5, 5),
    "contour1": (8, 8, 3, 39),
    "contour2": (1, 8, 5, 5),
    "note1": (32, 1, 7, 7),
    "note2": (1, 32, 7, 3),
    "onset2": (1, 33, 3, 3),
}


def _require(cond: bool, what: str) -> None:
    if not cond:
        raise ValueError(f"not the Basic Pitch graph: {what}")


def tensors_from_onnx(path_or_bytes) -> Dict[str, np.ndarray]:
    """The 18 tensors of the frozen graph, by the names the C library looks up.  Anything that is not that graph — a
    non-protobuf file, another model, a structurally damaged one — raises ValueError, the reference's contract for an
    unloadable model (inference.py:148-154)."""
    try:
        nodes, inits = load_graph(path_or_bytes)
    except Exception as e:  # the wire reader on a non-protobuf file
        raise ValueError(f"cannot be read as an ONNX model: {e}") from e
    try:
        return _tensors_from_graph(nodes, inits)
    except (KeyError, IndexError, TypeError, AttributeError) as e:  # a lookup the Basic Pitch graph always satisfies
        raise ValueError(f"not the Basic Pitch graph: {type(e).__name__}: {e}") from e


def _tensors_from_graph(nodes, inits) -> Dict[str, np.ndarray]:
    convs = [n for n in nodes if n["op_type"] == "Conv"]
    cqt_convs = [n for n in convs if n["attr"].get("kernel_shape") == [1, 256]]
    cnn_convs = [n for n in convs if n["attr"].get("kernel_shape") != [1, 256]]
    _require(len(cqt_convs) == 26 and len(cnn_convs) == 6, f"{len(cqt_convs)} CQT / {len(cnn_convs)} CNN convolutions")
    consumers = lambda name: [n for n in nodes if name in n["input"]]  # noqa: E731
    by_out = {o: n for n in nodes for o in n["output"]}
    tensors: Dict[str, np.ndarray] = {}

    # CQT: two 36-filter banks (the one whose result is negated, nnaudio.py:246, is the imaginary part) + decimator
    re = im = low = None
    for n in cqt_convs:
        w = inits[n["input"][1]]
        _require(np.all(inits[n["input"][2]] == 0.0), "CQT convolution with a bias")
        if w.shape[0] == 1:
            low = w.reshape(256)
            continue
        _require(w.shape[0] == 36, f"CQT bank of {w.shape[0]} filters")
        negated = False
        out, hops = n["output"][0], 0
        while hops < 4 and not negated:  # Conv -> (Squeeze / Transpose ...) -> Neg
            nxt = consumers(out)
            if len(nxt) != 1:
                break
            negated = nxt[0]["op_type"] == "Neg"
            out, hops = nxt[0]["output"][0], hops + 1
        if negated:
            im = w.reshape(36, 256)
        else:
            re = w.reshape(36, 256)
    _require(re is not None and im is not None and low is not None, "CQT banks / decimator not found")
    _require(sum(1 for n in nodes if n["op_type"] == "Neg") == 9, "expected 9 negated imaginary parts")
    sq = [v for v in inits.values() if v.dtype == np.float32 and v.size == 309]
    _require(len(sq) == 1, "sqrt(lengths) table not found")
    tensors["cqt_kernel_re"], tensors["cqt_kernel_im"] = re, im
    tensors["cqt_lowpass"], tensors["cqt_sqrt_len"] = low, sq[0].reshape(309)

    # NormalizedLog constants (Add eps -> Log -> Mul 1/ln10 -> Mul 10) and the folded BatchNorm affine (Mul -> Add)
    log_node = [n for n in nodes if n["op_type"] == "Log"]
    _require(len(log_node) == 1, "expected one Log node")
    eps = [inits[i] for i in by_out[log_node[0]["input"][0]]["input"] if i in inits]
    _require(len(eps) == 1, "log epsilon not found")
    tensors["log_eps"] = eps[0].reshape(1)
    mul1 = consumers(log_node[0]["output"][0])[0]
    c1 = [inits[i] for i in mul1["input"] if i in inits][0]
    mul2 = consumers(mul1["output"][0])[0]
    c2 = [inits[i] for i in mul2["input"] if i in inits][0]
    tensors["log_scale"] = np.asarray([c1.reshape(()), c2.reshape(())], dtype=np.float32)
    bn_mul = [n for n in nodes if n["op_type"] == "Mul" and "batch_normalization/FusedBatchNormV3" in n["output"][0]]
    _require(len(bn_mul) == 1, "folded BatchNorm scale not found")
    bn_scale = [inits[i] for i in bn_mul[0]["input"] if i in inits][0]
    bn_add = consumers(bn_mul[0]["output"][0])[0]
    _require(bn_add["op_type"] == "Add", "folded BatchNorm shift not found")
    bn_shift = [inits[i] for i in bn_add["input"] if i in inits][0]
    tensors["bn_affine"] = np.asarray([bn_scale.reshape(()), bn_shift.reshape(())], dtype=np.float32)

    for name, n in zip(CNN_ORDER, cnn_convs):
        w, b = inits[n["input"][1]], inits[n["input"][2]]
        _require(tuple(w.shape) == CNN_SHAPES[name] and b.shape == (w.shape[0],), f"{name}: weight shape {w.shape}")
        tensors[name + "_w"] = w.astype(np.float32)
        tensors[name + "_b"] = b.astype(np.float32)
    return tensors


def pack_blob(tensors: Dict[str, np.ndarray]) -> bytes:
    entries, data, off = b"", b"", 0
    for name, t in tensors.items():
        arr = np.ascontiguousarray(t, dtype="<f4")
        dims = list(arr.shape) + [1] * (4 - arr.ndim)
        entries += struct.pack("<24sI4III", name.encode(), arr.ndim, *dims, off, arr.size)
        data += arr.tobytes()
        off += arr.size
    return MAGIC + struct.pack("<II", 1, len(tensors)) + entries + data


def load_model_blob(model_path: Union[str, pathlib.Path]) -> bytes:
    """Bytes for `bp_create` from whatever `Model(model_path)` was given."""
    p = pathlib.Path(model_path)
    if p.is_dir() or p.suffix in (".tflite", ".mlpackage"):  # another artifact of the same model: use its nmp.onnx
        sibling = p.with_name(p.name.split(".")[0] + ".onnx")
        if not sibling.is_file():
            raise ValueError(
                f"File {model_path} cannot be loaded: the MI355X backend reads the weights blob or the ONNX artifact, "
                f"and no {sibling.name} lies next to it"
            )
        p = sibling
    try:
        data = p.read_bytes()
    except OSError as e:
        raise ValueError(f"File {model_path} cannot be loaded: {e}") from e
    if data[:8] == MAGIC:
        return data
    try:
        return pack_blob(tensors_from_onnx(data))
    except ValueError as e:
        raise ValueError(f"File {model_path} cannot be loaded into the MI355X backend: {e}") from e
