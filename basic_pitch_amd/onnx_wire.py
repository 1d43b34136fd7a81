"""Minimal protobuf wire-format reader for ONNX files (no `onnx` package needed).

Used by basic_pitch_amd/weights.py to pull the 18 constants of the frozen Basic Pitch graph out of the
reference's serialized model `saved_models/icassp_2022/nmp.onnx` when `Model(path)` is given that file
(basic_pitch/inference.py:78-154 takes the serialized model's path), and by tools/extract_weights.py.  Field numbers follow the public
onnx.proto3 schema (ModelProto.graph=7, GraphProto.node=1/initializer=5, NodeProto.input=1/
output=2/name=3/op_type=4/attribute=5, TensorProto.dims=1/data_type=2/name=8/raw_data=9, ...).
"""
from __future__ import annotations

import struct
from typing import Dict, Iterator, List, Tuple

import numpy as np


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def fields(buf: bytes) -> Iterator[Tuple[int, int, object]]:
    """Yield (field_number, wire_type, value) for every field of one message."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos : pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos : pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos : pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fno, wt, val


def _packed_varints(val: bytes) -> List[int]:
    out = []
    pos = 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(v)
    return out


_DTYPES = {1: np.float32, 6: np.int32, 7: np.int64, 11: np.float64}


def parse_tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype = 1
    name = ""
    raw = None
    float_data: List[float] = []
    int64_data: List[int] = []
    int32_data: List[int] = []
    for fno, wt, val in fields(buf):
        if fno == 1:
            dims.extend(_packed_varints(val) if wt == 2 else [val])
        elif fno == 2:
            dtype = val
        elif fno == 8:
            name = val.decode("utf-8", "replace")
        elif fno == 9:
            raw = val
        elif fno == 4:
            if wt == 2:
                float_data.extend(struct.unpack(f"<{len(val)//4}f", val))
            else:
                float_data.append(struct.unpack("<f", val)[0])
        elif fno == 7:
            int64_data.extend(_packed_varints(val) if wt == 2 else [val])
        elif fno == 5:
            int32_data.extend(_packed_varints(val) if wt == 2 else [val])
    np_dtype = _DTYPES.get(dtype)
    if np_dtype is None:
        return name, np.zeros(0)
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_dtype).newbyteorder("<")).astype(np_dtype)
    elif float_data:
        arr = np.asarray(float_data, dtype=np_dtype)
    elif int64_data:
        arr = np.asarray(int64_data, dtype=np_dtype)
    else:
        arr = np.asarray(int32_data, dtype=np_dtype)
    return name, arr.reshape(dims) if dims else arr.reshape(())


def parse_attribute(buf: bytes) -> Tuple[str, object]:
    name = ""
    out: Dict[str, object] = {}
    ints: List[int] = []
    floats: List[float] = []
    for fno, wt, val in fields(buf):
        if fno == 1:
            name = val.decode()
        elif fno == 2:
            out["f"] = struct.unpack("<f", val)[0]
        elif fno == 3:
            out["i"] = val
        elif fno == 4:
            out["s"] = val
        elif fno == 5:
            out["t"] = parse_tensor(val)[1]
        elif fno == 7:
            if wt == 2:
                floats.extend(struct.unpack(f"<{len(val)//4}f", val))
            else:
                floats.append(struct.unpack("<f", val)[0])
        elif fno == 8:
            ints.extend(_packed_varints(val) if wt == 2 else [val])
    if ints:
        return name, ints
    if floats:
        return name, floats
    for k in ("t", "s", "i", "f"):
        if k in out:
            return name, out[k]
    return name, None


def parse_node(buf: bytes) -> dict:
    node = {"input": [], "output": [], "name": "", "op_type": "", "attr": {}}
    for fno, wt, val in fields(buf):
        if fno == 1:
            node["input"].append(val.decode())
        elif fno == 2:
            node["output"].append(val.decode())
        elif fno == 3:
            node["name"] = val.decode()
        elif fno == 4:
            node["op_type"] = val.decode()
        elif fno == 5:
            k, v = parse_attribute(val)
            node["attr"][k] = v
    return node


def load_graph(path_or_bytes) -> Tuple[List[dict], Dict[str, np.ndarray]]:
    """Return (nodes in graph order, {initializer name: array})."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        model = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            model = f.read()
    graph = None
    for fno, wt, val in fields(model):
        if fno == 7:
            graph = val
    if graph is None:
        raise ValueError("no GraphProto in file")
    nodes: List[dict] = []
    inits: Dict[str, np.ndarray] = {}
    for fno, wt, val in fields(graph):
        if fno == 1:
            nodes.append(parse_node(val))
        elif fno == 5:
            name, arr = parse_tensor(val)
            inits[name] = arr
    return nodes, inits
