#!/usr/bin/env python
"""Dev-time asset builder: frozen-graph constants -> basic_pitch_amd/assets/nmp_weights.bin.

Reads the reference's serialized model (`basic_pitch/saved_models/icassp_2022/nmp.onnx`,
SURVEY.md App. A) with the wire reader in basic_pitch_amd/onnx_wire.py (through basic_pitch_amd/weights.py) and writes the 18 tensors the hot
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
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from basic_pitch_amd.weights import pack_blob, tensors_from_onnx  # noqa: E402


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

    # the same extraction `Model("nmp.onnx")` performs at load time (basic_pitch_amd/weights.py) ...
    tensors = tensors_from_onnx(args.onnx)
    # ... pinned to the reference's SOURCE as well: the four CQT constants regenerated from nnaudio.py's formulas
    re_f, im_f, lp_f, sq_f = regenerate_cqt_constants()
    assert np.array_equal(tensors["cqt_kernel_re"], re_f), "real CQT bank differs from the nnaudio.py formulas"
    assert np.array_equal(tensors["cqt_kernel_im"], im_f), "imaginary CQT bank differs from the nnaudio.py formulas"
    assert np.array_equal(tensors["cqt_lowpass"], lp_f), "lowpass differs from firwin2 regeneration"
    assert np.array_equal(tensors["cqt_sqrt_len"], sq_f), "sqrt(lengths) differs from regeneration"
    names = list(tensors.keys())
    blob = pack_blob(tensors)
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
