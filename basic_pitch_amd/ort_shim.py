"""An `onnxruntime`-shaped front for the MI355X backend: the zero-patch seam into the unmodified reference.

The reference's ONNX leg (basic_pitch/inference.py:130-146, 168-182) needs exactly three things from `onnxruntime`:

    ort.get_available_providers()
    ort.InferenceSession(str(model_path), providers=[...])
    session.run([out_name, ...], {"serving_default_input_2:0": x})      # x float32 [n, 43844, 1]

`install()` registers this module as `onnxruntime` in `sys.modules` (explicit opt-in, never done on import), after
which `basic_pitch.inference.Model(ICASSP_2022_MODEL_PATH)` loads `nmp.onnx` "through" libbasicpitch_amd.so and the
reference's `predict()` / `note_creation` run unchanged on top of it.  The session does not interpret
the ONNX graph op by op: it checks that the file has the structure of the reference's
`saved_models/icassp_2022/nmp.onnx`, extracts its 18 constants (basic_pitch_amd/weights.py) and runs the hand-written
kernels built for that graph with them.  Any other model file is refused, like an unloadable model in the reference
(ValueError).  `NMP_ONNX_SHA256` is the digest of the v0.4.0 artifact (`session.is_reference_artifact`).
"""
from __future__ import annotations

import hashlib
import sys
from typing import Dict, List, Optional, Sequence

import numpy as np

from .weights import MAGIC, NMP_ONNX_SHA256, pack_blob, tensors_from_onnx  # noqa: E402

PROVIDER = "MI355XExecutionProvider"
INPUT_NAME = "serving_default_input_2:0"  # inference.py:178
# ONNX output name -> key of the reference's output dict (inference.py:168-182)
OUTPUT_KEYS = {"StatefulPartitionedCall:1": "note", "StatefulPartitionedCall:2": "onset", "StatefulPartitionedCall:0": "contour"}

__version__ = "0.0-basic_pitch_amd"


def get_available_providers() -> List[str]:
    return [PROVIDER]


def get_device() -> str:
    return "GPU"


class _Arg:
    def __init__(self, name: str, shape, type_: str = "tensor(float)"):
        self.name, self.shape, self.type = name, shape, type_


class InferenceSession:
    def __init__(self, path_or_bytes, sess_options=None, providers: Optional[Sequence[str]] = None, device: int = 0,
                 max_windows: int = 256, **_ignored):
        from .inference import Model  # the ctypes binding; raises NativeLibraryError without the library / GPU

        import os
        import tempfile

        data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(str(path_or_bytes), "rb").read()
        self.is_reference_artifact = hashlib.sha256(data).hexdigest() == NMP_ONNX_SHA256
        # ValueError unless the bytes are the Basic Pitch graph (or this package's pre-extracted weights blob)
        blob = bytes(data) if data[:8] == MAGIC else pack_blob(tensors_from_onnx(bytes(data)))
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
            f.write(blob)
        try:
            self._model = Model(f.name, device=device, max_windows=max_windows)
        finally:
            os.unlink(f.name)
        self._providers = list(providers) if providers else [PROVIDER]

    def get_providers(self) -> List[str]:
        return self._providers

    def get_inputs(self):
        return [_Arg(INPUT_NAME, ["unk__749", 43844, 1])]

    def get_outputs(self):
        return [_Arg("StatefulPartitionedCall:0", ["unk__750", 172, 264]), _Arg("StatefulPartitionedCall:1", ["unk__751", 172, 88]),
                _Arg("StatefulPartitionedCall:2", ["unk__752", 172, 88])]

    def run(self, output_names: Optional[Sequence[str]], input_feed: Dict[str, np.ndarray], run_options=None) -> List[np.ndarray]:
        if set(input_feed) != {INPUT_NAME}:
            raise ValueError(f"expected exactly the input {INPUT_NAME!r}, got {sorted(input_feed)}")
        names = list(output_names) if output_names else [a.name for a in self.get_outputs()]
        unknown = [n for n in names if n not in OUTPUT_KEYS]
        if unknown:
            raise ValueError(f"unknown output name(s) {unknown}")
        out = self._model.predict(np.asarray(input_feed[INPUT_NAME], dtype=np.float32))
        return [out[OUTPUT_KEYS[n]] for n in names]


def install() -> None:
    """Make `import onnxruntime` resolve to this module (only if no real onnxruntime is already imported)."""
    mod = sys.modules[__name__]
    existing = sys.modules.get("onnxruntime")
    if existing is not None and existing is not mod:
        raise RuntimeError("a different `onnxruntime` is already imported; refusing to shadow it")
    sys.modules["onnxruntime"] = mod
