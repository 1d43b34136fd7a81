"""Host-side mirror of the reference's inference interface for the hot path.

Same names, argument meaning and error behaviour as `basic_pitch/inference.py` (spotify/basic-pitch
v0.4.0): `Model` (71-182), `window_audio_file` (194-219), `get_audio_input` (222-244),
`unwrap_output` (247-279), `run_inference` (282-330), `predict` (431-506).  The arithmetic runs in
libbasicpitch_amd.so (hand-written HIP for gfx950) — there is no CPU execution path here.

Differences by design (the GPU needs batches; the reference loops batch-1):
  * `Model.predict(x)` takes any n >= 0 windows per call (the frozen graph's batch dim is dynamic);
  * `run_inference` sends the whole track to the device once (`bp_infer_track`: windowing and
    un-overlapping happen on the GPU) instead of calling predict once per window.
Both produce the values the per-window loop would.
"""
from __future__ import annotations

import ctypes as C
import enum
import json
import pathlib
from typing import Any, Dict, Iterable, List, Optional, Tuple, Union

import numpy as np

from . import _native
from . import audio as _audio
from .constants import (
    ANNOTATIONS_FPS,
    AUDIO_N_SAMPLES,
    AUDIO_SAMPLE_RATE,
    AUDIO_WINDOW_LENGTH,
    FFT_HOP,
    ANNOT_N_FRAMES,
    N_FREQ_BINS_CONTOURS,
    N_FREQ_BINS_NOTES,
)

DEFAULT_ONSET_THRESHOLD = 0.5
DEFAULT_FRAME_THRESHOLD = 0.3
DEFAULT_MINIMUM_NOTE_LENGTH_MS = 127.7
DEFAULT_MINIMUM_MIDI_TEMPO = 120
DEFAULT_SONIFICATION_SAMPLERATE = 44100
DEFAULT_OVERLAPPING_FRAMES = 30
DEFAULT_MIDI_VELOCITY_SCALE = 127

PKG_DIR = pathlib.Path(__file__).parent
ICASSP_2022_MODEL_PATH = PKG_DIR / "assets" / "nmp_weights.bin"


def _is_torch_cuda(x: Any) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "is_cuda") and bool(x.is_cuda)


class Model:
    """Drop-in for `basic_pitch.inference.Model`: load a serialized model, `predict(x) -> dict`.

    `model_path` is the weights blob built from the reference's `nmp.onnx` by
    tools/extract_weights.py (`ICASSP_2022_MODEL_PATH`).  Like the reference (inference.py:148-154)
    an unloadable file raises ValueError; a missing HIP library / GPU raises NativeLibraryError.
    """

    class MODEL_TYPES(enum.Enum):
        MI355X_HIP = enum.auto()

    def __init__(
        self,
        model_path: Union[pathlib.Path, str] = ICASSP_2022_MODEL_PATH,
        device: int = 0,
        max_windows: int = 256,
        stage_timing: bool = False,
        exact_f32_mfma: bool = False,
    ):
        self.model_type = Model.MODEL_TYPES.MI355X_HIP
        self._lib = _native.load_library()
        self._handle = C.c_void_p()
        try:
            blob = pathlib.Path(model_path).read_bytes()
        except OSError as e:
            raise ValueError(f"File {model_path} cannot be loaded: {e}") from e
        flags = _native.BP_FLAG_STAGE_TIMING if stage_timing else 0
        if exact_f32_mfma:  # A/B reference: contour conv1 on the exact-f32 MFMA kernel
            flags |= _native.BP_FLAG_F32_MFMA
        rc = self._lib.bp_create(blob, len(blob), int(device), flags, int(max_windows), C.byref(self._handle))
        if rc != _native.BP_OK:
            self._handle = C.c_void_p()
            _native.check(self._lib, None, rc, f"File {model_path} cannot be loaded into the MI355X backend")
        self.device = int(device)
        self.max_windows = int(max_windows)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_handle", None) is not None and self._handle.value:
            self._lib.bp_destroy(self._handle)
            self._handle = C.c_void_p()

    def __del__(self) -> None:  # pragma: no cover - interpreter shutdown order
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self) -> "Model":
        return self

    def __exit__(self, *exc: Any) -> None:
        self.close()

    # -- inference --------------------------------------------------------------------------------
    def predict(self, x: Any) -> Dict[str, Any]:
        """x: float32 [n, 43844] or [n, 43844, 1] -> {"note","onset","contour"} (inference.py:156-182).

        numpy in -> fresh, writable, C-contiguous numpy arrays out (the reference's consumer mutates
        them, note_creation.py:338-341).  torch CUDA tensor in -> torch CUDA tensors out (zero copy).
        """
        if _is_torch_cuda(x):
            return self._predict_device(x)
        x = np.asarray(x)
        if x.ndim == 3 and x.shape[2] == 1:
            x = x[:, :, 0]
        if x.ndim != 2 or x.shape[1] != AUDIO_N_SAMPLES:
            raise ValueError(f"expected input of shape (n, {AUDIO_N_SAMPLES}[, 1]), got {x.shape}")
        x = np.ascontiguousarray(x, dtype=np.float32)
        n = x.shape[0]
        out = {
            "note": np.empty((n, ANNOT_N_FRAMES, N_FREQ_BINS_NOTES), dtype=np.float32),
            "onset": np.empty((n, ANNOT_N_FRAMES, N_FREQ_BINS_NOTES), dtype=np.float32),
            "contour": np.empty((n, ANNOT_N_FRAMES, N_FREQ_BINS_CONTOURS), dtype=np.float32),
        }
        if n:
            rc = self._lib.bp_infer(
                self._handle,
                x.ctypes.data,
                n,
                out["note"].ctypes.data,
                out["onset"].ctypes.data,
                out["contour"].ctypes.data,
                _native.BP_MEM_HOST,
            )
            _native.check(self._lib, self._handle, rc, "bp_infer")
        return out

    def _predict_device(self, x: Any, out: Optional[Dict[str, Any]] = None, sync: bool = True) -> Dict[str, Any]:
        import torch

        if x.dim() == 3 and x.shape[2] == 1:
            x = x[:, :, 0]
        if x.dim() != 2 or x.shape[1] != AUDIO_N_SAMPLES or x.dtype != torch.float32:
            raise ValueError(f"expected float32 CUDA tensor of shape (n, {AUDIO_N_SAMPLES}[, 1]), got {tuple(x.shape)}")
        if x.device.index != self.device:
            raise ValueError(f"input lives on cuda:{x.device.index}, model on cuda:{self.device}")
        x = x.contiguous()
        n = x.shape[0]
        if out is None:
            out = {
                "note": torch.empty((n, ANNOT_N_FRAMES, N_FREQ_BINS_NOTES), dtype=torch.float32, device=x.device),
                "onset": torch.empty((n, ANNOT_N_FRAMES, N_FREQ_BINS_NOTES), dtype=torch.float32, device=x.device),
                "contour": torch.empty((n, ANNOT_N_FRAMES, N_FREQ_BINS_CONTOURS), dtype=torch.float32, device=x.device),
            }
        if n:
            self._lib.bp_set_stream(self._handle, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream))
            rc = self._lib.bp_infer_async(
                self._handle, x.data_ptr(), n, out["note"].data_ptr(), out["onset"].data_ptr(), out["contour"].data_ptr()
            )
            _native.check(self._lib, self._handle, rc, "bp_infer_async")
            if sync:
                _native.check(self._lib, self._handle, self._lib.bp_synchronize(self._handle), "bp_synchronize")
        return out

    def predict_track(self, samples: np.ndarray) -> Dict[str, np.ndarray]:
        """Whole mono 22.05 kHz track -> un-overlapped posteriorgrams (inference.py:282-315 on device)."""
        samples = np.ascontiguousarray(samples, dtype=np.float32)
        if samples.ndim != 1:
            raise ValueError("predict_track expects a 1-D mono signal")
        n = samples.shape[0]
        T = int(self._lib.bp_track_n_frames(n))
        out = {
            "note": np.empty((T, N_FREQ_BINS_NOTES), dtype=np.float32),
            "onset": np.empty((T, N_FREQ_BINS_NOTES), dtype=np.float32),
            "contour": np.empty((T, N_FREQ_BINS_CONTOURS), dtype=np.float32),
        }
        rc = self._lib.bp_infer_track(
            self._handle,
            samples.ctypes.data,
            n,
            out["note"].ctypes.data,
            out["onset"].ctypes.data,
            out["contour"].ctypes.data,
            _native.BP_MEM_HOST,
        )
        _native.check(self._lib, self._handle, rc, "bp_infer_track")
        return out

    # -- introspection ----------------------------------------------------------------------------
    def info(self) -> Dict[str, Any]:
        inf = _native.bp_info()
        _native.check(self._lib, self._handle, self._lib.bp_get_info(self._handle, C.byref(inf)), "bp_get_info")
        return {
            "device": inf.device_ordinal,
            "compute_units": inf.compute_units,
            "max_windows": inf.max_windows,
            "workspace_bytes": inf.workspace_bytes,
            "arch": inf.arch.decode(),
        }

    def stage_ms(self) -> Dict[str, float]:
        ms = (C.c_float * _native.BP_N_STAGES)()
        _native.check(self._lib, self._handle, self._lib.bp_get_stage_ms(self._handle, ms, _native.BP_N_STAGES), "bp_get_stage_ms")
        return {k: float(v) for k, v in zip(_native.STAGE_NAMES, ms)}


def window_audio_file(
    audio_original: np.ndarray, hop_size: int
) -> Iterable[Tuple[np.ndarray, Dict[str, float]]]:
    """Pad and window an audio signal into AUDIO_N_SAMPLES chunks (inference.py:194-219)."""
    for i in range(0, audio_original.shape[0], hop_size):
        window = audio_original[i : i + AUDIO_N_SAMPLES]
        if len(window) < AUDIO_N_SAMPLES:
            window = np.pad(window, pad_width=[[0, AUDIO_N_SAMPLES - len(window)]])
        t_start = float(i) / AUDIO_SAMPLE_RATE
        window_time = {"start": t_start, "end": t_start + (AUDIO_N_SAMPLES / AUDIO_SAMPLE_RATE)}
        yield np.expand_dims(window, axis=-1), window_time


def get_audio_input(
    audio_path: Union[pathlib.Path, str], overlap_len: int, hop_size: int
) -> Iterable[Tuple[np.ndarray, Dict[str, float], int]]:
    """Read a file as mono 22.05 kHz, prepend overlap_len/2 zeros, yield windows (inference.py:222-244)."""
    assert overlap_len % 2 == 0, f"overlap_length must be even, got {overlap_len}"
    audio_original, _ = _audio.load(str(audio_path), sr=AUDIO_SAMPLE_RATE, mono=True)
    original_length = audio_original.shape[0]
    audio_original = np.concatenate([np.zeros((int(overlap_len / 2),), dtype=np.float32), audio_original])
    for window, window_time in window_audio_file(audio_original, hop_size):
        yield np.expand_dims(window, axis=0), window_time, original_length


def unwrap_output(
    output: np.ndarray, audio_original_length: int, n_overlapping_frames: int, hop_size: int
) -> Optional[np.ndarray]:
    """Unwrap batched model predictions to a single matrix (inference.py:247-279)."""
    if len(output.shape) != 3:
        return None
    n_olap = int(0.5 * n_overlapping_frames)
    if n_olap > 0:
        output = output[:, n_olap:-n_olap, :]
    output_shape = output.shape
    unwrapped_output = output.reshape(output_shape[0] * output_shape[1], output_shape[2])
    n_expected_windows = audio_original_length / hop_size
    n_frames_per_window = (AUDIO_WINDOW_LENGTH * ANNOTATIONS_FPS) - n_overlapping_frames
    return unwrapped_output[: int(n_expected_windows * n_frames_per_window), :]


def run_inference(
    audio_path: Union[pathlib.Path, str],
    model_or_model_path: Union[Model, pathlib.Path, str] = ICASSP_2022_MODEL_PATH,
    debug_file: Optional[pathlib.Path] = None,
) -> Dict[str, np.ndarray]:
    """Run the model on the input audio path (inference.py:282-330).

    Returns {"note": (T,88), "onset": (T,88), "contour": (T,264)} float32, T = int(L/36164*142).
    """
    model = model_or_model_path if isinstance(model_or_model_path, Model) else Model(model_or_model_path)
    n_overlapping_frames = DEFAULT_OVERLAPPING_FRAMES
    overlap_len = n_overlapping_frames * FFT_HOP
    hop_size = AUDIO_N_SAMPLES - overlap_len

    audio_original, _ = _audio.load(str(audio_path), sr=AUDIO_SAMPLE_RATE, mono=True)
    audio_original_length = audio_original.shape[0]
    unwrapped_output = model.predict_track(audio_original)

    if debug_file:
        with open(debug_file, "w") as f:
            json.dump(
                {
                    "audio_original_length": audio_original_length,
                    "hop_size_samples": hop_size,
                    "overlap_length_samples": overlap_len,
                    "unwrapped_output": {k: v.tolist() for k, v in unwrapped_output.items()},
                },
                f,
            )
    return unwrapped_output


def run_inference_windowed(
    audio_path: Union[pathlib.Path, str], model: Model, batch: int = 64
) -> Dict[str, np.ndarray]:
    """The reference's own structure (host windowing -> Model.predict -> host unwrap), batched.

    Used by the tests to show that the on-device track path equals the per-window path.
    """
    n_overlapping_frames = DEFAULT_OVERLAPPING_FRAMES
    overlap_len = n_overlapping_frames * FFT_HOP
    hop_size = AUDIO_N_SAMPLES - overlap_len
    windows: List[np.ndarray] = []
    audio_original_length = 0
    for audio_windowed, _, audio_original_length in get_audio_input(audio_path, overlap_len, hop_size):
        windows.append(audio_windowed[0])
    output: Dict[str, List[np.ndarray]] = {"note": [], "onset": [], "contour": []}
    for i in range(0, len(windows), batch):
        for k, v in model.predict(np.stack(windows[i : i + batch])).items():
            output[k].append(v)
    return {
        k: unwrap_output(np.concatenate(output[k]), audio_original_length, n_overlapping_frames, hop_size)
        for k in output
    }
