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

import csv
import ctypes as C
import enum
import json
import os
import pathlib
import threading
from typing import Any, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import numpy as np

from . import _native
from . import audio as _audio
from . import note_creation as infer
from . import weights as _weights
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

    `model_path` is the serialized model, as in the reference: its `saved_models/icassp_2022/nmp.onnx` (the 18
    constants are extracted at load, basic_pitch_amd/weights.py; the `nmp/`, `nmp.tflite` and `nmp.mlpackage`
    artifacts resolve to the `nmp.onnx` next to them), or the pre-extracted blob shipped with this package
    (`ICASSP_2022_MODEL_PATH`, the default).  Like the reference (inference.py:148-154) an unloadable file raises
    ValueError; a missing HIP library / GPU raises NativeLibraryError.
    """

    class MODEL_TYPES(enum.Enum):
        MI355X_HIP = enum.auto()

    def __init__(
        self,
        model_path: Union[pathlib.Path, str] = ICASSP_2022_MODEL_PATH,
        device: int = 0,
        max_windows: int = 256,
        stage_timing: bool = False,
        time_dominant: bool = False,
        exact_f32_mfma: bool = False,
        bf16_weights: bool = False,
        ext_cqt_44k: bool = False,
        f16_corrections: bool = False,
        fp8_corrections: bool = False,
        blocking_wait: bool = False,
    ):
        self.model_type = Model.MODEL_TYPES.MI355X_HIP
        self._lib = _native.load_library()
        self._handle = C.c_void_p()
        blob = _weights.load_model_blob(model_path)  # ValueError if unloadable, like inference.py:148-154
        flags = _native.BP_FLAG_STAGE_TIMING if stage_timing else 0
        if time_dominant:  # HIP events around the dominant kernel only (2 records per chunk instead of 16)
            flags |= _native.BP_FLAG_TIME_DOMINANT
        if exact_f32_mfma:  # A/B reference: contour conv1 on the exact-f32 MFMA kernel
            flags |= _native.BP_FLAG_F32_MFMA
        if bf16_weights:  # BASELINE.json configs[3]: conv weights rounded to bf16, 2 matrix instructions per product
            flags |= _native.BP_FLAG_BF16_WEIGHTS
        if ext_cqt_44k:  # BASELINE.json configs[4]: 44.1 kHz windows of 87,688 samples, 10-octave / 345-bin CQT
            flags |= _native.BP_FLAG_EXT_CQT_44K
        if f16_corrections:  # the default arithmetic since round 3 (all three split-precision products on f16): a no-op
            flags |= _native.BP_FLAG_F16_CORRECTIONS
        if fp8_corrections:  # A/B library only since round 6 (the product library refuses the flag: ValueError)
            flags |= _native.BP_FLAG_FP8_CORRECTIONS
        if blocking_wait:  # whole-track calls sleep on an interrupt instead of spinning (file jobs: workers share cores)
            flags |= _native.BP_FLAG_BLOCKING_WAIT
        rc = self._lib.bp_create(blob, len(blob), int(device), flags, int(max_windows), C.byref(self._handle))
        if rc != _native.BP_OK:
            self._handle = C.c_void_p()
            _native.check(self._lib, None, rc, f"File {model_path} cannot be loaded into the MI355X backend")
        self.device = int(device)
        self.max_windows = int(max_windows)
        # window length / sample rate of this handle's geometry (43844 @ 22050 unless ext_cqt_44k)
        self.audio_n_samples = int(self._lib.bp_handle_window_samples(self._handle))
        self.sample_rate = int(self._lib.bp_handle_sample_rate(self._handle))

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
        if x.ndim != 2 or x.shape[1] != self.audio_n_samples:
            raise ValueError(f"expected input of shape (n, {self.audio_n_samples}[, 1]), got {x.shape}")
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
        if x.dim() != 2 or x.shape[1] != self.audio_n_samples or x.dtype != torch.float32:
            raise ValueError(f"expected float32 CUDA tensor of shape (n, {self.audio_n_samples}[, 1]), got {tuple(x.shape)}")
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
        T = int(self._lib.bp_handle_track_n_frames(self._handle, n))
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

    def predict_tracks(self, tracks: "List[Any]") -> "List[Dict[str, Any]]":
        """Several mono 22.05 kHz tracks in one call, windows packed across track boundaries into full batches
        (bp_infer_tracks).  All numpy arrays (host in / host out) or all 1-D float32 CUDA tensors (device in / out)."""
        n = len(tracks)
        if n == 0:
            return []
        on_device = all(hasattr(t, "is_cuda") and t.is_cuda for t in tracks)
        if on_device:
            import torch

            tr = [t.contiguous() for t in tracks]
            if any(t.dtype != torch.float32 or t.dim() != 1 for t in tr):
                raise ValueError("expected 1-D float32 CUDA tensors")
            lens = [int(t.shape[0]) for t in tr]
            outs = [
                {k: torch.empty((int(self._lib.bp_handle_track_n_frames(self._handle, L)), w), dtype=torch.float32, device=tr[0].device)
                 for k, w in (("note", N_FREQ_BINS_NOTES), ("onset", N_FREQ_BINS_NOTES), ("contour", N_FREQ_BINS_CONTOURS))}
                for L in lens
            ]
            ptr = lambda a: a.data_ptr()  # noqa: E731
            kind = _native.BP_MEM_DEVICE
            self._lib.bp_set_stream(self._handle, C.c_void_p(torch.cuda.current_stream(tr[0].device).cuda_stream))
        else:
            tr = [np.ascontiguousarray(t, dtype=np.float32) for t in tracks]
            if any(t.ndim != 1 for t in tr):
                raise ValueError("predict_tracks expects 1-D mono signals")
            lens = [int(t.shape[0]) for t in tr]
            outs = [
                {k: np.empty((int(self._lib.bp_handle_track_n_frames(self._handle, L)), w), dtype=np.float32)
                 for k, w in (("note", N_FREQ_BINS_NOTES), ("onset", N_FREQ_BINS_NOTES), ("contour", N_FREQ_BINS_CONTOURS))}
                for L in lens
            ]
            ptr = lambda a: a.ctypes.data  # noqa: E731
            kind = _native.BP_MEM_HOST
        arr = lambda vals: (C.c_void_p * n)(*[C.c_void_p(v) for v in vals])  # noqa: E731
        rc = self._lib.bp_infer_tracks(
            self._handle, n, arr([ptr(t) for t in tr]), (C.c_int64 * n)(*lens),
            arr([ptr(o["note"]) for o in outs]), arr([ptr(o["onset"]) for o in outs]),
            arr([ptr(o["contour"]) for o in outs]), kind,
        )
        _native.check(self._lib, self._handle, rc, "bp_infer_tracks")
        return outs

    def _predict_track_device(self, samples: "Any", out: Optional[Dict[str, "Any"]] = None) -> Dict[str, "Any"]:
        """Device-resident mono 22.05 kHz track (1-D float32 CUDA tensor) -> device posteriorgrams (zero copy)."""
        import torch

        if not (samples.is_cuda and samples.dtype == torch.float32 and samples.dim() == 1):
            raise ValueError("expected a 1-D float32 CUDA tensor")
        samples = samples.contiguous()
        n = int(samples.shape[0])
        T = int(self._lib.bp_handle_track_n_frames(self._handle, n))
        if out is None:
            out = {
                "note": torch.empty((T, N_FREQ_BINS_NOTES), dtype=torch.float32, device=samples.device),
                "onset": torch.empty((T, N_FREQ_BINS_NOTES), dtype=torch.float32, device=samples.device),
                "contour": torch.empty((T, N_FREQ_BINS_CONTOURS), dtype=torch.float32, device=samples.device),
            }
        self._lib.bp_set_stream(self._handle, C.c_void_p(torch.cuda.current_stream(samples.device).cuda_stream))
        rc = self._lib.bp_infer_track(
            self._handle, samples.data_ptr(), n, out["note"].data_ptr(), out["onset"].data_ptr(),
            out["contour"].data_ptr(), _native.BP_MEM_DEVICE,
        )
        _native.check(self._lib, self._handle, rc, "bp_infer_track")
        return out

    @staticmethod
    def _as_pcm(pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        if pcm.ndim == 1:
            pcm = pcm[:, None]
        if pcm.ndim != 2 or pcm.shape[1] < 1:
            raise ValueError("expected PCM as float32 [n_frames] or [n_frames, channels]")
        return pcm

    def resample(self, pcm: np.ndarray, sample_rate: int) -> np.ndarray:
        """Decoded PCM [n_frames(, channels)] at `sample_rate` -> mono 22.05 kHz float32, computed on the device
        (channel mean + polyphase FIR; the `librosa.load(..., sr=22050, mono=True)` tail of inference.py:239)."""
        pcm = self._as_pcm(pcm)
        n_out = int(self._lib.bp_handle_resampled_length(self._handle, pcm.shape[0], int(sample_rate)))
        out = np.empty((n_out,), dtype=np.float32)
        rc = self._lib.bp_resample(
            self._handle, pcm.ctypes.data, pcm.shape[0], pcm.shape[1], int(sample_rate), out.ctypes.data, _native.BP_MEM_HOST
        )
        _native.check(self._lib, self._handle, rc, "bp_resample")
        return out

    def note_candidates(self, output: Dict[str, Any], prm) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray], int]:
        """The dense half of note decoding on the device (`bp_note_candidates`, csrc/note_device.hip) for the posteriorgrams
        `output` (numpy arrays or CUDA tensors of T frames): (note map after constrain_frequency, onset-peak bitmap
        (T, 12) uint8, pitch-bend map (T, 88) int8 or None, status).  status 1: a NaN in the maps or an onset threshold
        <= 0 — decode the maps themselves (`note_creation.model_output_to_notes`).  `prm`: `note_creation._note_params`."""
        on_dev = _is_torch_cuda(output["note"])
        maps = {}
        for k, w in (("note", N_FREQ_BINS_NOTES), ("onset", N_FREQ_BINS_NOTES), ("contour", N_FREQ_BINS_CONTOURS)):
            a = output[k]
            if on_dev:
                a = a.contiguous().float()
            else:
                a = np.require(a, np.float32, ["C"])
            if a.ndim != 2 or a.shape[1] != w:
                raise ValueError(f"{k}: expected (T, {w})")
            maps[k] = a
        T = int(maps["note"].shape[0])
        note = np.empty((T, N_FREQ_BINS_NOTES), np.float32)
        bits = np.empty((T, 12), np.uint8)
        bend = np.empty((T, N_FREQ_BINS_NOTES), np.int8) if prm.include_pitch_bends else None
        status = C.c_int(0)
        ptr = (lambda a: a.data_ptr()) if on_dev else (lambda a: a.ctypes.data)
        if on_dev:
            import torch

            torch.cuda.current_stream(maps["note"].device).synchronize()
        rc = self._lib.bp_note_candidates(
            self._handle, ptr(maps["note"]), ptr(maps["onset"]), ptr(maps["contour"]), T, C.byref(prm),
            _native.BP_MEM_DEVICE if on_dev else _native.BP_MEM_HOST, note.ctypes.data, bits.ctypes.data,
            bend.ctypes.data if bend is not None else None, C.byref(status),
        )
        _native.check(self._lib, self._handle, rc, "bp_note_candidates")
        return note, bits, bend, int(status.value)

    def predict_pcm(self, pcm: np.ndarray, sample_rate: int) -> Dict[str, np.ndarray]:
        """Decoded PCM at any rate / channel count -> un-overlapped posteriorgrams: downmix, resampling, windowing,
        CQT + CNN and un-overlapping all on the device (bp_infer_pcm)."""
        pcm = self._as_pcm(pcm)
        n22 = int(self._lib.bp_handle_resampled_length(self._handle, pcm.shape[0], int(sample_rate)))
        T = int(self._lib.bp_handle_track_n_frames(self._handle, n22))
        out = {
            "note": np.empty((T, N_FREQ_BINS_NOTES), dtype=np.float32),
            "onset": np.empty((T, N_FREQ_BINS_NOTES), dtype=np.float32),
            "contour": np.empty((T, N_FREQ_BINS_CONTOURS), dtype=np.float32),
        }
        rc = self._lib.bp_infer_pcm(
            self._handle, pcm.ctypes.data, pcm.shape[0], pcm.shape[1], int(sample_rate),
            out["note"].ctypes.data, out["onset"].ctypes.data, out["contour"].ctypes.data, _native.BP_MEM_HOST,
        )
        _native.check(self._lib, self._handle, rc, "bp_infer_pcm")
        return out

    def predict_pcm_raw(self, samples, fmt: int, n_frames: int, channels: int, sample_rate: int) -> Dict[str, np.ndarray]:
        """predict_pcm on interleaved samples in the file's own format (`fmt` = a BP_PCM_* code; `samples` = any buffer
        of n_frames * channels of them): they cross PCIe as they are — 16-bit stereo is half the bytes of its float form
        — and become float on the device, bit-identical to converting on the host first (bp_infer_pcm_raw)."""
        buf = np.frombuffer(samples, dtype=np.uint8)
        width = {_native.BP_PCM_F32: 4, _native.BP_PCM_S16: 2, _native.BP_PCM_S24: 3, _native.BP_PCM_S32: 4,
                 _native.BP_PCM_U8: 1, _native.BP_PCM_F64: 8}[int(fmt)]
        if buf.size < int(n_frames) * int(channels) * width:
            raise ValueError("predict_pcm_raw: the buffer is shorter than n_frames * channels samples")
        n22 = int(self._lib.bp_handle_resampled_length(self._handle, int(n_frames), int(sample_rate)))
        T = int(self._lib.bp_handle_track_n_frames(self._handle, n22))
        out = {
            "note": np.empty((T, N_FREQ_BINS_NOTES), dtype=np.float32),
            "onset": np.empty((T, N_FREQ_BINS_NOTES), dtype=np.float32),
            "contour": np.empty((T, N_FREQ_BINS_CONTOURS), dtype=np.float32),
        }
        rc = self._lib.bp_infer_pcm_raw(
            self._handle, buf.ctypes.data if buf.size else None, int(fmt), int(n_frames), int(channels), int(sample_rate),
            out["note"].ctypes.data, out["onset"].ctypes.data, out["contour"].ctypes.data, _native.BP_MEM_HOST,
        )
        _native.check(self._lib, self._handle, rc, "bp_infer_pcm_raw")
        return out

    def flac_layout(self, data: bytes) -> Dict[str, int]:
        """STREAMINFO of a FLAC file's bytes (host, no decoding): channels, sample_rate, bits_per_sample, block sizes,
        n_frames (0: not in the header)."""
        lay = _native.bp_flac_stream_layout()
        rc = self._lib.bp_flac_layout(data, len(data), C.byref(lay))
        if rc != _native.BP_OK:
            raise ValueError(f"bp_flac_layout: {self._lib.bp_audio_last_error().decode(errors='replace')}")
        return {k: int(getattr(lay, k)) for k, _ in lay._fields_}

    def flac_decode_device(self, data: bytes) -> Tuple[np.ndarray, int]:
        """A FLAC file's bytes decoded ON THE DEVICE (csrc/flac_device.hip): (int32 samples [n_frames, channels], sample
        rate) — the integers the host decoder's floats are made of.  ValueError if the stream cannot be decoded there
        (`NativeLibraryError` BP_ERR_UNSUPPORTED for what is left to the host decoder)."""
        lay = self.flac_layout(data)
        pcm = np.empty((max(0, lay["n_frames"]), lay["channels"]), dtype=np.int32)
        n = C.c_int64(0)
        rc = self._lib.bp_flac_decode_device(self._handle, data, len(data), pcm.ctypes.data if pcm.size else None, pcm.shape[0], C.byref(n))
        _native.check(self._lib, self._handle, rc, "bp_flac_decode_device")
        return pcm[: n.value], lay["sample_rate"]

    def predict_flac(self, data: bytes) -> Dict[str, np.ndarray]:
        """The posteriorgrams of a FLAC file's bytes: decode on the device, then what predict_pcm_raw does (bp_infer_flac)."""
        lay = self.flac_layout(data)
        n22 = int(self._lib.bp_handle_resampled_length(self._handle, lay["n_frames"], lay["sample_rate"]))
        T = int(self._lib.bp_handle_track_n_frames(self._handle, n22))
        out = {
            "note": np.empty((T, N_FREQ_BINS_NOTES), dtype=np.float32),
            "onset": np.empty((T, N_FREQ_BINS_NOTES), dtype=np.float32),
            "contour": np.empty((T, N_FREQ_BINS_CONTOURS), dtype=np.float32),
        }
        rc = self._lib.bp_infer_flac(self._handle, data, len(data), out["note"].ctypes.data, out["onset"].ctypes.data,
                                     out["contour"].ctypes.data, _native.BP_MEM_HOST)
        _native.check(self._lib, self._handle, rc, "bp_infer_flac")
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


# `predict(path)` with a model PATH loads the model on every call in the reference (inference.py:291-292 -> Model(...)).
# Loading is 60 ms here (parsing, packing the operand fragments, 0.8 GB of device buffers) against ~1 ms for the
# reference's 10-second clip itself, so a loaded model is kept and reused — PER THREAD (the handle is stateful and not for
# two threads at once) and in thread-local storage, so that the models of a thread are released when the thread ends
# (a process-wide table keyed by thread id kept up to 8 x 0.8 GB alive for threads long gone).  At most
# _MODEL_CACHE_MAX models per thread, oldest evicted first; clear_model_cache() drops the calling thread's.
_MODEL_CACHE_MAX = 2


class _ThreadModels(threading.local):
    def __init__(self) -> None:
        self.models: "Dict[Tuple[Any, ...], Model]" = {}


_MODEL_CACHE = _ThreadModels()


def clear_model_cache() -> None:
    """Forget the models `predict(path)` / `run_inference(path)` keep loaded for the calling thread (their device
    buffers are freed when nobody else holds the Model)."""
    _MODEL_CACHE.models.clear()


def _model_from(model_or_model_path) -> Any:
    """A Model (or anything shaped like one) as it is; a path as the calling thread's cached Model loaded from that file."""
    if not isinstance(model_or_model_path, (str, os.PathLike)):
        return model_or_model_path
    try:
        real = os.path.realpath(os.fspath(model_or_model_path))
        st = os.stat(real)
        key = (real, st.st_mtime_ns, st.st_size)
    except OSError:
        return Model(model_or_model_path)  # raises what loading a missing file raises
    models = _MODEL_CACHE.models
    model = models.get(key)
    if model is not None and model._handle.value:
        return model
    model = Model(model_or_model_path)
    while len(models) >= _MODEL_CACHE_MAX:
        models.pop(next(iter(models)))  # oldest first; the handle is freed when nobody holds the Model
    models[key] = model
    return model


def run_inference(
    audio_path: Union[pathlib.Path, str],
    model_or_model_path: Union[Model, pathlib.Path, str] = ICASSP_2022_MODEL_PATH,
    debug_file: Optional[pathlib.Path] = None,
) -> Dict[str, np.ndarray]:
    """Run the model on the input audio path (inference.py:282-330).

    Returns {"note": (T,88), "onset": (T,88), "contour": (T,264)} float32, T = int(L/36164*142).
    """
    model = _model_from(model_or_model_path)
    n_overlapping_frames = DEFAULT_OVERLAPPING_FRAMES
    overlap_len = n_overlapping_frames * FFT_HOP
    hop_size = AUDIO_N_SAMPLES - overlap_len

    # decode on the host (container parsing), everything after it on the device: channel-mean downmix, resampling
    # to 22.05 kHz, the 3840-sample lead-in + windowing, CQT + CNN, un-overlapping (inference.py:239-244, 302-315)
    with open(audio_path, "rb") as f:
        head = f.read(12)
    if head[:4] == b"RIFF" and head[8:12] == b"WAVE" and hasattr(model, "predict_pcm_raw"):
        # a WAV file's samples go to the device in the file's own format (no float copy on the host, half the PCIe bytes
        # for 16-bit audio); same posteriorgrams, bit for bit, as decoding them here
        raw, tag, bits, channels, file_sr = _audio.wav_raw(str(audio_path))
        fmt = {(1, 8): _native.BP_PCM_U8, (1, 16): _native.BP_PCM_S16, (1, 24): _native.BP_PCM_S24,
               (1, 32): _native.BP_PCM_S32, (3, 32): _native.BP_PCM_F32, (3, 64): _native.BP_PCM_F64}[(tag, bits)]
        n_file_frames = len(raw) // (bits // 8) // channels
        unwrapped_output = model.predict_pcm_raw(raw, fmt, n_file_frames, channels, file_sr)
    elif head[:4] == b"fLaC" and hasattr(model, "predict_flac"):
        # a FLAC file's BYTES go to the device and are decoded there (csrc/flac_device.hip): half the PCIe bytes of the PCM and
        # no host core-time per sample; what the device decoder leaves to the host (or cannot follow) is decoded here as
        # before — the host decoder also names the fault of a corrupt file
        with open(audio_path, "rb") as f:
            blob = f.read()
        try:
            lay = model.flac_layout(blob)
            if lay["n_frames"] <= 0:
                raise _native.NativeLibraryError("no sample count in STREAMINFO")
            unwrapped_output = model.predict_flac(blob)
            n_file_frames, file_sr = lay["n_frames"], lay["sample_rate"]
        except (ValueError, _native.NativeLibraryError):
            pcm, file_sr = _audio.read_audio(str(audio_path))
            n_file_frames = pcm.shape[0]
            unwrapped_output = model.predict_pcm(pcm, file_sr)
    else:
        pcm, file_sr = _audio.read_audio(str(audio_path))
        n_file_frames = pcm.shape[0]
        unwrapped_output = model.predict_pcm(pcm, file_sr)
    audio_original_length = int(-(-n_file_frames * AUDIO_SAMPLE_RATE // file_sr))  # librosa.resample: ceil(n * sr / file_sr)

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


def predict_window_range(
    audio_path: Union[pathlib.Path, str],
    model_or_model_path: Union[Model, pathlib.Path, str],
    piece: int,
    n_pieces: int,
) -> Dict[str, Any]:
    """Piece `piece` of `n_pieces` of ONE long file (the window-range fallback of the file-sharded job).

    A window's samples are determined by its index alone (start = w * 36164 - 3840: inference.py:207,242) and a window's 142
    un-overlapped rows by the window alone (per-window normalisation, signal.py:177-183), so a rank can compute windows
    [w0, w1) = `sharding.split_windows(n_windows, n_pieces)[piece]` without any of its neighbours' data: the file is decoded
    and resampled (on the device, by the same kernel the whole-file call uses), this range's windows are cut on the host
    exactly as `window_audio_file` cuts them and go through `Model.predict` in full batches; `assemble_window_ranges`
    concatenates the pieces' rows — the same bits the unsplit call produces (the library's results do not depend on how
    windows are batched: tests/test_gpu_parity.py::test_batch_invariance_and_chunking)."""
    from .sharding import split_windows

    model = _model_from(model_or_model_path)
    verify_input_path(audio_path)
    pcm, file_sr = _audio.read_audio(str(audio_path))
    y = np.ascontiguousarray(model.resample(pcm, file_sr), dtype=np.float32)
    n_overlapping_frames = DEFAULT_OVERLAPPING_FRAMES
    overlap_len = n_overlapping_frames * FFT_HOP
    hop_size = AUDIO_N_SAMPLES - overlap_len
    n_olap = n_overlapping_frames // 2
    padded = np.concatenate([np.zeros((overlap_len // 2,), dtype=np.float32), y])
    n_windows = len(range(0, padded.shape[0], hop_size))
    w0, w1 = split_windows(n_windows, n_pieces)[piece]
    rows: Dict[str, List[np.ndarray]] = {"note": [], "onset": [], "contour": []}
    batch = int(getattr(model, "max_windows", 256))
    for a in range(w0, w1, batch):
        b = min(w1, a + batch)
        x = np.zeros((b - a, AUDIO_N_SAMPLES), dtype=np.float32)
        for w in range(a, b):
            seg = padded[w * hop_size : w * hop_size + AUDIO_N_SAMPLES]
            x[w - a, : len(seg)] = seg
        for k, v in model.predict(x).items():
            if k in rows:
                v = np.asarray(v)[:, n_olap:-n_olap, :]
                rows[k].append(v.reshape(v.shape[0] * v.shape[1], v.shape[2]))
    width = {"note": 88, "onset": 88, "contour": 264}
    return {"rows": {k: (np.concatenate(v) if v else np.zeros((0, width[k]), np.float32)) for k, v in rows.items()},
            "range": (w0, w1), "n_windows": n_windows, "original_length": int(y.shape[0])}


def assemble_window_ranges(
    parts: Sequence[Dict[str, Any]],
    onset_threshold: float = DEFAULT_ONSET_THRESHOLD,
    frame_threshold: float = DEFAULT_FRAME_THRESHOLD,
    minimum_note_length: float = DEFAULT_MINIMUM_NOTE_LENGTH_MS,
    minimum_frequency: Optional[float] = None,
    maximum_frequency: Optional[float] = None,
    multiple_pitch_bends: bool = False,
    melodia_trick: bool = True,
    midi_tempo: float = DEFAULT_MINIMUM_MIDI_TEMPO,
    **_ignored: Any,
) -> Tuple[Dict[str, np.ndarray], "infer.pretty_midi.PrettyMIDI", List["infer.NoteEvent"]]:
    """The pieces of `predict_window_range` back into what `predict()` returns for the file: rows concatenated in window
    order, trimmed to int(L / hop * 142) (`unwrap_output`, inference.py:277-279), notes decoded over the WHOLE file (a note
    may span pieces)."""
    parts = sorted(parts, key=lambda p: p["range"][0])
    at = 0
    for p in parts:
        if p["range"][0] != at:
            raise ValueError(f"window ranges do not tile the file: expected a piece starting at window {at}, got {p['range']}")
        at = p["range"][1]
    if not parts or at != parts[0]["n_windows"]:
        raise ValueError("window ranges do not cover the file")
    hop_size = AUDIO_N_SAMPLES - DEFAULT_OVERLAPPING_FRAMES * FFT_HOP
    n_rows = int(parts[0]["original_length"] / hop_size * (AUDIO_WINDOW_LENGTH * ANNOTATIONS_FPS - DEFAULT_OVERLAPPING_FRAMES))
    model_output = {k: np.ascontiguousarray(np.concatenate([p["rows"][k] for p in parts])[:n_rows]) for k in ("note", "onset", "contour")}
    min_note_len = int(np.round(minimum_note_length / 1000 * (AUDIO_SAMPLE_RATE / FFT_HOP)))
    midi_data, note_events = infer.model_output_to_notes(
        model_output, onset_thresh=onset_threshold, frame_thresh=frame_threshold, min_note_len=min_note_len,
        min_freq=minimum_frequency, max_freq=maximum_frequency, multiple_pitch_bends=multiple_pitch_bends,
        melodia_trick=melodia_trick, midi_tempo=midi_tempo,
    )
    return model_output, midi_data, note_events


class OutputExtensions(enum.Enum):
    MIDI = "mid"
    MODEL_OUTPUT_NPZ = "npz"
    MIDI_SONIFICATION = "wav"
    NOTE_EVENTS = "csv"


def verify_input_path(audio_path: Union[pathlib.Path, str]) -> None:
    """inference.py:340-354."""
    if not os.path.isfile(audio_path):
        raise ValueError(f"{audio_path} is not a file path.")


def verify_output_dir(output_dir: Union[pathlib.Path, str]) -> None:
    """inference.py:357-371."""
    if not os.path.isdir(output_dir):
        raise ValueError(f"{output_dir} is not a directory.")


def build_output_path(
    audio_path: Union[pathlib.Path, str], output_directory: Union[pathlib.Path, str], output_type: OutputExtensions
) -> pathlib.Path:
    """inference.py:374-406: <dir>/<stem>_basic_pitch.<ext>; IOError if it already exists."""
    basename, _ = os.path.splitext(os.path.basename(str(audio_path)))
    output_path = pathlib.Path(output_directory) / f"{basename}_basic_pitch.{output_type.value}"
    if output_path.exists():
        raise IOError(f"{output_path} already exists and would be overwritten. Skipping output files for {audio_path}.")
    return output_path


def save_note_events(note_events: List["infer.NoteEvent"], save_path: Union[pathlib.Path, str]) -> None:
    """inference.py:409-428: CSV with start_time_s, end_time_s, pitch_midi, velocity, pitch_bend..."""
    with open(save_path, "w") as fhandle:
        writer = csv.writer(fhandle, delimiter=",")
        writer.writerow(["start_time_s", "end_time_s", "pitch_midi", "velocity", "pitch_bend"])
        for start_time, end_time, note_number, amplitude, pitch_bend in note_events:
            row = [start_time, end_time, note_number, int(np.round(DEFAULT_MIDI_VELOCITY_SCALE * amplitude))]
            if pitch_bend:
                row.extend(pitch_bend)
            writer.writerow(row)


def predict(
    audio_path: Union[pathlib.Path, str],
    model_or_model_path: Union[Model, pathlib.Path, str] = ICASSP_2022_MODEL_PATH,
    onset_threshold: float = DEFAULT_ONSET_THRESHOLD,
    frame_threshold: float = DEFAULT_FRAME_THRESHOLD,
    minimum_note_length: float = DEFAULT_MINIMUM_NOTE_LENGTH_MS,
    minimum_frequency: Optional[float] = None,
    maximum_frequency: Optional[float] = None,
    multiple_pitch_bends: bool = False,
    melodia_trick: bool = True,
    debug_file: Optional[pathlib.Path] = None,
    midi_tempo: float = DEFAULT_MINIMUM_MIDI_TEMPO,
) -> Tuple[Dict[str, np.ndarray], "infer.pretty_midi.PrettyMIDI", List["infer.NoteEvent"]]:
    """Run a single prediction (inference.py:431-506): (model_output, midi_data, note_events)."""
    model_output = run_inference(audio_path, model_or_model_path, debug_file)
    min_note_len = int(np.round(minimum_note_length / 1000 * (AUDIO_SAMPLE_RATE / FFT_HOP)))
    midi_data, note_events = infer.model_output_to_notes(
        model_output,
        onset_thresh=onset_threshold,
        frame_thresh=frame_threshold,
        min_note_len=min_note_len,
        min_freq=minimum_frequency,
        max_freq=maximum_frequency,
        multiple_pitch_bends=multiple_pitch_bends,
        melodia_trick=melodia_trick,
        midi_tempo=midi_tempo,
    )
    if debug_file:
        with open(debug_file) as f:
            debug_data = json.load(f)
        with open(debug_file, "w") as f:
            json.dump(
                {
                    **debug_data,
                    "min_note_length": min_note_len,
                    "onset_thresh": onset_threshold,
                    "frame_thresh": frame_threshold,
                    "estimated_notes": [
                        (float(s0), float(s1), int(pitch), float(amp), [int(b) for b in pb] if pb else None)
                        for s0, s1, pitch, amp, pb in note_events
                    ],
                },
                f,
            )
    return model_output, midi_data, note_events


def predict_many(
    audio_paths: Sequence[Union[pathlib.Path, str]],
    model_or_model_path: Union[Model, pathlib.Path, str] = ICASSP_2022_MODEL_PATH,
    onset_threshold: float = DEFAULT_ONSET_THRESHOLD,
    frame_threshold: float = DEFAULT_FRAME_THRESHOLD,
    minimum_note_length: float = DEFAULT_MINIMUM_NOTE_LENGTH_MS,
    minimum_frequency: Optional[float] = None,
    maximum_frequency: Optional[float] = None,
    multiple_pitch_bends: bool = False,
    melodia_trick: bool = True,
    midi_tempo: float = DEFAULT_MINIMUM_MIDI_TEMPO,
    group: int = 64,
    decode_threads: Optional[int] = None,
    return_exceptions: bool = False,
    _finish: Optional[Any] = None,
) -> List[Tuple[Dict[str, np.ndarray], "infer.pretty_midi.PrettyMIDI", List["infer.NoteEvent"]]]:
    """predict() (inference.py:431-506) over many files, results in input order and identical to per-file predict().

    The reference runs one file after the other and one window per runtime call.  Here the files of a group are
    decoded on a host thread pool (file read + container parsing; the device does downmix and resampling:
    bp_resample), their windows packed across file boundaries into full batches (bp_infer_tracks), and the note
    decoding of a finished group (host C++, GIL released) runs on the same pool while the GPU works on the next
    group: one GPU produces posteriorgrams ~200 x faster than one host core decodes them.

    `return_exceptions=True` isolates failures per file like the reference's per-file try / except
    (inference.py:548-604): the entry of a file that cannot be read or decoded is the exception instead of a tuple.

    `_finish(index, result)` (internal; `predict_and_save_many`) runs on the pool right behind a file's note decoding and
    its return value replaces the file's entry — the writers of a batch job, so that ONE pipeline runs over all files
    (next group read while this one is on the GPU, decoding and writing of finished files behind it) and a file's
    posteriorgrams are dropped as soon as they are written.  The GPU is kept at most two groups ahead of the decoders.
    """
    import concurrent.futures as cf
    import os

    # a path loads the model like the reference does (once per thread: _model_from); a Model — or anything with its
    # resample / predict_tracks — is used
    model = _model_from(model_or_model_path)
    if group < 1:
        raise ValueError("group must be >= 1")
    paths = [pathlib.Path(p) for p in audio_paths]
    if not return_exceptions:
        for p in paths:
            verify_input_path(p)
    min_note_len = int(np.round(minimum_note_length / 1000 * (AUDIO_SAMPLE_RATE / FFT_HOP)))

    def decode(model_output: Dict[str, np.ndarray]):
        midi_data, note_events = infer.model_output_to_notes(
            model_output, onset_thresh=onset_threshold, frame_thresh=frame_threshold, min_note_len=min_note_len,
            min_freq=minimum_frequency, max_freq=maximum_frequency, multiple_pitch_bends=multiple_pitch_bends,
            melodia_trick=melodia_trick, midi_tempo=midi_tempo,
        )
        return model_output, midi_data, note_events

    def decode_and_finish(i: int, model_output: Dict[str, np.ndarray]):
        res = decode(model_output)
        return res if _finish is None else _finish(i, res)

    def read(p: pathlib.Path):
        verify_input_path(p)
        return _audio.read_audio(str(p))

    workers = decode_threads if decode_threads else min(16, os.cpu_count() or 1)
    results: List[Any] = [None] * len(paths)
    with cf.ThreadPoolExecutor(max_workers=workers) as pool:
        groups = [list(range(g0, min(g0 + group, len(paths)))) for g0 in range(0, len(paths), group)]
        pending = [pool.submit(read, paths[i]) for i in groups[0]] if groups else []
        for gi, ids in enumerate(groups):
            reads, pending = pending, []
            if gi + 1 < len(groups):  # the next group's files are read while this one is on the GPU
                pending = [pool.submit(read, paths[i]) for i in groups[gi + 1]]
            signals, good = [], []
            for i, fut in zip(ids, reads):
                try:
                    pcm, sr = fut.result()
                    signals.append(model.resample(pcm, sr))  # downmix + resampling on the device
                    good.append(i)
                except Exception as e:
                    if not return_exceptions:
                        raise
                    results[i] = e
            if _finish is not None and gi >= 2:  # back-pressure: posteriorgrams of at most two groups await decoding
                for i in groups[gi - 2]:
                    if isinstance(results[i], cf.Future):
                        cf.wait([results[i]])
            for i, out in zip(good, model.predict_tracks(signals)):
                results[i] = pool.submit(decode_and_finish, i, out)
        for i, r in enumerate(results):
            if isinstance(r, cf.Future):
                try:
                    results[i] = r.result()
                except Exception as e:
                    if not return_exceptions:
                        raise
                    results[i] = e
    return results


def _save_outputs(audio_path, output_directory, result, save_midi: bool, sonify_midi: bool, save_model_outputs: bool,
                  save_notes: bool, sonification_samplerate: int) -> Dict[str, str]:
    """The four writers of inference.py:565-602 for one file's `(model_output, midi_data, note_events)`; returns the paths."""
    model_output, midi_data, note_events = result
    written: Dict[str, str] = {}
    if save_model_outputs:
        path = build_output_path(audio_path, output_directory, OutputExtensions.MODEL_OUTPUT_NPZ)
        np.savez(path, basic_pitch_model_output=model_output)
        written["model_output"] = str(path)
    if save_midi:
        path = build_output_path(audio_path, output_directory, OutputExtensions.MIDI)
        midi_data.write(str(path))
        written["midi"] = str(path)
    if sonify_midi:
        path = build_output_path(audio_path, output_directory, OutputExtensions.MIDI_SONIFICATION)
        infer.sonify_midi(midi_data, path, sr=sonification_samplerate)
        written["sonification"] = str(path)
    if save_notes:
        path = build_output_path(audio_path, output_directory, OutputExtensions.NOTE_EVENTS)
        save_note_events(note_events, path)
        written["note_events"] = str(path)
    return written


def predict_and_save_many(
    audio_path_list,
    output_directory: Union[pathlib.Path, str],
    save_midi: bool,
    sonify_midi: bool,
    save_model_outputs: bool,
    save_notes: bool,
    model_or_model_path: Union[Model, str, pathlib.Path] = ICASSP_2022_MODEL_PATH,
    onset_threshold: float = DEFAULT_ONSET_THRESHOLD,
    frame_threshold: float = DEFAULT_FRAME_THRESHOLD,
    minimum_note_length: float = DEFAULT_MINIMUM_NOTE_LENGTH_MS,
    minimum_frequency: Optional[float] = None,
    maximum_frequency: Optional[float] = None,
    multiple_pitch_bends: bool = False,
    melodia_trick: bool = True,
    sonification_samplerate: int = DEFAULT_SONIFICATION_SAMPLERATE,
    midi_tempo: float = DEFAULT_MINIMUM_MIDI_TEMPO,
    group: int = 64,
    decode_threads: Optional[int] = None,
    return_exceptions: bool = False,
) -> List[Any]:
    """`predict_and_save` (inference.py:509-618) as a batch job: ONE `predict_many` pipeline over all files (windows
    packed across files; the next group's files are read while this one is on the GPU; note decoding AND the writers of
    a finished file run on host threads behind it), a file's posteriorgrams are dropped once its outputs are written and
    the GPU runs at most two groups ahead, so memory stays at a few groups of posteriorgrams.  Same files, same bytes
    as the per-file loop.  Returns per file `{"n_note_events": k, "outputs": {kind: path}}` — or, with
    `return_exceptions=True`, the exception that file raised (otherwise the first failure propagates, like the
    reference's `raise e`).  Two inputs that map to the same output name (same stem) cannot both be written: as in the
    reference's sequential loop (inference.py:401-404) the first one wins and every later one gets the "already exists"
    IOError — decided up front, so concurrent writers never race on the exists-check."""
    model = _model_from(model_or_model_path)
    paths = [pathlib.Path(p) for p in audio_path_list]
    saving = save_midi or sonify_midi or save_model_outputs or save_notes
    # two inputs with the same stem would write the same files.  The map is made up front only so that concurrent writers
    # never race on the exists-check; what it MEANS follows the reference's sequential loop (inference.py:548-604): every
    # earlier file is predicted and written first, the later one then finds the outputs and gets the IOError — if the
    # earlier file really produced them — and nothing can collide when nothing is saved.
    dup = duplicate_output_stems(paths) if saving else {}
    if dup and not return_exceptions:
        paths = paths[: min(dup)]  # the loop would never get past the first duplicate: predict and write what precedes it
    todo = [i for i in range(len(paths)) if i not in dup]

    def finish_for(ids):
        def finish(j: int, res):
            written = _save_outputs(paths[ids[j]], output_directory, res, save_midi, sonify_midi, save_model_outputs,
                                    save_notes, sonification_samplerate)
            return {"n_note_events": len(res[2]), "outputs": written}
        return finish

    def run(ids):
        return predict_many([paths[i] for i in ids], model, onset_threshold, frame_threshold, minimum_note_length,
                            minimum_frequency, maximum_frequency, multiple_pitch_bends, melodia_trick, midi_tempo, group=group,
                            decode_threads=decode_threads, return_exceptions=return_exceptions, _finish=finish_for(ids))

    report: List[Any] = [None] * len(paths)
    for i, r in zip(todo, run(todo)):
        report[i] = r
    if dup and not return_exceptions:
        raise dup[min(dup)]  # every file in front of it has been written, like the reference's `raise e`
    # a duplicate whose earlier namesake FAILED (nothing was written under that stem) is an ordinary file after all
    first_of: Dict[str, int] = {}
    retry: List[int] = []
    for i in range(len(paths)):
        stem, _ = os.path.splitext(os.path.basename(str(paths[i])))
        if i not in dup:
            first_of.setdefault(stem, i)
        elif isinstance(report[first_of.get(stem, i)], BaseException) and stem not in {os.path.splitext(os.path.basename(str(paths[k])))[0] for k in retry}:
            retry.append(i)
        else:
            report[i] = dup[i]
    for i, r in zip(retry, run(retry) if retry else []):
        report[i] = r
    return report


def lane_models(model_path: Union[str, pathlib.Path] = ICASSP_2022_MODEL_PATH, devices: Optional[Sequence[int]] = None,
                lanes: int = 3) -> List[Model]:
    """The GPU lanes of a native file job: `lanes` models on every device of `devices` (default: device 0), device-major
    — lane i sits on devices[i // lanes].  Small workspaces (128 windows: one 3-minute track is 110), blocking waits so
    that a host thread parked on a lane leaves its core to the workers."""
    return [Model(model_path, device=int(d), max_windows=128, blocking_wait=True)
            for d in (devices if devices is not None else [0]) for _ in range(max(1, int(lanes)))]


def transcribe_files(
    audio_path_list,
    output_directory: Union[pathlib.Path, str],
    save_midi: bool = True,
    save_notes: bool = True,
    model_or_model_path: Union[Model, str, pathlib.Path] = ICASSP_2022_MODEL_PATH,
    onset_threshold: float = DEFAULT_ONSET_THRESHOLD,
    frame_threshold: float = DEFAULT_FRAME_THRESHOLD,
    minimum_note_length: float = DEFAULT_MINIMUM_NOTE_LENGTH_MS,
    minimum_frequency: Optional[float] = None,
    maximum_frequency: Optional[float] = None,
    multiple_pitch_bends: bool = False,
    melodia_trick: bool = True,
    midi_tempo: float = DEFAULT_MINIMUM_MIDI_TEMPO,
    models: Optional[Sequence[Model]] = None,
    lanes: int = 3,
    threads: int = 0,
    devices: Optional[Sequence[int]] = None,
    host_decode: bool = False,
    direct_io: bool = False,
    host_flac: bool = False,
) -> List[Dict[str, Any]]:
    """The batch job of `predict_and_save` (inference.py:509-604) for WAV / FLAC input and MIDI / note-event output, run
    natively: ONE call into the library (`bp_transcribe_files`, csrc/file_pipeline.cpp), C++ worker threads from the
    file's bytes to its outputs, no Python in the loop.  Same bytes as `predict_and_save(..., save_midi, False, False,
    save_notes)`.  `models` (or `lanes` models per device of `devices` — default: device 0 — built from
    `model_or_model_path`) are the GPU lanes the workers queue for; lanes may live on different GPUs, a worker takes
    whichever is free, so ONE call shards its files over all GPUs of a node by itself (file-granular, no collective).
    Returns per file `{"status": 0 | bp_status, "n_note_events": k, "n_frames": T, "message": str, "ms": {stage: wall
    milliseconds of the worker}}`; per-file
    failures are reported, not raised (the reference's per-file try / except).  By default the dense half of note
    decoding runs on the device and 7 MB per 3-minute track come back instead of 27.6 (`host_decode=True`: the round-4
    path, all three posteriorgrams decoded on the host; same bytes).  `direct_io=True` reads the files with O_DIRECT straight
    into the page-locked buffers the GPU copies from (no page-cache copy; for corpora larger than the page cache)."""
    own: List[Model] = []
    if models is None:
        if isinstance(model_or_model_path, Model):
            models = [model_or_model_path]
        else:
            own = lane_models(model_or_model_path, devices, lanes)
            models = own
    try:
        lib = models[0]._lib
        verify_output_dir(output_directory)
        paths = [os.fsencode(str(p)) for p in audio_path_list]
        n = len(paths)
        prm = _native.bp_transcribe_params()
        lib.bp_transcribe_params_default(C.byref(prm))
        prm.notes.onset_threshold, prm.notes.frame_threshold = float(onset_threshold), float(frame_threshold)
        prm.notes.min_note_len = int(np.round(minimum_note_length / 1000 * (AUDIO_SAMPLE_RATE / FFT_HOP)))
        prm.notes.melodia_trick = int(bool(melodia_trick))
        prm.notes.min_freq_hz = float(minimum_frequency) if minimum_frequency is not None else 0.0
        prm.notes.max_freq_hz = float(maximum_frequency) if maximum_frequency is not None else 0.0
        prm.midi_tempo = float(midi_tempo)
        prm.multiple_pitch_bends = int(bool(multiple_pitch_bends))
        prm.save_midi, prm.save_notes, prm.threads = int(bool(save_midi)), int(bool(save_notes)), int(threads)
        prm.host_decode = int(bool(host_decode))
        prm.direct_io = int(bool(direct_io))
        prm.host_flac = int(bool(host_flac))  # FLAC files: decoded on the device unless asked otherwise
        handles = (C.c_void_p * len(models))(*[m._handle for m in models])
        cpaths = (C.c_char_p * max(1, n))(*paths)
        reports = (_native.bp_file_report * max(1, n))()
        rc = lib.bp_transcribe_files(handles, len(models), cpaths, n, os.fsencode(str(output_directory)), C.byref(prm), reports)
        if rc != _native.BP_OK:
            raise ValueError(f"bp_transcribe_files: {lib.bp_files_last_error().decode(errors='replace')}")
        return [{"status": int(reports[i].status), "n_note_events": int(reports[i].n_note_events),
                 "n_frames": int(reports[i].n_frames), "message": reports[i].message.decode(errors="replace"),
                 "ms": {k: float(getattr(reports[i], "ms_" + k)) for k in ("read", "lane_wait", "device", "notes", "write")}}
                for i in range(n)]
    finally:
        for m in own:
            m.close()


def duplicate_output_stems(paths: Sequence[Union[pathlib.Path, str]]) -> Dict[int, IOError]:
    """Indices of inputs whose output name `<stem>_basic_pitch.*` (build_output_path) an EARLIER input of the list
    already claims, with the IOError the reference's sequential loop would give them (inference.py:401-404)."""
    seen: Dict[str, int] = {}
    dup: Dict[int, IOError] = {}
    for i, p in enumerate(paths):
        stem, _ = os.path.splitext(os.path.basename(str(p)))
        if stem in seen:
            dup[i] = IOError(f"the outputs of {p} would overwrite those of {paths[seen[stem]]} (same file stem "
                             f"'{stem}'). Skipping output files for {p}.")
        else:
            seen[stem] = i
    return dup


def predict_and_save(
    audio_path_list,
    output_directory: Union[pathlib.Path, str],
    save_midi: bool,
    sonify_midi: bool,
    save_model_outputs: bool,
    save_notes: bool,
    model_or_model_path: Union[Model, str, pathlib.Path] = ICASSP_2022_MODEL_PATH,
    onset_threshold: float = DEFAULT_ONSET_THRESHOLD,
    frame_threshold: float = DEFAULT_FRAME_THRESHOLD,
    minimum_note_length: float = DEFAULT_MINIMUM_NOTE_LENGTH_MS,
    minimum_frequency: Optional[float] = None,
    maximum_frequency: Optional[float] = None,
    multiple_pitch_bends: bool = False,
    melodia_trick: bool = True,
    debug_file: Optional[pathlib.Path] = None,
    sonification_samplerate: int = DEFAULT_SONIFICATION_SAMPLERATE,
    midi_tempo: float = DEFAULT_MINIMUM_MIDI_TEMPO,
) -> None:
    """inference.py:509-618: model output (.npz), MIDI (.mid), sonified MIDI (.wav), note events (.csv) per file."""
    model = _model_from(model_or_model_path)
    for audio_path in audio_path_list:
        result = predict(
            pathlib.Path(audio_path), model, onset_threshold, frame_threshold, minimum_note_length,
            minimum_frequency, maximum_frequency, multiple_pitch_bends, melodia_trick, debug_file, midi_tempo,
        )
        _save_outputs(audio_path, output_directory, result, save_midi, sonify_midi, save_model_outputs, save_notes,
                      sonification_samplerate)
