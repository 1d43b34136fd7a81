"""ctypes binding of libbasicpitch_amd.so (C ABI in include/basic_pitch_amd.h).

The product path FAILS LOUDLY when the HIP library is missing or no MI355X is visible: there is no
CPU fallback in this package (the CPU restatement under oracle/ is test infrastructure only).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build

BP_OK = 0
BP_ERR_INVALID_ARG = -1
BP_ERR_UNSUPPORTED = -6
BP_MEM_HOST = 0
BP_MEM_DEVICE = 1
BP_FLAG_STAGE_TIMING = 1
BP_FLAG_TIME_DOMINANT = 16
BP_FLAG_F32_MFMA = 2
BP_FLAG_BF16_WEIGHTS = 4
BP_FLAG_EXT_CQT_44K = 8
BP_FLAG_F16_CORRECTIONS = 32
BP_FLAG_FP8_CORRECTIONS = 64
BP_FLAG_BLOCKING_WAIT = 128
BP_PCM_F32, BP_PCM_S16, BP_PCM_S24, BP_PCM_S32, BP_PCM_U8, BP_PCM_F64 = range(6)
BP_N_STAGES = 15
BP_Z_ROW = 448
BP_Z_ROWS = 174
BP_Z_PAD = 56
BP_PYR_STRIDE = 43712

STAGE_NAMES = [
    "pyramid", "filterbank", "contour1", "contour2", "note1", "note2", "onset1", "onset2",
    "zpack", "note", "onset", "contour", "contour_conv1", "contour_conv2", "contour_conv1_edge",
]

_ERR_NAMES = {
    -1: "BP_ERR_INVALID_ARG",
    -2: "BP_ERR_BAD_WEIGHTS",
    -3: "BP_ERR_NO_DEVICE",
    -4: "BP_ERR_HIP",
    -5: "BP_ERR_OUT_OF_MEMORY",
    -6: "BP_ERR_UNSUPPORTED",
    -7: "BP_ERR_BAD_AUDIO",
}


class NativeLibraryError(RuntimeError):
    """The HIP library is missing / unloadable, or a HIP call failed."""


class bp_info(C.Structure):
    _fields_ = [
        ("device_ordinal", C.c_int),
        ("compute_units", C.c_int),
        ("max_windows", C.c_int64),
        ("workspace_bytes", C.c_int64),
        ("arch", C.c_char * 32),
    ]


class bp_stage_buffers(C.Structure):
    _fields_ = [
        ("audio", C.c_void_p),
        ("pyr", C.c_void_p),
        ("lp", C.c_void_p),
        ("mm", C.c_void_p),
        ("c1", C.c_void_p),
        ("contour", C.c_void_p),
        ("n1", C.c_void_p),
        ("note", C.c_void_p),
        ("o1", C.c_void_p),
        ("onset", C.c_void_p),
        ("zp", C.c_void_p),
    ]


class bp_note_params(C.Structure):
    _fields_ = [
        ("onset_threshold", C.c_double),
        ("frame_threshold", C.c_double),
        ("min_freq_hz", C.c_double),
        ("max_freq_hz", C.c_double),
        ("min_note_len", C.c_int32),
        ("infer_onsets", C.c_int32),
        ("melodia_trick", C.c_int32),
        ("include_pitch_bends", C.c_int32),
        ("energy_tol", C.c_int32),
        ("reserved", C.c_int32),
    ]


class bp_flac_stream_layout(C.Structure):
    _fields_ = [
        ("channels", C.c_int32),
        ("sample_rate", C.c_int32),
        ("bits_per_sample", C.c_int32),
        ("min_block", C.c_int32),
        ("max_block", C.c_int32),
        ("n_frames", C.c_int64),
        ("audio_start", C.c_int64),
    ]


class bp_transcribe_params(C.Structure):
    _fields_ = [
        ("notes", bp_note_params),
        ("midi_tempo", C.c_double),
        ("multiple_pitch_bends", C.c_int32),
        ("save_midi", C.c_int32),
        ("save_notes", C.c_int32),
        ("threads", C.c_int32),
        ("host_decode", C.c_int32),
        ("direct_io", C.c_int32),
        ("host_flac", C.c_int32),
        ("reserved", C.c_int32 * 1),
    ]


class bp_file_report(C.Structure):
    _fields_ = [
        ("status", C.c_int32),
        ("n_note_events", C.c_int32),
        ("n_frames", C.c_int64),
        ("message", C.c_char * 240),
        ("ms_read", C.c_float),
        ("ms_lane_wait", C.c_float),
        ("ms_device", C.c_float),
        ("ms_notes", C.c_float),
        ("ms_write", C.c_float),
    ]


class bp_note_event(C.Structure):
    _fields_ = [
        ("start_s", C.c_double),
        ("end_s", C.c_double),
        ("bend_offset", C.c_int64),
        ("start_frame", C.c_int32),
        ("end_frame", C.c_int32),
        ("pitch_midi", C.c_int32),
        ("n_bends", C.c_int32),
        ("amplitude", C.c_float),
        ("reserved", C.c_int32),
    ]


# every symbol include/basic_pitch_amd.h declares (tests check they are all exported)
EXPORTED_SYMBOLS = [
    "bp_create",
    "bp_destroy",
    "bp_last_error",
    "bp_infer",
    "bp_infer_async",
    "bp_infer_track",
    "bp_infer_tracks",
    "bp_resampled_length",
    "bp_resample",
    "bp_infer_pcm",
    "bp_infer_pcm_raw",
    "bp_host_alloc",
    "bp_host_free",
    "bp_files_release_buffers",
    "bp_files_direct_reads",
    "bp_files_read_probe",
    "bp_track_n_windows",
    "bp_handle_track_n_windows",
    "bp_handle_track_n_frames",
    "bp_handle_window_samples",
    "bp_handle_sample_rate",
    "bp_handle_resampled_length",
    "bp_track_n_frames",
    "bp_set_stream",
    "bp_synchronize",
    "bp_get_info",
    "bp_get_stage_ms",
    "bp_run_stage",
    "bp_pyramid_layout",
    "bp_version",
    "bp_device_count",
    "bp_note_params_default",
    "bp_notes_decode",
    "bp_note_candidates",
    "bp_infer_pcm_raw_candidates",
    "bp_track_maps",
    "bp_notes_decode_candidates",
    "bp_notes_last_error",
    "bp_flac_info",
    "bp_flac_decode",
    "bp_flac_layout",
    "bp_flac_decode_device",
    "bp_infer_flac",
    "bp_infer_flac_candidates",
    "bp_audio_last_error",
    "bp_transcribe_params_default",
    "bp_transcribe_files",
    "bp_notes_to_midi",
    "bp_notes_to_csv",
    "bp_wav_info",
    "bp_wav_decode",
    "bp_files_last_error",
]

_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen the in-tree library and declare argtypes.  Raises NativeLibraryError if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("BASIC_PITCH_AMD_LIB") or _build.LIB_PATH
    if not os.path.exists(p):
        raise NativeLibraryError(
            f"{p} not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950). basic_pitch_amd has no CPU fallback."
        )
    # ONE HIP runtime per process.  PyTorch's libraries ask for "libamdhip64.so" (and find the copy bundled in torch/lib),
    # this library for "libamdhip64.so.7" (the system's /opt/rocm copy); both files carry the SONAME libamdhip64.so.7.  When
    # torch loads first, this library's request matches the SONAME of the copy already in the process.  The other way
    # round torch's request matched nothing and a SECOND runtime was loaded — whichever initialised second then saw no
    # device (measured: build() followed by smoke() in one process).  Loading the runtime by torch's name first puts that
    # name on the link map, so a later `import torch` reuses it.
    try:
        C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    except OSError:
        pass
    try:
        lib = C.CDLL(p)
    except OSError as e:  # missing libamdhip64 etc.
        raise NativeLibraryError(f"cannot load {p}: {e}") from e
    vp, i64, fp = C.c_void_p, C.c_int64, C.c_void_p
    lib.bp_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_uint, i64, C.POINTER(vp)]
    lib.bp_create.restype = C.c_int
    lib.bp_destroy.argtypes = [vp]
    lib.bp_destroy.restype = None
    lib.bp_last_error.argtypes = [vp]
    lib.bp_last_error.restype = C.c_char_p
    lib.bp_infer.argtypes = [vp, fp, i64, fp, fp, fp, C.c_int]
    lib.bp_infer.restype = C.c_int
    lib.bp_infer_async.argtypes = [vp, fp, i64, fp, fp, fp]
    lib.bp_infer_async.restype = C.c_int
    lib.bp_infer_track.argtypes = [vp, fp, i64, fp, fp, fp, C.c_int]
    lib.bp_infer_track.restype = C.c_int
    lib.bp_infer_tracks.argtypes = [vp, i64, vp, vp, vp, vp, vp, C.c_int]
    lib.bp_infer_tracks.restype = C.c_int
    lib.bp_resampled_length.argtypes = [i64, C.c_int]
    lib.bp_resampled_length.restype = i64
    lib.bp_resample.argtypes = [vp, fp, i64, C.c_int, C.c_int, fp, C.c_int]
    lib.bp_resample.restype = C.c_int
    lib.bp_infer_pcm.argtypes = [vp, fp, i64, C.c_int, C.c_int, fp, fp, fp, C.c_int]
    lib.bp_infer_pcm.restype = C.c_int
    lib.bp_infer_pcm_raw.argtypes = [vp, vp, C.c_int, i64, C.c_int, C.c_int, fp, fp, fp, C.c_int]
    lib.bp_infer_pcm_raw.restype = C.c_int
    lib.bp_flac_layout.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(bp_flac_stream_layout)]
    lib.bp_flac_layout.restype = C.c_int
    lib.bp_flac_decode_device.argtypes = [vp, C.c_char_p, C.c_size_t, vp, i64, C.POINTER(i64)]
    lib.bp_flac_decode_device.restype = C.c_int
    lib.bp_infer_flac.argtypes = [vp, C.c_char_p, C.c_size_t, fp, fp, fp, C.c_int]
    lib.bp_infer_flac.restype = C.c_int
    lib.bp_host_alloc.argtypes = [C.c_size_t]
    lib.bp_host_alloc.restype = C.c_void_p
    lib.bp_host_free.argtypes = [C.c_void_p]
    lib.bp_host_free.restype = None
    lib.bp_files_release_buffers.argtypes = []
    lib.bp_files_release_buffers.restype = None
    lib.bp_files_direct_reads.argtypes = []
    lib.bp_files_direct_reads.restype = C.c_int64
    lib.bp_files_read_probe.argtypes = [C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    lib.bp_files_read_probe.restype = C.c_int64
    lib.bp_handle_track_n_windows.argtypes = [vp, i64]
    lib.bp_handle_track_n_windows.restype = i64
    lib.bp_handle_track_n_frames.argtypes = [vp, i64]
    lib.bp_handle_track_n_frames.restype = i64
    lib.bp_handle_window_samples.argtypes = [vp]
    lib.bp_handle_window_samples.restype = i64
    lib.bp_handle_sample_rate.argtypes = [vp]
    lib.bp_handle_sample_rate.restype = C.c_int
    lib.bp_handle_resampled_length.argtypes = [vp, i64, C.c_int]
    lib.bp_handle_resampled_length.restype = i64
    lib.bp_track_n_windows.argtypes = [i64]
    lib.bp_track_n_windows.restype = i64
    lib.bp_track_n_frames.argtypes = [i64]
    lib.bp_track_n_frames.restype = i64
    lib.bp_set_stream.argtypes = [vp, vp]
    lib.bp_set_stream.restype = C.c_int
    lib.bp_synchronize.argtypes = [vp]
    lib.bp_synchronize.restype = C.c_int
    lib.bp_get_info.argtypes = [vp, C.POINTER(bp_info)]
    lib.bp_get_info.restype = C.c_int
    lib.bp_get_stage_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
    lib.bp_get_stage_ms.restype = C.c_int
    lib.bp_run_stage.argtypes = [vp, C.c_int, C.POINTER(bp_stage_buffers), i64]
    lib.bp_run_stage.restype = C.c_int
    lib.bp_pyramid_layout.argtypes = [C.c_int, C.POINTER(i64), C.POINTER(i64)]
    lib.bp_pyramid_layout.restype = C.c_int
    lib.bp_device_count.argtypes = []
    lib.bp_device_count.restype = C.c_int
    lib.bp_version.argtypes = []
    lib.bp_version.restype = C.c_char_p
    lib.bp_note_params_default.argtypes = [C.POINTER(bp_note_params)]
    lib.bp_note_params_default.restype = None
    lib.bp_notes_decode.argtypes = [
        vp, vp, vp, i64, C.POINTER(bp_note_params), vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)
    ]
    lib.bp_notes_decode.restype = C.c_int
    lib.bp_note_candidates.argtypes = [vp, vp, vp, vp, i64, C.POINTER(bp_note_params), C.c_int, vp, vp, vp, C.POINTER(C.c_int)]
    lib.bp_note_candidates.restype = C.c_int
    lib.bp_infer_pcm_raw_candidates.argtypes = [
        vp, vp, C.c_int, i64, C.c_int, C.c_int, C.POINTER(bp_note_params), vp, vp, vp, C.POINTER(C.c_int)
    ]
    lib.bp_infer_pcm_raw_candidates.restype = C.c_int
    lib.bp_track_maps.argtypes = [vp, i64, vp, vp, vp, C.c_int]
    lib.bp_track_maps.restype = C.c_int
    lib.bp_notes_decode_candidates.argtypes = [
        vp, vp, vp, i64, C.POINTER(bp_note_params), vp, i64, vp, i64, C.POINTER(i64), C.POINTER(i64)
    ]
    lib.bp_notes_decode_candidates.restype = C.c_int
    lib.bp_notes_last_error.argtypes = []
    lib.bp_notes_last_error.restype = C.c_char_p
    pi = C.POINTER(C.c_int)
    lib.bp_flac_info.argtypes = [C.c_char_p, C.c_size_t, pi, pi, pi, C.POINTER(i64)]
    lib.bp_flac_info.restype = C.c_int
    lib.bp_flac_decode.argtypes = [C.c_char_p, C.c_size_t, fp, i64, C.POINTER(i64)]
    lib.bp_flac_decode.restype = C.c_int
    lib.bp_audio_last_error.argtypes = []
    lib.bp_audio_last_error.restype = C.c_char_p
    lib.bp_transcribe_params_default.argtypes = [C.POINTER(bp_transcribe_params)]
    lib.bp_transcribe_params_default.restype = None
    lib.bp_transcribe_files.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(C.c_char_p), i64, C.c_char_p,
                                        C.POINTER(bp_transcribe_params), C.POINTER(bp_file_report)]
    lib.bp_transcribe_files.restype = C.c_int
    lib.bp_notes_to_midi.argtypes = [C.c_void_p, i64, C.c_void_p, C.c_int, C.c_double, C.c_void_p, i64]
    lib.bp_notes_to_midi.restype = i64
    lib.bp_notes_to_csv.argtypes = [C.c_void_p, i64, C.c_void_p, C.c_void_p, i64]
    lib.bp_notes_to_csv.restype = i64
    lib.bp_wav_info.argtypes = [C.c_char_p, C.c_size_t, pi, pi, pi, C.POINTER(i64)]
    lib.bp_wav_info.restype = C.c_int
    lib.bp_wav_decode.argtypes = [C.c_char_p, C.c_size_t, fp, i64, C.POINTER(i64)]
    lib.bp_wav_decode.restype = C.c_int
    lib.bp_files_last_error.argtypes = []
    lib.bp_files_last_error.restype = C.c_char_p
    if path is None:
        _lib = lib
    return lib


def check(lib: C.CDLL, handle, rc: int, what: str) -> None:
    """Map a negative bp_status to the exception the reference's callers would see."""
    if rc == BP_OK:
        return
    msg = lib.bp_last_error(handle)
    text = f"{what}: {_ERR_NAMES.get(rc, rc)}: {msg.decode(errors='replace') if msg else ''}"
    if rc in (-1, -2, -6, -7):  # -7: undecodable audio (what audio.read_audio raises ValueError for)
        raise ValueError(text)  # reference: ValueError for bad model / bad shapes (inference.py:148-154)
    raise NativeLibraryError(text)
