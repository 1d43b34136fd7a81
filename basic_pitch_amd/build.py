"""In-tree build of libbasicpitch_amd.so (hipcc, gfx950 only) — no JIT cache, no site-packages.

hipcc cross-compiles without a GPU, so this runs in the dev container as well as on the GPU box
(where the prebuilt .so shipped with the snapshot is normally reused).
"""
from __future__ import annotations

import os
import shutil
import subprocess
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_DIR = os.path.join(PKG_DIR, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libbasicpitch_amd.so")
HEADER = os.path.join(PKG_DIR, "..", "include", "basic_pitch_amd.h")

SOURCES = [
    "bp_api.hip",
    "cqt_pyramid.hip",
    "cqt_filterbank.hip",
    "cqt_planes.hip",
    "conv_contour1.hip",
    "conv_contour_rim.hip",
    "conv_contour_rim_march.hip",
    "conv_contour_march.hip",
    "conv_contour2.hip",
    "conv_stride3.hip",
    "conv_heads.hip",
    "conv_branch.hip",
    "note_march16.hip",
    "onset_march16.hip",
    "note_device.hip",
    "audio_ingest.hip",
    "flac_device.hip",
    "note_decode.cpp",
    "flac_decode.cpp",
    "file_pipeline.cpp",
]
# Superseded decompositions of a kernel, kept for comparison runs: compiled only into the A/B library
# (`build_library(ab=True)` -> lib/libbasicpitch_amd_ab.so, every source with -DBP_AB_KERNELS, which also turns the BP_*
# environment switches on: csrc/bp_common.h ab_env).  The product library carries neither.
AB_SOURCES = [
    "conv_contour_direct.hip",  # exact 8-channel conv1 (BP_RIM=exact, BP_CONV1=full), round-2 folded conv1 (BP_CONV1=rounds)
    "onset_march.hip",          # onset march on 32x32x16 (BP_ONSET=march32)
    "note_march.hip",           # note march on 32x32x16 (BP_NOTE=march32)
    "conv_contour_fold_mx.hip", # contour conv1 of the fp8-corrections mode (BP_FLAG_FP8_CORRECTIONS: A/B library only since round 6)
]
AB_LIB_PATH = os.path.join(LIB_DIR, "libbasicpitch_amd_ab.so")


def _sources(ab: bool = False) -> List[str]:
    names = SOURCES + (AB_SOURCES if ab else [])
    return [os.path.join(CSRC, s) for s in names if os.path.exists(os.path.join(CSRC, s))]


def _stale(ab: bool = False) -> bool:
    lib = AB_LIB_PATH if ab else LIB_PATH
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = _sources(ab) + [os.path.join(CSRC, "bp_common.h"), HEADER]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def find_hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the MI355X library cannot be built (there is no CPU fallback)")


FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-Wall",
    "-Wno-unused-function",
    # the fully unrolled 63-k-step MFMA loops exceed clang's default size limit for `#pragma unroll`
    "-mllvm",
    "-pragma-unroll-threshold=400000",
    # no SLP packing of adjacent f32 operations into v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: beside matrix instructions
    # a packed-f32 operation costs ~3.5 x a plain VALU operation (profiles/r04_ubench_shadow.md); note march -5 %,
    # filterbank -3 %, onset -2 % (round 4, same-box A/B)
    "-fno-slp-vectorize",
]


def build_library(force: bool = False, verbose: bool = False, ab: bool = False) -> str:
    """Compile every HIP source for gfx950 into basic_pitch_amd/lib/libbasicpitch_amd.so.

    One object per source under lib/obj/ (compiled in parallel, rebuilt only when the source or a shared header is
    newer), then one link: a kernel edit costs one file's compile time.  `ab=True` builds the A/B library instead
    (lib/libbasicpitch_amd_ab.so, objects under lib/obj_ab/): see AB_SOURCES."""
    lib_path = AB_LIB_PATH if ab else LIB_PATH
    if not force and not _stale(ab):
        return lib_path
    from concurrent.futures import ThreadPoolExecutor

    hipcc = find_hipcc()
    obj_dir = os.path.join(LIB_DIR, "obj_ab" if ab else "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, "bp_common.h"), HEADER]
    t_hdr = max(os.path.getmtime(h) for h in headers if os.path.exists(h))
    flags = FLAGS + (["-DBP_AB_KERNELS"] if ab else [])

    def compile_one(src: str) -> str:
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(src), t_hdr):
            return obj
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n" + res.stdout + res.stderr)
        if verbose and res.stderr.strip():
            print(res.stderr)
        return obj

    sources = _sources(ab)
    # objects of sources that left the list (a renamed or retired file) must not reach tools that link lib/obj/*.o
    keep = {os.path.basename(s) + ".o" for s in sources}
    for stale in os.listdir(obj_dir):
        if stale.endswith(".o") and stale not in keep:
            os.remove(os.path.join(obj_dir, stale))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(compile_one, sources))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path + ".tmp"] + objs
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(lib_path + ".tmp", lib_path)
    return lib_path


if __name__ == "__main__":
    import sys

    print(build_library(force=True, verbose=True, ab="--ab" in sys.argv))
