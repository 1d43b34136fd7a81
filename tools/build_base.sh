#!/bin/bash
# Build the library of a git revision (default HEAD) into tools/ubench/bin/libbase.so (git-ignored, travels with
# gpurun) so that a change can be measured against it on the SAME box: BASIC_PITCH_AMD_LIB=.../libbase.so python bench.py
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/bp_base && mkdir -p /tmp/bp_base "$ROOT/tools/ubench/bin"
git -C "$ROOT" archive "$REV" basic_pitch_amd/csrc include | tar -x -C /tmp/bp_base
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-function \
  -mllvm -pragma-unroll-threshold=400000 -o "$ROOT/tools/ubench/bin/libbase.so" \
  /tmp/bp_base/basic_pitch_amd/csrc/*.hip /tmp/bp_base/basic_pitch_amd/csrc/note_decode.cpp
ls -la "$ROOT/tools/ubench/bin/libbase.so"
