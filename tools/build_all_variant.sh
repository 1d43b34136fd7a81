#!/bin/bash
# build_all_variant.sh NAME [-DFLAG | -fflag ...]: a copy of the library with EVERY source compiled with the extra flags,
# as basic_pitch_amd/lib/var_NAME.so (select it with BASIC_PITCH_AMD_LIB=...): whole-library A/B runs in one GPU call.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/basic_pitch_amd/lib
mkdir -p /tmp/var_$name
pids=()
for src in $root/basic_pitch_amd/csrc/*.hip $root/basic_pitch_amd/csrc/*.cpp; do
  b=$(basename $src)
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=400000 -fno-slp-vectorize "$@" -c $src -o /tmp/var_$name/$b.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/var_$name.so /tmp/var_$name/*.o
echo $lib/var_$name.so
