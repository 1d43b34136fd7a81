#!/bin/bash
# build_variant.sh NAME SOURCE.hip [-DFLAG ...]: a copy of the library with ONE source compiled with extra flags, as
# basic_pitch_amd/lib/var_NAME.so (select it with BASIC_PITCH_AMD_LIB=...): several A/B variants in one GPU call.
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
lib=$root/basic_pitch_amd/lib
python -m basic_pitch_amd.build > /dev/null
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -pragma-unroll-threshold=400000 -fno-slp-vectorize "$@" -c $root/basic_pitch_amd/csrc/$src -o /tmp/var_$name.o 2>&1 | grep -A4 "error" || true
objs=$(ls $lib/obj/*.o | grep -v "/$src.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $lib/var_$name.so $objs /tmp/var_$name.o
echo $lib/var_$name.so
