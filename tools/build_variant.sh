#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: recompile one kernel translation unit (SRC, default the sixteen-wave plain kernel
# propagate_kernel.hip) with extra flags and link it with the other objects of the in-tree build into tools/_bin/libnyx_<name>.so
# (select with NYX_HIP_LIB; tools/ab_lib.sh).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_bin
B=nyx_amd/csrc/build
SRC=${SRC:-propagate_kernel.hip}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value "$@" -c nyx_amd/csrc/$SRC -o tools/_bin/pk_$name.o
objs=$(ls $B/*.o | grep -v "/$SRC.o$")
hipcc --offload-arch=gfx950 -shared -fPIC tools/_bin/pk_$name.o $objs -lz -o tools/_bin/libnyx_$name.so
echo built tools/_bin/libnyx_$name.so
