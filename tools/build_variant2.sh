#!/bin/bash
# tools/build_variant2.sh <name> "<flags of the product object>" ["<flags of the accounting twin>"]: recompile one kernel translation unit
# (SRC, default propagate_kernel.hip) - the product object and, when a third argument is given, its NYX_PROF twin - with extra flags and
# link them with the other objects of the in-tree build into tools/_bin/libnyx_<name>.so (select with NYX_HIP_LIB; tools/ab_lib.sh).
set -e
cd "$(dirname "$0")/.."
name=$1
mkdir -p tools/_bin
B=nyx_amd/csrc/build
SRC=${SRC:-propagate_kernel.hip}
CC="hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value"
$CC $2 -c nyx_amd/csrc/$SRC -o tools/_bin/pk_$name.o &
p1=$!
excl="/$SRC.o$"
twin=""
if [ -n "${3+x}" ]; then
  $CC -DNYX_PROF=1 $3 -c nyx_amd/csrc/$SRC -o tools/_bin/pk_$name.prof.o &
  p2=$!
  wait $p2
  twin=tools/_bin/pk_$name.prof.o
  excl="/$SRC\(.prof\)\?.o$"
fi
wait $p1
objs=$(ls $B/*.o | grep -v "$excl")
hipcc --offload-arch=gfx950 -shared -fPIC tools/_bin/pk_$name.o $twin $objs -lz -o tools/_bin/libnyx_$name.so
echo built tools/_bin/libnyx_$name.so
