#!/bin/bash
# A/B of builds of libnyx_hip.so on the same GPU box: tools/ab_lib.sh "<lib1.so> <lib2.so> ..." [config n hours]; "-" = the in-tree build.
# (interleaved runs, three each; the boxes of the pool differ by several percent, runs on one box by ~0.5 %)
libs=$1; shift
cfg=${1:-2}; n=${2:-10000}; hours=${3:-3}
for rep in 1 2 3; do
  for lib in $libs; do
    l=$([ "$lib" = "-" ] && echo "" || echo $lib)
    NYX_HIP_LIB=$l timeout 120 python tools/time_config.py $cfg $n $hours 2>&1 | grep device | sed "s|^|$lib: |; s/evals .*per trajectory-lane, //; s/acc .*algorithmic/alg/"
  done
done
