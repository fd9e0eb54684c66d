#!/bin/bash
export TMPDIR=/tmp
for lib in "" gpurun_variants/libnyx_OLD_TAIL.so gpurun_variants/libnyx_NO_SKIP.so; do
  echo "=== lib: ${lib:-current}"
  for args in "8 4" "21 4" "21 16" "8 0"; do
    NYX_HIP_LIB=${lib:+$PWD/$lib} timeout 60 python tools/bisect_quad.py $args 2>&1 | grep "deg\|Error\|error" ; echo "  rc $?"
  done
done
