#!/bin/bash
# round 6, GPU call 9: step control - one pow in front of the accept / reject branches, the sums' loop unrolled; the attempt boundary in pieces
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r6_call9.log
{
  echo "== 640 x 1 h, parity on 64 (digest e51455ac59db)"
  for v in pow0 pow1 pow1u4; do echo "-- $v"; NYX_HIP_LIB=tools/_bin/libnyx_$v.so timeout 120 python tools/sweep.py 2 640 1 '{"base":{}}' 1 64 || echo "RC $?"; done
  echo "== 10000 x 3 h (digest 415720a47b92), pieces"
  for v in pow0 pow1; do echo "-- $v"; NYX_HIP_LIB=tools/_bin/libnyx_$v.so timeout 200 python tools/sweep.py 2 0 3 '{"base":{},"prof":{"profile":1}}' || echo "RC $?"; done
  echo "== configs[1] 10000 x 24 h"; bash tools/ab_lib.sh "tools/_bin/libnyx_pow0.so tools/_bin/libnyx_pow1.so tools/_bin/libnyx_pow1u4.so" 2 10000 24
  echo "== 1250 x 24 h (fan-out: the fan kernel is the in-tree object in all three - control)"; bash tools/ab_lib.sh "tools/_bin/libnyx_pow0.so tools/_bin/libnyx_pow1.so" 2 1250 24
  echo "== 24 h digests (0299bb16009e)"
  for v in pow1 pow1u4; do NYX_HIP_LIB=tools/_bin/libnyx_$v.so timeout 200 python tools/sweep.py 2 0 0 '{"base":{}}' || echo "RC $?"; done
} > $L 2>&1
tail -60 $L
