#!/bin/bash
# round 6, GPU call 30: the plain-field specialisation (no drag / tides / second field as compile-time constants) of the fan-out kernel and of the headline kernel
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== 1250 x 24 h (1152333ec1b0)"; bash tools/ab_lib.sh "- tools/_bin/libnyx_fan_pf1.so" 2 1250 24
  echo "== 2500 x 24 h"; bash tools/ab_lib.sh "- tools/_bin/libnyx_fan_pf1.so" 2 2500 24
  echo "== 10000 x 24 h"; bash tools/ab_lib.sh "- tools/_bin/libnyx_pk_pf1.so" 2 10000 24
  echo "== full chip"; bash tools/ab_lib.sh "- tools/_bin/libnyx_pk_pf1.so" 2 16384 3
  echo "== digests"; NYX_HIP_LIB=tools/_bin/libnyx_fan_pf1.so timeout 200 python tools/sweep.py 2 1250 24 '{"base":{}}' | grep base; NYX_HIP_LIB=tools/_bin/libnyx_pk_pf1.so timeout 200 python tools/sweep.py 2 0 0 '{"base":{}}' | grep base
} > gpurun_out/r6_call30.log 2>&1
cat gpurun_out/r6_call30.log
