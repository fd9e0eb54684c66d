#!/bin/bash
# round 6, GPU call 5: same-box A/B of the out-of-line stage sums (in-tree) against the inline sums (tools/_bin/libnyx_inl.so)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== configs[1] 10000 x 24 h"; bash tools/ab_lib.sh "- tools/_bin/libnyx_inl.so" 2 10000 24
  echo "== 1250 x 24 h (fan-out)"; bash tools/ab_lib.sh "- tools/_bin/libnyx_inl.so" 2 1250 24
  echo "== 5000 x 24 h (fan-out)"; bash tools/ab_lib.sh "- tools/_bin/libnyx_inl.so" 2 5000 24
  echo "== full chip 16384 x 3 h"; bash tools/ab_lib.sh "- tools/_bin/libnyx_inl.so" 2 16384 3
} > gpurun_out/r6_call5.log 2>&1
cat gpurun_out/r6_call5.log
