#!/bin/bash
# round 6, GPU call 8: jobs in flight inside a helper (HELPER_SLOTS 2 / 3 / 4) with the out-of-line integrator (same box, interleaved)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== configs[1] 10000 x 24 h"; bash tools/ab_lib.sh "- tools/_bin/libnyx_ns3.so tools/_bin/libnyx_ns4.so" 2 10000 24
  echo "== config 5, 6 h"; bash tools/ab_lib.sh "- tools/_bin/libnyx_ns3.so" 5 6250 6
} > gpurun_out/r6_call8.log 2>&1
cat gpurun_out/r6_call8.log
