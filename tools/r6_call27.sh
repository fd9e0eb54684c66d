#!/bin/bash
# round 6, GPU call 27: fan-out kernel with the batched lead - with and without the column-less waves' skip (three interleaved rounds)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  for rep in 1 2 3; do for lib in tools/_bin/libnyx_fan_lead.so tools/_bin/libnyx_fan_lead_noskip.so; do
    echo "== 1250 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 1250 24 '{"nosums":{"debug_flags":1073741824},"sums":{}}' | grep "sums"
  done; done
  for lib in tools/_bin/libnyx_fan_lead.so tools/_bin/libnyx_fan_lead_noskip.so; do
    echo "== 2500 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 2500 24 '{"nosums":{"debug_flags":1073741824},"sums":{}}' | grep "sums"
    echo "== 5000 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 5000 24 '{"nosums":{"debug_flags":1073741824},"sums":{}}' | grep "sums"
  done
} > gpurun_out/r6_call27.log 2>&1
cat gpurun_out/r6_call27.log
