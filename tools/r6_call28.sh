#!/bin/bash
# round 6, GPU call 28: fan-out kernel - the producer's poll is the fetch (fan_pf) against the in-tree build (batched lead, skip); streamed feed in the helpers
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
V='{"base":{},"hs":{"harmonics_feed":1}}'
{
  echo "== 640 x 20 min, parity on 64 (fan_pf)"; NYX_HIP_LIB=tools/_bin/libnyx_fan_pf.so timeout 60 python tools/sweep.py 2 640 0.34 "$V" 1 64 || echo "RC $?"
  echo "== fallback (fan_pf)"; NYX_HIP_LIB=tools/_bin/libnyx_fan_pf.so timeout 100 python tools/sweep.py 2 1280 1 '{"base":{},"mute":{"coop_mute":1}}' 1 64 || echo "RC $?"
  for rep in 1 2 3; do for lib in "" tools/_bin/libnyx_fan_pf.so; do
    echo "== 1250 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 1250 24 "$V" | grep "base\|hs "
  done; done
  for lib in "" tools/_bin/libnyx_fan_pf.so; do
    echo "== 2500 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 2500 24 "$V" | grep "base\|hs "
    echo "== 5000 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 5000 24 "$V" | grep "base\|hs "
  done
} > gpurun_out/r6_call28.log 2>&1
cat gpurun_out/r6_call28.log
