#!/bin/bash
# round 6, GPU call 26: fan-out kernel - the lead helper collects the other parts' sums four parts per round trip (fan_lead, with the column-less waves' skip)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
V='{"sums":{},"nosums":{"debug_flags":1073741824},"rf_spec_sums":{"role_fanout":1,"debug_flags":268435456},"rf_spec_nosums":{"role_fanout":1,"debug_flags":1342177280}}'
{
  echo "== 640 x 20 min, parity on 64 (fan_lead)"; NYX_HIP_LIB=tools/_bin/libnyx_fan_lead.so timeout 60 python tools/sweep.py 2 640 0.34 "$V" 1 64 || echo "RC $?"
  echo "== 1250 x 1 h, fallback"; NYX_HIP_LIB=tools/_bin/libnyx_fan_lead.so timeout 100 python tools/sweep.py 2 1280 1 '{"base":{},"mute":{"coop_mute":1}}' 1 64 || echo "RC $?"
  for lib in "" tools/_bin/libnyx_fan_lead.so; do
    echo "== 1250 x 24 h (digest 1152333ec1b0) lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 1250 24 "$V" 2 || echo "RC $?"
  done
  echo "== 1250 x 3 h prof, fan_lead"; NYX_HIP_LIB=tools/_bin/libnyx_fan_lead.so timeout 300 python tools/sweep.py 2 1250 3 '{"sums_prof":{"profile":1},"nosums_prof":{"debug_flags":1073741824,"profile":1}}' || echo "RC $?"
  for lib in "" tools/_bin/libnyx_fan_lead.so; do
    echo "== 2500 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 2500 24 "$V" || echo "RC $?"
    echo "== 5000 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 5000 24 '{"sums":{},"nosums":{"debug_flags":1073741824}}' || echo "RC $?"
  done
} > gpurun_out/r6_call26.log 2>&1
grep -v "hwave\|    wave  [3-9]\|    wave 1[0-4]" gpurun_out/r6_call26.log | tail -90
