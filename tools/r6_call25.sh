#!/bin/bash
# round 6, GPU call 25: fan-out kernel - column-less waves skip their walk (fan_skip) x sums wave x almanac fan-out with chained attempts
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
V='{"sums":{},"nosums":{"debug_flags":1073741824},"rf_spec_sums":{"role_fanout":1,"debug_flags":268435456},"rf_spec_nosums":{"role_fanout":1,"debug_flags":1342177280}}'
{
  echo "== 640 x 20 min, parity on 64 (fan_skip)"; NYX_HIP_LIB=tools/_bin/libnyx_fan_skip.so timeout 60 python tools/sweep.py 2 640 0.34 "$V" 1 64 || echo "RC $?"
  for lib in "" tools/_bin/libnyx_fan_skip.so; do
    echo "== 1250 x 24 h (digest 1152333ec1b0) lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 1250 24 "$V" 2 || echo "RC $?"
  done
  echo "== 1250 x 3 h prof, fan_skip"; NYX_HIP_LIB=tools/_bin/libnyx_fan_skip.so timeout 300 python tools/sweep.py 2 1250 3 '{"sums_prof":{"profile":1},"rf_spec_sums_prof":{"role_fanout":1,"debug_flags":268435456,"profile":1,"show_sched":1}}' || echo "RC $?"
  for lib in "" tools/_bin/libnyx_fan_skip.so; do
    echo "== 2500 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 2500 24 "$V" || echo "RC $?"
    echo "== 5000 x 24 h lib=[$lib]"; NYX_HIP_LIB=$lib timeout 300 python tools/sweep.py 2 5000 24 "$V" || echo "RC $?"
  done
} > gpurun_out/r6_call25.log 2>&1
grep -v "hwave" gpurun_out/r6_call25.log | tail -120
