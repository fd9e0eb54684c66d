#!/bin/bash
# round 6, GPU call 6: fan-out shard with the almanac / perturbation duties fanned out over the owner's idle waves
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== 1250 x 3 h"; timeout 300 python tools/sweep.py 2 1250 3 '{"fan":{},"fan_rf":{"role_fanout":1},"fan_rf_spec":{"role_fanout":1,"debug_flags":33554432},"fan_rf_spec_prof":{"role_fanout":1,"debug_flags":33554432,"profile":1,"show_sched":1}}' 1 64 || echo "RC $?"
  echo "== 5000 x 3 h"; timeout 300 python tools/sweep.py 2 5000 3 '{"fan":{},"fan_rf":{"role_fanout":1},"fan_rf_spec":{"role_fanout":1,"debug_flags":33554432}}' 1 64 || echo "RC $?"
} > gpurun_out/r6_call6.log 2>&1
cat gpurun_out/r6_call6.log
