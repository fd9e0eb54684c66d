#!/bin/bash
# round 6, GPU call 12: fan-out shard with the sums wave - almanac / perturbation duties fanned out over the owner's idle waves, with and without chained attempts
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== 1250 x 3 h"; timeout 300 python tools/sweep.py 2 1250 3 '{"sums":{},"rf":{"role_fanout":1},"rf_spec":{"role_fanout":1,"debug_flags":268435456},"rf_spec_nosums":{"role_fanout":1,"debug_flags":1342177280},"sums_prof":{"profile":1},"rf_prof":{"role_fanout":1,"profile":1,"show_sched":1},"rf_spec_prof":{"role_fanout":1,"debug_flags":268435456,"profile":1,"show_sched":1}}' 1 64 || echo "RC $?"
  echo "== 2500 x 3 h"; timeout 300 python tools/sweep.py 2 2500 3 '{"sums":{},"rf":{"role_fanout":1},"rf_spec":{"role_fanout":1,"debug_flags":268435456}}' 1 64 || echo "RC $?"
  echo "== 5000 x 3 h"; timeout 300 python tools/sweep.py 2 5000 3 '{"sums":{},"rf":{"role_fanout":1},"rf_spec":{"role_fanout":1,"debug_flags":268435456}}' 1 64 || echo "RC $?"
} > gpurun_out/r6_call12.log 2>&1
grep -v hwave gpurun_out/r6_call12.log
