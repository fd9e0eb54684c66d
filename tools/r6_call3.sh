#!/bin/bash
# round 6, GPU call 3: the fan-out mode (dedicated helper workgroups for small shards) - small and short first, then the shard sizes
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r6_call3.log
{
  echo "== 64 x 20 min, parity"; timeout 120 python tools/sweep.py 2 64 0.34 '{"fan":{"show_sched":1},"nofan":{"debug_flags":16777216}}' 1 64 || echo "RC $?"
  echo "== 1250 x 1 h, parity on 64"; timeout 120 python tools/sweep.py 2 1250 1 '{"fan":{"show_sched":1},"nofan":{"debug_flags":16777216},"alone":{"cooperative":0}}' 1 64 || echo "RC $?"
  for n in 1250 2500 5000; do echo "== $n x 3 h"; timeout 200 python tools/sweep.py 2 $n 3 '{"fan":{"show_sched":1,"profile":1},"fanp":{},"nofan":{"debug_flags":16777216}}' 1 64 || echo "RC $?"; done
  for n in 1250 2500 5000 10000; do echo "== $n x 24 h"; timeout 300 python tools/sweep.py 2 $n 24 '{"fan":{}}' || echo "RC $?"; done
  echo "== config 4 / config 3 (unchanged kernels?)"; timeout 120 python tools/sweep.py 4 0 0 '{"base":{}}'; timeout 120 python tools/sweep.py 3 0 0 '{"base":{}}'
} > $L 2>&1
tail -70 $L
