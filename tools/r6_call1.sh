#!/bin/bash
# round 6, GPU call 1: what a rank runs at the shard sizes the metric names (strong scaling of configs[1] / config 3),
# the RCCL / host facts the C++ collective twin needs, then the full-ensemble oracle parity pass of configs[1].
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== host"; nproc; free -g | head -2
  ls /opt/rocm/lib/librccl* 2>&1 | head; ls /opt/rocm/include/rccl 2>&1 | head
  echo "== shard sizes, configs[1] x 24 h"
  for n in 10000 5000 2500 1250; do timeout 300 python tools/sweep.py 2 $n 24 '{"base":{"show_sched":1}}'; done
  echo "== shard sizes, config 3 x 30 d"
  for n in 5000 2500 1250 625; do timeout 120 python tools/sweep.py 3 $n 0 '{"base":{}}'; done
  echo "== config 4 / 5 reference"
  timeout 120 python tools/sweep.py 4 0 0 '{"base":{}}'
} > gpurun_out/r6_call1.log 2>&1
timeout 1500 python tools/full_parity.py 2 0 0 round06_cfg2_full_parity > gpurun_out/r6_full_parity.log 2>&1
tail -3 gpurun_out/r6_full_parity.log
