#!/bin/bash
# round 6, GPU call 23: full-ensemble oracle parity of the fan-out shards of configs[1] (every trajectory, 24 h) and of config 3
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== configs[1] shard 1250 x 24 h"; timeout 600 python tools/full_parity.py 2 1250 24 round06_shard1250_full_parity || echo "RC $?"
  echo "== configs[1] shard 2500 x 24 h"; timeout 900 python tools/full_parity.py 2 2500 24 round06_shard2500_full_parity || echo "RC $?"
  echo "== configs[1] shard 5000 x 24 h"; timeout 1200 python tools/full_parity.py 2 5000 24 round06_shard5000_full_parity || echo "RC $?"
  echo "== config 3, 5000 x 30 d"; timeout 600 python tools/full_parity.py 3 0 0 round06_cfg3_full_parity || echo "RC $?"
} > gpurun_out/r6_call23.log 2>&1
tail -40 gpurun_out/r6_call23.log
