#!/bin/bash
# round-6 profiles (run from the repo root under gpurun): the four BASELINE configurations and the full-chip launch of configs[1]'s force model
export TMPDIR=/tmp
mkdir -p gpurun_out/prof_cfg2
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/prof_cfg2/bench_with_others.json 2> gpurun_out/prof_cfg2/bench_with_others.err
tail -c 400 gpurun_out/prof_cfg2/bench_with_others.json; echo
for c in 2 5 3 4; do
  echo "=== config $c"; CFG=$c STEPS=3 bash tools/profile_round.sh 2>&1 | tail -25
done
echo "=== full chip"; CFG=2 STEPS=3 TAG=fullchip EXTRA="--n 16384 --hours 3" bash tools/profile_round.sh 2>&1 | tail -25
