#!/bin/bash
# round-4 profiles of the four BASELINE configurations (run from the repo root under gpurun)
for c in 2 5 3 4; do
  echo "=== config $c"; CFG=$c STEPS=3 bash tools/profile_round.sh 2>&1 | tail -25
done
