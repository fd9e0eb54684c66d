#!/bin/bash
# round 6, GPU call 16: config 3 (47.05 ms in call 15 against 43.9 at the start of the round): the eight-wave kernel of the round's first commit, the current one, and the current one with the two-pow step control
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== config 3, 5000 x 30 d"; bash tools/ab_lib.sh "tools/_bin/libnyx_w8n_old.so - tools/_bin/libnyx_w8n_pow0.so" 3 5000 720
} > gpurun_out/r6_call16.log 2>&1
cat gpurun_out/r6_call16.log
