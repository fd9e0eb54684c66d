#!/bin/bash
# round 6, GPU call 36: oracle parity of config 5 over the full 72 h, the last 1 130 trajectories (5 120 .. 6 249): the pass is whole with this
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1750 python tools/full_parity.py 5 0 0 round06_cfg5_parity_72h_c 256 5120 1130 > gpurun_out/r6_call36.log 2>&1
tail -2 gpurun_out/r6_call36.log | cut -c1-300
