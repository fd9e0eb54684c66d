#!/bin/bash
# round 6, GPU call 35: oracle parity of config 5 over the full 72 h, trajectories 2 560 .. 5 119 of the 6 250 (the second hour of host threads)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 3480 python tools/full_parity.py 5 0 0 round06_cfg5_parity_72h_b 256 2560 2560 > gpurun_out/r6_call35.log 2>&1
tail -3 gpurun_out/r6_call35.log | cut -c1-400
