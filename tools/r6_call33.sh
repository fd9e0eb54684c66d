#!/bin/bash
# round 6, GPU call 33: oracle parity of config 5 over the full 72 h on as many trajectories of the 6 250 as one GPU call holds (the file is
# rewritten after every 256: the call's limit cuts the pass, not the result)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 3480 python tools/full_parity.py 5 0 0 round06_cfg5_parity_72h 256 0 2560 > gpurun_out/r6_call33.log 2>&1
tail -4 gpurun_out/r6_call33.log | cut -c1-600
