#!/bin/bash
# round 6, GPU call 32: full-ensemble oracle parity of config 5 (6 250 x 72 h, every trajectory; ~2.4 h of the box's 256 host threads)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 10500 python tools/full_parity.py 5 0 0 round06_cfg5_full_parity 256 > gpurun_out/r6_call32.log 2>&1
tail -5 gpurun_out/r6_call32.log
