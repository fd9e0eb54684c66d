#!/bin/bash
set -u
export TMPDIR=/tmp
for cf in 0.26 0.30 0.33 0.36; do
echo "== coop_frac $cf"
NYX_HIP_COOP_FRAC=$cf timeout 200 python tools/time_gpu.py 10000 6 2>&1 | grep -E "kernel"
done
