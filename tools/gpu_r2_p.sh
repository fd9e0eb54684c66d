#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2p
mkdir -p $O
for sp in 1 0; do
NYX_HIP_SPEC=$sp NYX_HIP_PROFILE=1 timeout 200 python tools/time_gpu.py 10000 6 > $O/cycles_spec$sp.txt 2>&1; cat $O/cycles_spec$sp.txt | grep -v amdgpu.ids
done
