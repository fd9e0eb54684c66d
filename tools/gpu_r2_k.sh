#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2k; mkdir -p $O
for c in 4 3; do NYX_HIP_PROFILE=1 timeout 120 python tools/time_config.py $c 2>&1 | grep -v amdgpu | tee $O/cycles_c$c.txt | head -22; done
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
