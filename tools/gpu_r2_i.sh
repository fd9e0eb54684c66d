#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
NYX_HIP_PROFILE=1 timeout 120 python tools/time_config.py 4 2>&1 | tee $O/cycles_c4.txt
for c in 2 5; do NYX_HIP_PROFILE=1 timeout 300 python tools/time_config.py $c 0 6 2>&1 | tee $O/cycles_c$c.txt | grep "config\|weights"; done
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
