#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2e
mkdir -p $O
NYX_HIP_PROFILE=1 timeout 300 python tools/time_config.py 4 > $O/cycles_c4.txt 2>&1; cat $O/cycles_c4.txt
for c in 2 5; do timeout 300 python tools/time_config.py $c 0 6 2>&1 | grep "config\|weights"; done
timeout 300 python tools/time_config.py 3 2>&1 | grep "config\|weights"
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
