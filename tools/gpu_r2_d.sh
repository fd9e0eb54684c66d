#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2d
mkdir -p $O
for c in 4 5 2; do
  H=0; [ $c = 5 ] && H=6; [ $c = 2 ] && H=6
  NYX_HIP_PROFILE=1 timeout 300 python tools/time_config.py $c 0 $H > $O/cycles_c$c.txt 2>&1; cat $O/cycles_c$c.txt
  NYX_HIP_CALIBRATE=0 timeout 300 python tools/time_config.py $c 0 $H 2>&1 | grep "config\|weights"
done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
