#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2h; mkdir -p $O
NYX_HIP_PROFILE=1 timeout 120 python tools/time_config.py 4 2>&1 | tee $O/cycles_c4.txt
timeout 900 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
