#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2b
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stm_quad.py -x -q -s > $O/quad.log 2>&1; echo "quad rc $?"; tail -15 $O/quad.log
for q in 0 1; do NYX_HIP_STM_QUAD=$q NYX_HIP_PROFILE=1 timeout 200 python tools/time_config.py 4 > $O/cycles_c4_q$q.txt 2>&1; cat $O/cycles_c4_q$q.txt; done
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log
