#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2x
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_frame_swap.py tests/test_gpu_traj.py tests/test_abi.py -m gpu -x -q --timeout 200 -s > $O/pytest.log 2>&1; echo "pytest rc $?"; grep -E "swap:|passed|failed|Error|error" $O/pytest.log | head -20
