#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2l; mkdir -p $O
NYX_HIP_PROFILE=1 timeout 120 python tools/time_config.py 4 2>&1 | grep -v amdgpu | tee $O/cycles_c4.txt | head -22
NYX_HIP_PIPE=0 timeout 120 python tools/time_config.py 4 2>&1 | grep config
timeout 400 python -m pytest tests/test_gpu_stm_quad.py tests/test_gpu_predict.py tests/test_gpu_headline.py -x -q --timeout 100 2>&1 | tail -8
