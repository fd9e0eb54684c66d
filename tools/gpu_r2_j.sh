#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r2j; mkdir -p $O
NYX_HIP_PROFILE=1 timeout 120 python tools/time_config.py 4 2>&1 | tee $O/cycles_c4.txt
timeout 600 python -m pytest tests/test_gpu_stm_quad.py tests/test_gpu_predict.py tests/test_gpu_headline.py -x -q --timeout 200 2>&1 | tail -5
