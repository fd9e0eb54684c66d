#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2f
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_stm_quad.py tests/test_gpu_predict.py -q -s 2>&1 | tail -40
