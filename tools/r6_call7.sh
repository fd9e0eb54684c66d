#!/bin/bash
# round 6, GPU call 7: the covariance-mapping loop in ONE launch (segment_update) against the launch-per-segment loop (0x20000000)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== config 4: fused vs per-segment (digest e203310ebb09)"; timeout 200 python tools/sweep.py 4 0 0 '{"fused":{},"perseg":{"debug_flags":536870912},"fused2":{},"perseg2":{"debug_flags":536870912}}' || echo "RC $?"
  echo "== D3 layout"; timeout 200 python tools/sweep.py 4 0 0 '{"fused_d3":{"stm_quad":0},"perseg_d3":{"stm_quad":0,"debug_flags":536870912}}' || echo "RC $?"
  timeout 900 python -m pytest tests/test_gpu_predict.py tests/test_gpu_stm_quad.py tests/test_gpu_stm_textbook.py -q -m gpu 2>&1 | tail -15
} > gpurun_out/r6_call7.log 2>&1
cat gpurun_out/r6_call7.log
