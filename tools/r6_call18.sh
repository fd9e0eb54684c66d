#!/bin/bash
# round 6, GPU call 18: config 4 - the attempt / segment boundary of the integrator wave in pieces (NYX_SEG_PROF twin of the quad STM kernel)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== config 4 (digest e203310ebb09)"; NYX_HIP_LIB=tools/_bin/libnyx_stmq_seg.so timeout 300 python tools/sweep.py 4 0 0 '{"base":{},"prof":{"profile":1}}' || echo "RC $?"
} > gpurun_out/r6_call18.log 2>&1
cat gpurun_out/r6_call18.log
