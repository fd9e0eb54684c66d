#!/bin/bash
# round 6, GPU call 19: config 4 - the time update's loads at once (a), + the step control's sums four stages per batch (b), + one pow (c); stmq_seg = before
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  for v in stmq_seg stmq_a stmq_b stmq_c; do echo "== $v (digest e203310ebb09)"; NYX_HIP_LIB=tools/_bin/libnyx_$v.so timeout 300 python tools/sweep.py 4 0 0 '{"base":{},"prof":{"profile":1},"base2":{}}' | grep -v "wave" || echo "RC $?"; done
  echo "== interleaved"; bash tools/ab_lib.sh "tools/_bin/libnyx_stmq_seg.so tools/_bin/libnyx_stmq_a.so tools/_bin/libnyx_stmq_b.so tools/_bin/libnyx_stmq_c.so" 4 1000 1
} > gpurun_out/r6_call19.log 2>&1
cat gpurun_out/r6_call19.log
