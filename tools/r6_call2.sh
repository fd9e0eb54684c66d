#!/bin/bash
# round 6, GPU call 2: the out-of-line integrator (INTEG_OOL) - small first, under short timeouts (HISTORY r5-32), then digests, cycle table, tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r6_call2.log
{
  echo "== 640 x 1 h, parity on 64"; timeout 120 python tools/sweep.py 2 640 1 '{"base":{}}' 1 64 || echo "RC $?"
  echo "== 10000 x 3 h (digest of round 5: 415720a47b92)"; timeout 200 python tools/sweep.py 2 0 3 '{"base":{},"prof":{"profile":1}}' 1 64 || echo "RC $?"
  echo "== 10000 x 24 h (digest of round 5: 0299bb16009e)"; timeout 300 python tools/sweep.py 2 0 0 '{"base":{},"base2":{},"prof":{"profile":1}}' || echo "RC $?"
  echo "== config 5, 6 h (digest 48dd2474d8d4)"; timeout 300 python tools/sweep.py 5 0 6 '{"base":{},"prof":{"profile":1}}' || echo "RC $?"
  echo "== full chip (digest 5038b80c38e5)"; timeout 200 python tools/sweep.py 2 16384 3 '{"base":{},"prof":{"profile":1}}' || echo "RC $?"
  echo "== coop mute (every owner falls back)"; timeout 200 python tools/sweep.py 2 1280 1 '{"base":{},"mute":{"coop_mute":1},"alone":{"cooperative":0}}' 1 64 || echo "RC $?"
} > $L 2>&1
timeout 1500 python -m pytest tests/test_gpu_reproducible.py tests/test_gpu_coop_contexts.py tests/test_gpu_tuning_paths.py tests/test_gpu_interface.py tests/test_rccl_twin.py -x -q -m gpu > gpurun_out/r6_call2_tests.log 2>&1
tail -5 gpurun_out/r6_call2_tests.log
tail -30 $L
