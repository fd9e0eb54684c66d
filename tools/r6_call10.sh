#!/bin/bash
# round 6, GPU call 10: step control out of line (integ_step) - parity first, digests, the cycle table, A/B against the inline step control, tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r6_call10.log
{
  echo "== 640 x 1 h, parity on 64 (digest bfab2277febc)"; timeout 120 python tools/sweep.py 2 640 1 '{"base":{},"nofan":{"debug_flags":134217728}}' 1 64 || echo "RC $?"
  echo "== 10000 x 3 h (digest 415720a47b92)"; timeout 200 python tools/sweep.py 2 0 3 '{"base":{},"prof":{"profile":1}}' 1 64 || echo "RC $?"
  echo "== configs[1] 10000 x 24 h"; bash tools/ab_lib.sh "tools/_bin/libnyx_pow1.so -" 2 10000 24
  echo "== 1250 x 24 h (fan-out)"; bash tools/ab_lib.sh "tools/_bin/libnyx_pow1.so -" 2 1250 24
  echo "== 24 h digest (0299bb16009e)"; timeout 200 python tools/sweep.py 2 0 0 '{"base":{}}' || echo "RC $?"
  echo "== 1250 x 3 h prof (digest 48f933abde74)"; timeout 200 python tools/sweep.py 2 1250 3 '{"base":{},"prof":{"profile":1}}' || echo "RC $?"
  echo "== config 5 6 h (48dd2474d8d4)"; timeout 300 python tools/sweep.py 5 0 6 '{"base":{}}' || echo "RC $?"
} > $L 2>&1
timeout 1500 python -m pytest tests/test_gpu_traj.py tests/test_gpu_parity.py tests/test_gpu_reproducible.py tests/test_gpu_fan.py tests/test_gpu_events.py -x -q -m gpu > gpurun_out/r6_call10_tests.log 2>&1
tail -5 gpurun_out/r6_call10_tests.log
grep -v "hwave" $L | tail -70
