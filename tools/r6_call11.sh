#!/bin/bash
# round 6, GPU call 11: the sums wave of the fan-out mode (fan_sums) - 640 trajectories under a short timeout first (HISTORY r5-32), digests against
# the integrator's own sums (debug_flags 0x40000000), shard sizes over the full day, the cycle table, tests
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r6_call11.log
{
  echo "== 640 x 20 min, parity on 64"; timeout 60 python tools/sweep.py 2 640 0.34 '{"sums":{},"nosums":{"debug_flags":1073741824}}' 1 64 || echo "RC $?"
  echo "== 640 x 1 h, parity on 64 (digest bfab2277febc)"; timeout 60 python tools/sweep.py 2 640 1 '{"sums":{},"nosums":{"debug_flags":1073741824}}' 1 64 || echo "RC $?"
  echo "== 1250 x 3 h (digest 48f933abde74)"; timeout 120 python tools/sweep.py 2 1250 3 '{"sums":{},"nosums":{"debug_flags":1073741824},"prof":{"profile":1},"sums2":{}}' 1 64 || echo "RC $?"
  echo "== 1250 x 24 h (digest 1152333ec1b0)"; timeout 200 python tools/sweep.py 2 1250 24 '{"sums":{},"nosums":{"debug_flags":1073741824},"sums2":{}}' || echo "RC $?"
  echo "== 2500 x 24 h (digest 4f07c40d4cc0)"; timeout 200 python tools/sweep.py 2 2500 24 '{"sums":{},"nosums":{"debug_flags":1073741824}}' || echo "RC $?"
  echo "== 5000 x 24 h (digest 7b4c2bb94c33)"; timeout 200 python tools/sweep.py 2 5000 24 '{"sums":{},"nosums":{"debug_flags":1073741824}}' || echo "RC $?"
  echo "== 10000 x 24 h (digest 0299bb16009e)"; timeout 200 python tools/sweep.py 2 0 0 '{"base":{},"base2":{}}' || echo "RC $?"
  echo "== fallback: no helper answers"; timeout 200 python tools/sweep.py 2 1280 1 '{"base":{},"mute":{"coop_mute":1}}' 1 64 || echo "RC $?"
} > $L 2>&1
grep -v "hwave" $L | tail -80
timeout 1200 python -m pytest tests/test_gpu_fan.py tests/test_gpu_reproducible.py tests/test_gpu_traj.py tests/test_gpu_tuning_paths.py -x -q -m gpu > gpurun_out/r6_call11_tests.log 2>&1
tail -5 gpurun_out/r6_call11_tests.log
