#!/bin/bash
# round 6, GPU call 24: the sums wave again, its six values in the drag rows of the perturbation buffers (no LDS of its own: the padded ephemeris records stay in LDS)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== 640 x 20 min, parity on 64"; timeout 60 python tools/sweep.py 2 640 0.34 '{"sums":{},"nosums":{"debug_flags":1073741824}}' 1 64 || echo "RC $?"
  echo "== 1250 x 3 h (digest 48f933abde74)"; timeout 120 python tools/sweep.py 2 1250 3 '{"sums":{},"nosums":{"debug_flags":1073741824},"sums_prof":{"profile":1},"nosums_prof":{"debug_flags":1073741824,"profile":1}}' 1 64 || echo "RC $?"
  echo "== 1250 x 24 h (digest 1152333ec1b0)"; timeout 200 python tools/sweep.py 2 1250 24 '{"sums":{},"nosums":{"debug_flags":1073741824}}' 3 || echo "RC $?"
  echo "== 2500 x 24 h (digest 4f07c40d4cc0)"; timeout 200 python tools/sweep.py 2 2500 24 '{"sums":{},"nosums":{"debug_flags":1073741824}}' 2 || echo "RC $?"
  echo "== 5000 x 24 h (digest 7b4c2bb94c33)"; timeout 200 python tools/sweep.py 2 5000 24 '{"sums":{},"nosums":{"debug_flags":1073741824}}' 2 || echo "RC $?"
  echo "== config 3 (b229b2dfc30a)"; timeout 200 python tools/sweep.py 3 0 0 '{"base":{}}' || echo "RC $?"
  echo "== 10000 x 24 h (0299bb16009e)"; timeout 200 python tools/sweep.py 2 0 0 '{"base":{}}' || echo "RC $?"
} > gpurun_out/r6_call24.log 2>&1
grep -v "hwave" gpurun_out/r6_call24.log | tail -90
timeout 900 python -m pytest tests/test_gpu_fan.py tests/test_gpu_reproducible.py tests/test_gpu_tuning_paths.py -x -q -m gpu 2>&1 | tail -4
