#!/bin/bash
# round 6, GPU call 29: the GPU suite and the digests of every configuration on the final build
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== 10000 x 24 h (0299bb16009e)"; timeout 200 python tools/sweep.py 2 0 0 '{"base":{},"base2":{}}' || echo "RC $?"
  echo "== 1250 / 2500 / 5000 x 24 h (1152333ec1b0 / 4f07c40d4cc0 / 7b4c2bb94c33)"; for n in 1250 2500 5000; do timeout 200 python tools/sweep.py 2 $n 24 '{"base":{},"base2":{}}' | grep base; done
  echo "== config 3 (b229b2dfc30a)"; timeout 200 python tools/sweep.py 3 0 0 '{"base":{}}' | grep base
  echo "== config 4 (e203310ebb09)"; timeout 200 python tools/sweep.py 4 0 0 '{"base":{}}' | grep base
  echo "== config 5 6 h (48dd2474d8d4)"; timeout 300 python tools/sweep.py 5 0 6 '{"base":{}}' | grep base
  echo "== full chip (5038b80c38e5)"; timeout 200 python tools/sweep.py 2 16384 3 '{"base":{}}' | grep base
} > gpurun_out/r6_call29.log 2>&1
cat gpurun_out/r6_call29.log
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6_call29_tests.log 2>&1
tail -5 gpurun_out/r6_call29_tests.log
