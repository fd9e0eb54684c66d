#!/bin/bash
# round 6, GPU call 17: after the LDS rows of the sums wave were taken out again - config 3 and the others on the final build, the GPU suite
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== config 3 (digest b229b2dfc30a)"; timeout 300 python tools/sweep.py 3 0 0 '{"base":{},"base2":{}}' || echo "RC $?"
  echo "== config 3, old eight-wave kernel"; bash tools/ab_lib.sh "tools/_bin/libnyx_w8n_old.so -" 3 5000 720
  echo "== config 4 (digest e203310ebb09)"; timeout 300 python tools/sweep.py 4 0 0 '{"base":{},"base2":{}}' || echo "RC $?"
  echo "== 1250 x 24 h (digest 1152333ec1b0)"; timeout 200 python tools/sweep.py 2 1250 24 '{"base":{},"base2":{}}' || echo "RC $?"
  echo "== 10000 x 24 h (digest 0299bb16009e)"; timeout 200 python tools/sweep.py 2 0 0 '{"base":{},"hs":{"harmonics_feed":1}}' 2 || echo "RC $?"
  echo "== config 5 6 h (48dd2474d8d4)"; timeout 300 python tools/sweep.py 5 0 6 '{"base":{}}' || echo "RC $?"
} > gpurun_out/r6_call17.log 2>&1
grep -v hwave gpurun_out/r6_call17.log
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6_call17_tests.log 2>&1
tail -5 gpurun_out/r6_call17_tests.log
