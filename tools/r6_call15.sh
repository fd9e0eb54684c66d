#!/bin/bash
# round 6, GPU call 15: config 4 cycle table on the current build; helpers on the streamed feed over the full day (three interleaved pairs)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== config 4 (digest e203310ebb09)"; timeout 300 python tools/sweep.py 4 0 0 '{"base":{},"prof":{"profile":1,"show_sched":1},"base2":{}}' || echo "RC $?"
  echo "== config 3 (digest b229b2dfc30a)"; timeout 300 python tools/sweep.py 3 0 0 '{"base":{},"prof":{"profile":1,"show_sched":1},"base2":{}}' || echo "RC $?"
  echo "== configs[1] 24 h: helpers' feed"; timeout 600 python tools/sweep.py 2 0 0 '{"base":{},"hs":{"harmonics_feed":1}}' 3 || echo "RC $?"
  echo "== config 5 6 h (48dd2474d8d4)"; timeout 300 python tools/sweep.py 5 0 6 '{"base":{}}' || echo "RC $?"
} > gpurun_out/r6_call15.log 2>&1
grep -v hwave gpurun_out/r6_call15.log
