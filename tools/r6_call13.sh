#!/bin/bash
# round 6, GPU call 13: the helpers' share and feed re-measured behind the out-of-line integrator (the answer is no longer on the owner's critical path)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== 10000 x 3 h"; timeout 600 python tools/sweep.py 2 0 3 '{"base":{},"hs":{"harmonics_feed":1},"f38":{"coop_fraction":0.38},"f38hs":{"coop_fraction":0.38,"harmonics_feed":1},"f40":{"coop_fraction":0.40},"f40hs":{"coop_fraction":0.40,"harmonics_feed":1},"f43hs":{"coop_fraction":0.43,"harmonics_feed":1},"f33":{"coop_fraction":0.33},"base2":{}}' 2 || echo "RC $?"
} > gpurun_out/r6_call13.log 2>&1
grep -v hwave gpurun_out/r6_call13.log
