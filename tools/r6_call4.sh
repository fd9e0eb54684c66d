#!/bin/bash
# round 6, GPU call 4: out-of-line stage sums (integ_sums) - digests, fan-out shard sizes, headline
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
L=gpurun_out/r6_call4.log
{
  echo "== 640 x 1 h, parity"; timeout 120 python tools/sweep.py 2 640 1 '{"fan":{},"nofan":{"debug_flags":16777216}}' 1 64 || echo "RC $?"
  echo "== 10000 x 3 h (415720a47b92)"; timeout 200 python tools/sweep.py 2 0 3 '{"base":{},"prof":{"profile":1}}' 1 64 || echo "RC $?"
  echo "== 1250 x 3 h fan"; timeout 200 python tools/sweep.py 2 1250 3 '{"fan":{"profile":1},"fanp":{}}' 1 64 || echo "RC $?"
  for n in 1250 2500 5000 10000; do echo "== $n x 24 h"; timeout 300 python tools/sweep.py 2 $n 24 '{"run":{},"run2":{}}' || echo "RC $?"; done
  echo "== config 5 6 h (48dd2474d8d4)"; timeout 300 python tools/sweep.py 5 0 6 '{"base":{}}' || echo "RC $?"
  echo "== full chip (5038b80c38e5)"; timeout 200 python tools/sweep.py 2 16384 3 '{"base":{}}' || echo "RC $?"
} > $L 2>&1
grep -v "^    wave\|hwave" $L | tail -60
