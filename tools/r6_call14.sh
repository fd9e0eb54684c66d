#!/bin/bash
# round 6, GPU call 14: the start of a column walk kept across the stages (harm_stream_setup / harmonics_stream_d) against the per-stage lookup
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  echo "== 640 x 1 h nofan, parity on 64 (digest e51455ac59db)"
  for v in hsd0 hsd1; do echo "-- $v"; NYX_HIP_LIB=tools/_bin/libnyx_$v.so timeout 120 python tools/sweep.py 2 640 1 '{"nofan":{"debug_flags":134217728}}' 1 64 || echo "RC $?"; done
  echo "== 10000 x 3 h (digest 415720a47b92)"
  for v in hsd0 hsd1; do echo "-- $v"; NYX_HIP_LIB=tools/_bin/libnyx_$v.so timeout 200 python tools/sweep.py 2 0 3 '{"base":{},"hs":{"harmonics_feed":1},"base2":{},"hs2":{"harmonics_feed":1}}' || echo "RC $?"; done
  echo "== configs[1] 10000 x 24 h"; bash tools/ab_lib.sh "tools/_bin/libnyx_hsd0.so tools/_bin/libnyx_hsd1.so" 2 10000 24
  echo "== full chip 16384 x 3 h"; bash tools/ab_lib.sh "tools/_bin/libnyx_hsd0.so tools/_bin/libnyx_hsd1.so" 2 16384 3
  echo "== 24 h digest (0299bb16009e)"; NYX_HIP_LIB=tools/_bin/libnyx_hsd1.so timeout 200 python tools/sweep.py 2 0 0 '{"base":{},"hs":{"harmonics_feed":1}}' || echo "RC $?"
  echo "== full chip digest (5038b80c38e5)"; NYX_HIP_LIB=tools/_bin/libnyx_hsd1.so timeout 200 python tools/sweep.py 2 16384 3 '{"base":{}}' || echo "RC $?"
} > gpurun_out/r6_call14.log 2>&1
grep -v hwave gpurun_out/r6_call14.log
