#!/bin/bash
# round 6, GPU call 34: helpers with a second column on the OLDEST column wave of each SIMD (NYX_HIP_COOP_DEAL=4), scalar and streamed feed
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
E='"env":{"NYX_HIP_COOP_DEAL":4}'
V="{\"base\":{},\"hs\":{\"harmonics_feed\":1},\"d16\":{$E,\"coop_max_columns\":16,\"show_sched\":1},\"d16hs\":{$E,\"coop_max_columns\":16,\"harmonics_feed\":1},\"d18\":{$E,\"coop_max_columns\":18},\"d18hs\":{$E,\"coop_max_columns\":18,\"harmonics_feed\":1,\"show_sched\":1},\"d20hs\":{$E,\"coop_max_columns\":20,\"harmonics_feed\":1},\"d22hs\":{$E,\"coop_max_columns\":22,\"harmonics_feed\":1},\"base2\":{}}"
{
  echo "== 640 x 20 min nofan parity"; NYX_HIP_LIB=tools/_bin/libnyx_deal4.so timeout 100 python tools/sweep.py 2 640 0.34 "{\"d18hs\":{$E,\"coop_max_columns\":18,\"harmonics_feed\":1,\"debug_flags\":134217728},\"base\":{\"debug_flags\":134217728}}" 1 64 || echo "RC $?"
  echo "== 10000 x 3 h"; NYX_HIP_LIB=tools/_bin/libnyx_deal4.so timeout 600 python tools/sweep.py 2 0 3 "$V" 2 64 || echo "RC $?"
} > gpurun_out/r6_call34.log 2>&1
grep -v "hwave\|roles (kind" gpurun_out/r6_call34.log | tail -70
