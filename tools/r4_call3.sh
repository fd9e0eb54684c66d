#!/bin/bash
O=gpurun_out/r4_call3; mkdir -p $O
{
timeout 300 python tools/sweep.py 2 10000 3 '{"new":{}, "new_prof":{"profile":1}, "no_incr":{"debug_flags":16384}, "no_incr_seg":{"debug_flags":20480}, "no_incr_pad":{"debug_flags":24576}, "all_off":{"debug_flags":28672}, "all_off_prof":{"debug_flags":28672,"profile":1}, "new_feed1":{"harmonics_feed":1}, "new_cal":{"schedule":1}}' 2 64
timeout 200 python tools/sweep.py 3 5000 240 '{"new":{}, "no_incr_off":{"debug_flags":28672}}' 1 64
timeout 200 python tools/sweep.py 4 0 0 '{"new":{}, "new_prof":{"profile":1}, "all_off":{"debug_flags":28672}}' 2
timeout 200 python tools/sweep.py 5 6250 1 '{"new":{}, "all_off":{"debug_flags":28672}}' 1 64
} > $O/log.txt 2>&1
cat $O/log.txt
