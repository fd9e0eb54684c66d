#!/bin/bash
O=gpurun_out/r4_call5; mkdir -p $O
{
timeout 300 python tools/sweep.py 2 10000 3 '{"base":{}, "blk":{"debug_flags":32768}, "blk_f1":{"debug_flags":32768,"harmonics_feed":1}, "blk_f2":{"debug_flags":32768,"harmonics_feed":2}, "f2":{"harmonics_feed":2}, "blk_f2_prof":{"debug_flags":32768,"harmonics_feed":2,"profile":1}, "blk_f1_prof":{"debug_flags":32768,"harmonics_feed":1,"profile":1}}' 2 64
timeout 300 python tools/sweep.py 5 6250 1 '{"base":{}, "blk":{"debug_flags":32768}, "blk_prof":{"debug_flags":32768,"profile":1}, "blk_f2":{"debug_flags":32768,"harmonics_feed":2}}' 2 64
timeout 300 python tools/sweep.py 2 16384 3 '{"base":{}, "blk":{"debug_flags":32768}, "blk_f1":{"debug_flags":32768,"harmonics_feed":1}, "f1":{"harmonics_feed":1}}' 1
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
