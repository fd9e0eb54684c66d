#!/bin/bash
O=gpurun_out/r4_call10; mkdir -p $O
{
timeout 300 python tools/sweep.py 2 10000 3 '{"new":{}, "old_sched":{"debug_flags":32768,"harmonics_feed":0}}' 3 64
timeout 300 python tools/sweep.py 2 16384 3 '{"new":{}, "old_sched":{"debug_flags":32768,"harmonics_feed":0}}' 2
timeout 300 python tools/sweep.py 5 6250 1 '{"new":{}, "old_sched":{"debug_flags":32768}}' 2 64
timeout 300 python tools/sweep.py 2 3000 1 '{"new":{}, "old_sched":{"debug_flags":32768,"harmonics_feed":0}, "det":{"deterministic":1}}' 2 64
timeout 300 python tools/sweep.py 2 40000 1 '{"new":{}, "old_sched":{"debug_flags":32768,"harmonics_feed":0}}' 1
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
