#!/bin/bash
{
timeout 300 python tools/sweep.py 2 10000 24 '{"new":{}}' 3 64
timeout 300 python tools/sweep.py 2 10000 3 '{"new":{}}' 2
NYX_HIP_LIB=tools/_bin/libnyx_head.so timeout 300 python tools/sweep.py 2 10000 24 '{"head":{}}' 2
timeout 300 python tools/sweep.py 2 5000 6 '{"new":{}, "old_sched":{"debug_flags":32768,"harmonics_feed":0}}' 2
timeout 300 python tools/sweep.py 2 2000 6 '{"new":{}, "old_sched":{"debug_flags":32768,"harmonics_feed":0}}' 2
} 2>&1 | grep -v amdgpu
