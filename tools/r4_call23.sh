#!/bin/bash
{
timeout 600 python tools/sweep.py 2 10000 24 '{"new":{}, "no_dcm_incr":{"debug_flags":16384}, "old_sched":{"debug_flags":32768,"harmonics_feed":0}, "old_sched_nodcm":{"debug_flags":49152,"harmonics_feed":0}, "round8":{"debug_flags":131072}, "new_prof":{"profile":1}}' 1
NYX_HIP_LIB=tools/_bin/libnyx_head.so timeout 300 python tools/sweep.py 2 10000 24 '{"head":{}}' 1
} 2>&1 | grep -v amdgpu
