#!/bin/bash
{
NYX_HIP_LIB=tools/_bin/libnyx_head.so timeout 300 python tools/sweep.py 2 10000 24 '{"head_prof":{"profile":1}}' 1
timeout 300 python tools/sweep.py 2 10000 24 '{"new_prof":{"profile":1}}' 1
NYX_HIP_LIB=tools/_bin/libnyx_head.so timeout 300 python tools/sweep.py 2 10000 6 '{"head6_prof":{"profile":1}}' 1
timeout 300 python tools/sweep.py 2 10000 6 '{"new6_prof":{"profile":1}}' 1
} 2>&1 | grep -v amdgpu | grep "prof\|wave  [0-3]:\|wave 1[25]:"
