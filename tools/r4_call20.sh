#!/bin/bash
{
timeout 200 python tools/sweep.py 2 10000 3 '{"new":{}, "old_round":{"debug_flags":131072}}' 2 64
NYX_HIP_LIB=tools/_bin/libnyx_head.so timeout 200 python tools/sweep.py 2 10000 3 '{"head":{}}' 2
timeout 200 python tools/sweep.py 5 6250 1 '{"auto":{}, "f65":{"coop_fraction":0.65}, "f70":{"coop_fraction":0.70}, "one_part":{"debug_flags":524288}}' 2 64
} 2>&1 | grep -v amdgpu
