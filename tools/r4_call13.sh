#!/bin/bash
{
timeout 300 python tools/sweep.py 2 10000 3 '{"new":{}, "new_prof":{"profile":1}}' 2
NYX_HIP_LIB=tools/_bin/libnyx_base.so timeout 300 python tools/sweep.py 2 10000 3 '{"base_lib":{}}' 2
timeout 300 python tools/sweep.py 5 6250 1 '{"f60":{"coop_fraction":0.60}, "f65":{"coop_fraction":0.65}, "f70":{"coop_fraction":0.70}, "f75":{"coop_fraction":0.75}, "f65_prof":{"coop_fraction":0.65,"profile":1}}' 1
} 2>&1 | grep -v amdgpu.ids
