#!/bin/bash
O=gpurun_out/r4_call7; mkdir -p $O
{
echo "=== tagged granules"
timeout 300 python tools/sweep.py 2 10000 3 "$(cat tools/_v7.json)" 1 64
echo "=== previous build"
NYX_HIP_LIB=tools/_bin/libnyx_base.so timeout 300 python tools/sweep.py 2 10000 3 '{"base":{}}' 2
echo "=== config 5"
timeout 300 python tools/sweep.py 5 6250 1 '{"base":{}, "r155_f45":{"coop_helper_ratio":1.55,"coop_fraction":0.45}, "r155":{"coop_helper_ratio":1.55}, "base_prof":{"profile":1}}' 1 64
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
