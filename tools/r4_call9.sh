#!/bin/bash
O=gpurun_out/r4_call9; mkdir -p $O
{
timeout 300 python tools/tune_schedule.py 2 16384 3 8
timeout 300 python tools/tune_schedule.py 2 10000 3 8 '{}' 1.000,1.503,0.961,1.887,1.733,1.409,1.384,1.273,0.966,0.769,0.752,0.766,0.568,0.382,0.323,0.318
timeout 300 python tools/tune_schedule.py 5 6250 1 6 '{}' 1.000,1.755,1.708,1.804,1.733,1.237,1.209,1.183,1.094,0.655,0.632,0.608,0.555,0.259,0.306,0.256
timeout 300 python tools/tune_schedule.py 5 6250 1 6 '{"coop_helper_ratio":1.3,"coop_fraction":0.40}' 1.000,1.755,1.708,1.804,1.733,1.237,1.209,1.183,1.094,0.655,0.632,0.608,0.555,0.259,0.306,0.256
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
