#!/bin/bash
O=gpurun_out/r4_call31b; mkdir -p $O
{
timeout 150 python tools/sweep.py 2 0 3 '{"base":{},"fan":{"role_fanout":1},"fan_prof":{"role_fanout":1,"profile":1},"base_prof":{"profile":1}}' 1 64
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
