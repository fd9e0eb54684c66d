#!/bin/bash
O=gpurun_out/r4_call6; mkdir -p $O
{
timeout 300 python tools/sweep.py 2 10000 3 "$(cat tools/_v6.json)" 1
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
