#!/bin/bash
O=gpurun_out/r4_call8; mkdir -p $O
{
timeout 300 python tools/tune_schedule.py 2 10000 3 10 '{"debug_flags":32768,"harmonics_feed":2}'
timeout 300 python tools/tune_schedule.py 2 10000 3 6 '{"harmonics_feed":2}'
timeout 300 python tools/tune_schedule.py 5 6250 1 8 '{"debug_flags":32768}'
} > $O/log.txt 2>&1
grep -v amdgpu.ids $O/log.txt
