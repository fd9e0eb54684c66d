#!/bin/bash
O=gpurun_out/r4_call4; mkdir -p $O
{
timeout 300 python tools/tune_schedule.py 2 10000 3 14
timeout 300 python tools/tune_schedule.py 5 6250 1 12
timeout 300 python tools/tune_schedule.py 5 6250 1 10 '{"coop_helper_ratio":1.55,"coop_fraction":0.42}'
} > $O/log.txt 2>&1
cat $O/log.txt
