#!/bin/bash
# round 4, GPU call 1: where does config 2's evaluation period go?  (run from the repo root under gpurun)
export NYX_HIP_TUNING_ENV=1
O=gpurun_out/r4_diag1; mkdir -p $O
run() { echo "### $*"; env "$@" 2>&1 | grep -v '^$' ; }
{
run timeout 200 python tools/time_config.py 2 10000 3
run timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_PROFILE=1 timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_CALIBRATE=1 timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_CALIBRATE=1 NYX_HIP_DEBUG=0x100 timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_CALIBRATE=1 NYX_HIP_DEBUG=0x100 NYX_HIP_PROFILE=1 timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_DEBUG=0x100 timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_CALIBRATE=1 NYX_HIP_DEBUG=0x200 timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_CALIBRATE=1 NYX_HIP_DEBUG=0x300 timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_COOP=0 timeout 200 python tools/time_config.py 2 10000 3
run NYX_HIP_COOP=0 NYX_HIP_DEBUG=0x100 NYX_HIP_CALIBRATE=1 timeout 200 python tools/time_config.py 2 10000 3
run timeout 200 python tools/time_config.py 2 16384 3
run NYX_HIP_DEBUG=0x100 NYX_HIP_CALIBRATE=1 timeout 200 python tools/time_config.py 2 16384 3
run timeout 300 python tools/time_config.py 5 6250 1
run NYX_HIP_PROFILE=1 timeout 300 python tools/time_config.py 5 6250 1
run timeout 300 python tools/time_config.py 3
run timeout 300 python tools/time_config.py 4
} > $O/log.txt 2>&1
cat $O/log.txt
