#!/bin/bash
timeout 800 python tools/tune_schedule.py 2 10000 24 10 2>&1 | grep -v amdgpu | grep "^iter\|^best\|^config"
timeout 300 python tools/sweep.py 2 10000 24 '{"c14":{}, "f40_c15":{"coop_fraction":0.40,"coop_max_columns":15}, "f34":{"coop_fraction":0.34}}' 1 2>&1 | grep -v amdgpu
