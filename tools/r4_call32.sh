#!/bin/bash
timeout 800 python tools/tune_schedule.py 2 16384 24 12 2>&1 | grep -v amdgpu | grep "^iter\|^best\|^config"
timeout 800 python tools/tune_schedule.py 5 6250 6 8 2>&1 | grep -v amdgpu | grep "^iter\|^best\|^config"
