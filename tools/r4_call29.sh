#!/bin/bash
timeout 600 python tools/tune_schedule.py 2 10000 24 10 2>&1 | grep -v amdgpu | grep "^iter\|^best\|^config"
