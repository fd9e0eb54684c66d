#!/bin/bash
{
timeout 300 python tools/sweep.py 3 0 0 '{"new":{}, "new_prof":{"profile":1}}' 2 64
timeout 300 python tools/sweep.py 4 0 0 '{"new":{}, "new_prof":{"profile":1}}' 2
} 2>&1 | grep -v amdgpu
