#!/bin/bash
{
timeout 300 python tools/sweep.py 2 10000 24 '{"new":{}, "lazy":{"debug_flags":2097152}}' 3
timeout 300 python tools/sweep.py 5 6250 2 '{"new":{}, "lazy":{"debug_flags":2097152}}' 2
tools/_bin/cheby_mb
} 2>&1 | grep -v amdgpu
