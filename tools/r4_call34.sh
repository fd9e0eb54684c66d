#!/bin/bash
{
timeout 400 python tools/sweep.py 2 10000 24 '{"new":{}, "scan1":{"debug_flags":4194304}, "scan2":{"debug_flags":8388608}, "scan4":{"debug_flags":16777216}, "spec_fetch":{"debug_flags":2097152}}' 2
timeout 300 python tools/sweep.py 5 6250 2 '{"new":{}, "scan2":{"debug_flags":8388608}}' 2
} 2>&1 | grep -v amdgpu
