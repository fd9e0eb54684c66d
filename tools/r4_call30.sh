#!/bin/bash
timeout 800 python tools/tune_schedule.py 2 10000 24 16 '{}' 1.000,1.420,0.607,1.965,1.893,1.494,1.488,1.402,1.057,0.817,0.879,0.621,0.445,0.292,0.424,0.186 2>&1 | grep -v amdgpu | grep "^iter\|^best\|^config"
