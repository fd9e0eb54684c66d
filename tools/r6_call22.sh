#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 600 python -m pytest tests/test_gpu_fan.py -x -q -m gpu 2>&1 | tail -30
