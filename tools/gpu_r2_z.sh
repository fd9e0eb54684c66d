#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2z
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 600 -k "cooperative or pipelined" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
timeout 300 python bench.py --config 5 --steps 1 --warmup 1 --no-cpu-baseline --no-dense-output --no-host-call > $O/b5.json 2>$O/b5.err; python -c "
import json; d=json.loads(open('$O/b5.json').read().strip().splitlines()[-1]); print('config5', d['value'], d['kernel_ms'], d['roofline']['frac'], d['occupancy']['helper_workgroups'])"
