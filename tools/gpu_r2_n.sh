#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2n
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
timeout 300 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --no-dense-output --no-host-call > $O/bench_c2.json 2>$O/bench_c2.err; python -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('config2', d['value'], d['kernel_ms'], d['roofline']['frac'])"
