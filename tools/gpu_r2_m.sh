#!/bin/bash
# 48-byte table entries: bench first (the number that decides), then the parity tests
set -u
export TMPDIR=/tmp
O=gpurun_out/r2m
mkdir -p $O
for c in 2 5 3 4; do
timeout 300 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-dense-output --no-host-call > $O/bench_c$c.json 2>$O/bench_c$c.err; python -c "
import json; d=json.loads(open('$O/bench_c$c.json').read().strip().splitlines()[-1]); print('config$c', d['value'], d['kernel_ms'], d['roofline']['frac'])"
done
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
