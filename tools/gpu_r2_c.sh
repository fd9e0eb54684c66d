#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2c
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
for c in 3 4; do NYX_HIP_PROFILE=1 timeout 200 python tools/time_config.py $c > $O/cycles_c$c.txt 2>&1; cat $O/cycles_c$c.txt; done
NYX_HIP_FANOUT=0 timeout 200 python tools/time_config.py 3 2>&1 | grep config
timeout 300 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --no-dense-output --no-host-call > $O/bench_c2.json 2>$O/bench_c2.err; python -c "
import json; d=json.loads(open('$O/bench_c2.json').read().strip().splitlines()[-1]); print('config2', d['value'], d['kernel_ms'], d['roofline']['frac'])"
