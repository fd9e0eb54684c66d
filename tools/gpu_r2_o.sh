#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2o
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 -k "pipelined or golden or cooperative" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log
for sp in 1 0; do
NYX_HIP_SPEC=$sp timeout 300 python bench.py --config 2 --steps 2 --warmup 1 --no-cpu-baseline --no-dense-output --no-host-call > $O/bench_c2_$sp.json 2>$O/bench_c2_$sp.err; python -c "
import json; d=json.loads(open('$O/bench_c2_$sp.json').read().strip().splitlines()[-1]); print('config2 spec=$sp', d['value'], d['kernel_ms'], d['roofline']['frac'])"
done
