#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite, one bench line per configuration, cycle accounting of configs 3/4/5
set -u
export TMPDIR=/tmp
O=gpurun_out/r2a
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for c in 2 3 4 5; do
  timeout 400 python bench.py --config $c --steps 2 --warmup 1 > $O/bench_c$c.json 2> $O/bench_c$c.err; echo "bench $c rc $?"
  tail -c 400 $O/bench_c$c.json
done
for c in 3 4 5; do
  NYX_HIP_PROFILE=1 timeout 200 python tools/time_config.py $c > $O/cycles_c$c.txt 2>&1
  cat $O/cycles_c$c.txt
done
