#!/bin/bash
# round 6, GPU call 31: the driver's view - default bench run (wall time), smoke, the forced-collectives path
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{ time python bench.py > gpurun_out/r6_bench_final.json 2> gpurun_out/r6_bench_final.err; } 2> gpurun_out/r6_bench_final.time
tail -3 gpurun_out/r6_bench_final.time; tail -c 600 gpurun_out/r6_bench_final.json; echo
{ time python __graft_entry__.py smoke; } 2>&1 | tail -5
{ time python bench.py --force-collectives --steps 2 --warmup 1 --no-other-configs --no-cpu-baseline > gpurun_out/r6_bench_coll.json 2>gpurun_out/r6_bench_coll.err; } 2>&1 | tail -3
python -c "
import json; d=json.loads(open('gpurun_out/r6_bench_coll.json').read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','launch','rccl_ranks','all_gather_ms','backend')})"
