#!/bin/bash
# round 6, GPU call 20: the in-kernel cycle accounting of the final build (profiles/round06_cycle_accounting.md)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
{
  timeout 400 python tools/sweep.py 2 0 0 '{"product":{},"accounting":{"profile":1,"show_sched":1}}' || echo "RC $?"
  echo; timeout 300 python tools/sweep.py 2 1250 24 '{"product":{},"accounting":{"profile":1,"show_sched":1}}' || echo "RC $?"
  echo; timeout 300 python tools/sweep.py 3 0 0 '{"product":{},"accounting":{"profile":1,"show_sched":1}}' || echo "RC $?"
  echo; timeout 300 python tools/sweep.py 4 0 0 '{"product":{},"accounting":{"profile":1,"show_sched":1}}' || echo "RC $?"
  echo; timeout 400 python tools/sweep.py 5 0 6 '{"product":{},"accounting":{"profile":1,"show_sched":1}}' || echo "RC $?"
  echo; timeout 300 python tools/sweep.py 2 16384 3 '{"product":{},"accounting":{"profile":1,"show_sched":1}}' || echo "RC $?"
} > gpurun_out/r6_cycle_accounting.log 2>&1
grep -c . gpurun_out/r6_cycle_accounting.log; grep "product\|accounting\|RC" gpurun_out/r6_cycle_accounting.log
