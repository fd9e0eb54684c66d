#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
timeout 300 python tools/sweep.py 4 0 0 '{"product":{},"accounting":{"profile":1,"show_sched":1}}' > gpurun_out/r6_cycle_accounting_cfg4.log 2>&1
cat gpurun_out/r6_cycle_accounting_cfg4.log
