#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r2w
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -6 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
