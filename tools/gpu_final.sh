#!/bin/bash
# end-of-round check on the GPU box: the -m gpu suite, smoke(), and the default bench line
set -u
export TMPDIR=/tmp
O=gpurun_out/final
mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q --timeout 300 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json
