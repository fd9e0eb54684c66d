#!/bin/bash
# A/B of library builds on config 3 (interleaved): tools/_exp/ab3.sh "<name> ..." '<sweep json>'
for rep in 1 2; do
for v in $1; do
  l=$([ "$v" = "-" ] && echo "" || echo tools/_bin/libnyx_$v.so)
  echo "== $v"
  NYX_HIP_LIB=$l timeout 120 python tools/sweep.py 3 0 0 "$2" 1 16 2>&1 | grep -v "^config"
done
done
