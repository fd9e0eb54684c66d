#!/bin/bash
# interleaved A/B of library builds on configs[1] through tools/sweep.py (digest + ms): tools/_exp/ab2.sh "<lib or -> ..." <hours> '<json>'
for rep in 1 2 3; do
for v in $1; do
  l=$([ "$v" = "-" ] && echo "" || echo $v)
  echo "== $v"
  NYX_HIP_LIB=$l timeout 300 python tools/sweep.py 2 0 $2 "$3" 1 2>&1 | grep -v "^config"
done
done
