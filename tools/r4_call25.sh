#!/bin/bash
for rep in 1 2; do for lib in head V1 V2 new; do
  l=tools/_bin/libnyx_$lib.so; [ $lib = new ] && l=""
  echo "== $lib"; NYX_HIP_LIB=$l timeout 200 python tools/sweep.py 2 10000 8 '{"x":{}}' 1 2>&1 | grep "^x"
done; done
