#!/bin/bash
for lib in cur nop1 nop2 nop3 base; do
  echo "== $lib"; NYX_HIP_LIB=tools/_bin/libnyx_$lib.so timeout 200 python tools/sweep.py 2 10000 3 '{"x":{}}' 2 2>&1 | grep "^x"
done
