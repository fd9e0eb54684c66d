#!/bin/bash
for lib in head D cur; do
  echo "== $lib"; NYX_HIP_LIB=tools/_bin/libnyx_$lib.so timeout 200 python tools/sweep.py 2 10000 3 '{"x":{}, "xp":{"profile":1}}' 1 2>&1 | grep "^x\|wave  [0-2]:\|wave 15"
done
