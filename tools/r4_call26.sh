#!/bin/bash
for rep in 1 2; do
  echo "== head"; NYX_HIP_LIB=tools/_bin/libnyx_head.so timeout 200 python tools/sweep.py 2 10000 8 '{"x":{}}' 1 2>&1 | grep "^x"
  echo "== V2"; NYX_HIP_LIB=tools/_bin/libnyx_V2.so timeout 200 python tools/sweep.py 2 10000 8 '{"pack":{}, "round8":{"debug_flags":131072}}' 1 2>&1 | grep "^pack\|^round8"
done
