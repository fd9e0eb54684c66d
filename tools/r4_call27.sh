#!/bin/bash
for rep in 1 2; do
  echo "== head"; NYX_HIP_LIB=tools/_bin/libnyx_head.so timeout 200 python tools/sweep.py 2 10000 8 '{"x":{}}' 1 2>&1 | grep "^x"
  echo "== new"; timeout 200 python tools/sweep.py 2 10000 8 '{"x":{}}' 1 2>&1 | grep "^x"
done
timeout 200 python tools/sweep.py 5 6250 1 '{"auto":{}}' 2 64 2>&1 | grep "^auto\|parity"
