#!/bin/bash
for lib in E D p26880 p28k p32k p36k; do
  echo "== $lib"; NYX_HIP_LIB=tools/_bin/libnyx_$lib.so timeout 200 python tools/sweep.py 2 10000 3 '{"x":{}}' 2 2>&1 | grep "^x"
done
echo "== config 5 two parts"
for lib in D p28k p32k p36k; do
  echo "== $lib"; NYX_HIP_LIB=tools/_bin/libnyx_$lib.so timeout 200 python tools/sweep.py 5 6250 1 '{"f70":{"coop_fraction":0.70}}' 2 2>&1 | grep "^f70"
done
