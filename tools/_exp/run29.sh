V='{"a":{},"b":{},"c":{}}'
for lib in "" tools/_bin/libnyx_noprof.so tools/_bin/libnyx_seg.so "" tools/_bin/libnyx_noprof.so; do
  echo "=== LIB $lib"
  NYX_HIP_LIB=$lib python tools/sweep.py 2 0 3 "$V" 1 2>&1 | grep -v "^   "
done
echo "=== full chip"
for lib in "" tools/_bin/libnyx_noprof.so; do
  echo "=== LIB $lib"
  NYX_HIP_LIB=$lib python tools/sweep.py 2 16384 3 '{"f":{},"g":{}}' 1 2>&1 | grep -v "^   "
  NYX_HIP_LIB=$lib python tools/sweep.py 3 0 0 '{"c3":{},"c3b":{}}' 1 2>&1 | grep -v "^   "
done
