echo "=== main: config 2 plain and profile"
python tools/sweep.py 2 0 3 '{"a":{},"p":{"profile":1},"b":{}}' 1 2>&1 | grep -v "^    \(wave\|hwave\)"
for lib in "" tools/_bin/libnyx_stmq_np.so "" tools/_bin/libnyx_stmq_np.so; do
  echo "=== c4 LIB $lib"
  NYX_HIP_LIB=$lib python tools/sweep.py 4 0 0 '{"a":{},"b":{},"c":{}}' 1 2>&1 | grep -v "^   "
done
for lib in "" tools/_bin/libnyx_w8n_np.so "" tools/_bin/libnyx_w8n_np.so; do
  echo "=== c3 LIB $lib"
  NYX_HIP_LIB=$lib python tools/sweep.py 3 0 0 '{"a":{},"b":{},"c":{}}' 1 2>&1 | grep -v "^   "
done
python -m pytest tests/test_gpu_tuning_paths.py -m gpu -x -q 2>&1 | tail -3
