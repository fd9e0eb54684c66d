#!/bin/bash
export TMPDIR=/tmp
cd /root/repo
for lib in head cur; do
  rm -rf gpurun_out/ic_$lib
  NYX_HIP_LIB=tools/_bin/libnyx_$lib.so timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d gpurun_out/ic_$lib --output-format csv -- python tools/sweep.py 2 10000 3 '{"x":{}}' 1 > gpurun_out/ic_$lib.log 2>&1
  python - "$lib" <<'PY'
import csv,glob,collections,sys
lib=sys.argv[1]
acc=collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/ic_{lib}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nyx_propagate" in r.get("Kernel_Name",""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(lib, {k:(max(v), len(v)) for k,v in acc.items()})
PY
done
