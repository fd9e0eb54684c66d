#!/bin/bash
# Collects what profiles/ holds for a round on the GPU box: bench line, rocprofv3 kernel trace of the same command, and
# the PMC passes (each counter set in its own run, never combined with tracing).  Run from the repo root under gpurun.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/prof_round
mkdir -p $OUT
if [ -z "${SKIP_BENCH:-}" ]; then timeout 400 python bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err; fi
tail -c 600 $OUT/bench.json
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format rocpd -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-dense-output > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" $OUT/kernel_trace_stats.md > /dev/null 2>&1 || echo "summary failed"
head -4 $OUT/kernel_trace_stats.md
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU"; do
  TAG=$(echo $C | tr ' ' '_')
  timeout 150 rocprofv3 --pmc $C -d $OUT/pmc_$TAG --output-format csv -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-dense-output > $OUT/pmc_$TAG.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float)
for f in glob.glob("gpurun_out/prof_round/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nyx_propagate" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
open("gpurun_out/prof_round/pmc_summary.txt", "w").write("\n".join(f"{k} {v}" for k, v in sorted(acc.items())))
print(dict(acc))
PY
