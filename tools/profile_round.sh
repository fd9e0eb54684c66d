#!/bin/bash
# Collects what profiles/ holds for one BASELINE configuration on the GPU box: the bench line, the rocprofv3 kernel trace of
# the same command, and the PMC passes (each counter set in its own run, never combined with tracing).
#   CFG=2|3|4|5 [STEPS=3] bash tools/profile_round.sh        (run from the repo root under gpurun)
set -u
export TMPDIR=/tmp
CFG=${CFG:-2}
STEPS=${STEPS:-3}
# TAG / EXTRA: another workload of the same configuration (round 5: TAG=fullchip EXTRA="--n 16384 --hours 3" = configs[1]'s force model on a full chip)
TAG=${TAG:-cfg$CFG}
EXTRA="${EXTRA:-} --no-other-configs"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
if [ -z "${SKIP_BENCH:-}" ]; then timeout 900 python bench.py --config $CFG --steps $STEPS --warmup 1 $EXTRA > $OUT/bench.json 2> $OUT/bench.err; fi
tail -c 300 $OUT/bench.json; echo
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt --output-format rocpd -- python bench.py --config $CFG --steps $STEPS --warmup 1 --no-cpu-baseline --no-dense-output --no-host-call $EXTRA > $OUT/kt.log 2>&1
DB=$(find $OUT/kt -name "*.db" | head -1)
python tools/rocpd_summary.py "$DB" $OUT/kernel_trace_stats.md > /dev/null 2>&1 || echo "summary failed"
head -6 $OUT/kernel_trace_stats.md
[ -n "${SKIP_PMC:-}" ] && exit 0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_SALU" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
  PTAG=$(echo $C | tr ' ' '_')
  timeout 400 rocprofv3 --pmc $C -d $OUT/pmc_$PTAG --output-format csv -- python bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-dense-output --no-host-call $EXTRA > $OUT/pmc_$PTAG.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "nyx_propagate" in k:
            acc[(k.split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k.split("(")[0], r["Counter_Name"])] += 1
lines = [f"{k[0]} {k[1]} total {v:.6g} dispatches {cnt[k]} per_dispatch {v / cnt[k]:.6g}" for k, v in sorted(acc.items())]
open(out + "/pmc_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
