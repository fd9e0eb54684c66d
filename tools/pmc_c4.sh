#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_c4; mkdir -p $O
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS" "SQ_IFETCH SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_SALU"; do
  TAG=$(echo $C | tr ' ' '_')
  timeout 150 rocprofv3 --pmc $C -d $O/$TAG --output-format csv -- python tools/time_config.py 4 > $O/$TAG.log 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmc_c4/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "nyx_propagate_kernel_stmq" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(acc): print(k, acc[k], "over", cnt[k], "dispatches ->", acc[k] / cnt[k], "per dispatch")
PY
