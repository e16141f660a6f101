#!/bin/bash
# Counter passes over one python script (run on the GPU box).  Usage: tools/pmc_run.sh <out_dir> <kernel substring[,substring...]> <script> [args...]
# Each pass is its own rocprofv3 run under `timeout` (a TA_* pass once hung a box until gpurun's limit: never list those).
OUT=$1; KSUB=$2; shift 2
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
  "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" \
  "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$OUT/p$i -- python3 $ROOT/"$@" > $ROOT/$OUT/p$i.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
subs = "$KSUB".split(",")
agg = collections.OrderedDict((s_, collections.OrderedDict()) for s_ in subs)
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        for s_ in subs:
            if s_ in r["Kernel_Name"]:
                agg[s_].setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for s_ in subs:
        fh.write("== kernels matching %r\n" % s_); print("==", s_)
        for k, v in agg[s_].items():
            line = f"{k:34s} mean/launch {sum(v)/len(v):18.1f}  (n={len(v)})"
            print(line); fh.write(line + "\n")
PY
