#!/bin/bash
# PMC passes over wae_glu_layer_fwd (C2 shape).  Usage: tools/pmc_glu.sh <nw> <out_dir>   (run on the GPU box)
NW=${1:-4}
OUT=${2:-gpurun_out/pmc_glu_nw$NW}
ROOT=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" \
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS" \
  "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS" \
  "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_TA_TCP_STATE_READ" \
  "TCP_TOTAL_ACCESSES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_READ" \
  "TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL" \
  "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $ROOT/$OUT/p$i -- python3 $ROOT/tools/run_glu.py $NW 0 0 6 > $ROOT/$OUT/p$i.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/p*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "glu_fwd" not in r["Kernel_Name"]:
            continue
        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open("$OUT/summary.txt", "w") as fh:
    for k, v in agg.items():
        line = f"{k:34s} mean/launch {sum(v)/len(v):16.1f}  (n={len(v)})"
        print(line); fh.write(line + "\n")
PY
