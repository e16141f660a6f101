#!/bin/bash
# HBM-side fetch of the static weight-gradient launch with and without team pacing: tools/pmc_tq_pace.sh <tag> [windows...]
TAG=$1; shift; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for w in "${@:-0 4}"; do
  export WAE_TQ_PACE=$w
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_w$w -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-ar > $OUT/pmc_w$w.log 2>&1
done
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
for d in sorted(glob.glob(out + "/pmc_w*/")):
    vals = []
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "FETCH_SIZE" and "gemm_tn_static" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    print(os.path.basename(d.rstrip("/")), "gemm_tn_static launches", len(vals), "FETCH_SIZE mean %.1f MB (x2 on gfx950: %.1f MB)" % (sum(vals) / max(len(vals), 1) / 1024, 2 * sum(vals) / max(len(vals), 1) / 1024))
PY
rm -rf $OUT/pmc_w*/
