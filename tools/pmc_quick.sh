#!/bin/bash
# quick per-kernel HBM traffic of one bench step: tools/pmc_quick.sh <tag>
TAG=$1; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-ar > $OUT/pmc_$c.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.OrderedDict()
    for f in glob.glob("$OUT/pmc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: agg.setdefault(r["Kernel_Name"].split("(")[0][:60], []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if sum(v) / len(v) > 20000: print(c, k, len(v), "mean MB %.1f" % (sum(v) / len(v) / 1024))
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
