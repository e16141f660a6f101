#!/bin/bash
ROOT=$(pwd); OUT=$ROOT/gpurun_out/c3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/c3_probe.py > $OUT/line.json 2> $OUT/stats.err
for f in $OUT/stats/*/*kernel_stats.csv; do cp $f $OUT/kernel_stats.csv; done
rm -rf $OUT/stats
