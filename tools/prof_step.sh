#!/bin/bash
# kernel-stats of 13 bench steps -> gpurun_out/ks.csv  (tools/prof_step.sh [extra bench args])
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-ar "$@" > /tmp/st.log 2>&1
cp /tmp/st/*/*kernel_stats.csv $ROOT/gpurun_out/ks.csv
