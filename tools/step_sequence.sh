#!/bin/bash
# every launch of ONE steady-state C2 train step in issue order with its duration and the idle gap in front of it:
# tools/step_sequence.sh <tag> [bench_train.py args]  -> gpurun_out/<tag>/step_sequence.txt
TAG=${1:-seq}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/trs
# (STEP_CONFIG=c3|c5: the same for another bench configuration, through bench.py)
if [ -n "$STEP_CONFIG" ]; then
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -- python3 $ROOT/bench.py --config $STEP_CONFIG --steps 6 --warmup 2 --no-sub --no-ar --no-fp32 --no-cpu > $OUT/run.log 2>&1
else
  rocprofv3 --kernel-trace --output-format csv -d /tmp/trs -- python3 $ROOT/tools/bench_train.py ${2:-bf16} 6 > $OUT/run.log 2>&1
fi
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob("/tmp/trs/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "clip_adam_ema" in r["Kernel_Name"]]
lo, hi = ends[-3] + 1, ends[-2] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
prev = None
big = ("glu_fwd_static", "glu_bwd_pair", "gemm_tn_static")
with open(out + "/step_sequence.txt", "w") as fo:
    tail = 0.0
    for r in step:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].split("(")[0][:70]
        gap = 0 if prev is None else s - prev
        prev = e
        if not any(b in n for b in big):
            tail += (e - s + max(gap, 0)) / 1e3
        fo.write(f"{(s - t0) / 1e3:9.1f} us  +{gap / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {n}\n")
    fo.write(f"span {(prev - t0) / 1e3:.1f} us; launches {len(step)}; outside the three big kernel families (busy + gaps in front): {tail:.1f} us\n")
print(open(out + "/step_sequence.txt").read()[-400:])
PY
