#!/bin/bash
# kernel trace of C2 train steps -> start/end of every gemm_tm launch of one backward sweep (do the two halves overlap?):
# tools/sweep_trace.sh <tag>   (environment, e.g. WAE_BWD_HALVES, is inherited)
TAG=${1:-sweep}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $ROOT/tools/bench_train.py bf16 6 > $OUT/run.log 2>&1
cd $ROOT
tail -3 $OUT/run.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob("/tmp/tr/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "clip_adam_ema" in r["Kernel_Name"]]
lo, hi = ends[-3] + 1, ends[-2] + 1
step = rows[lo:hi]
t0 = int(step[0]["Start_Timestamp"])
tm = [r for r in step if "gemm_tm_kernel" in r["Kernel_Name"]]
span0, span1 = int(tm[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in tm)
with open(out + "/sweep_timeline.txt", "w") as fo:
    def P(s):
        print(s); fo.write(s + "\n")
    P(f"step span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us; gemm_tm launches {len(tm)}; first..last gemm_tm {(span1 - span0) / 1e3:.1f} us")
    for r in step:
        n = r["Kernel_Name"]
        if "gemm_tm_kernel" not in n and "gemm_tn" not in n and "head_bwd" not in n:
            continue
        short = n.split("(")[0][:60]
        P(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f}  dur {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  q{r.get('Queue_Id', '?')} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?'))} {short}")
PY
