#!/bin/bash
# kernel trace (start / end stamps) of C2 train steps -> per-step timeline: tools/trace_step.sh <tag>
TAG=${1:-trace}; ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $ROOT/tools/bench_train.py bf16 6 > $OUT/run.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob("/tmp/tr/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps: split at clip_adam_ema_kernel (last kernel of a step)
ends = [i for i, r in enumerate(rows) if "clip_adam_ema" in r["Kernel_Name"]]
assert len(ends) >= 4
lo, hi = ends[-3] + 1, ends[-2] + 1          # one full step in steady state
step = rows[lo:hi]
t0, t1 = int(step[0]["Start_Timestamp"]), int(step[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step)
print(f"one train step: {len(step)} launches, span {(t1 - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, idle {(t1 - t0 - busy) / 1e3:.1f} us")
agg = collections.OrderedDict()
prev_end = None
gaps = collections.Counter()
for r in step:
    n = r["Kernel_Name"].split("(")[0][:56]
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(n, [0, 0, 0])
    a[0] += 1; a[1] += d
    if prev_end is not None:
        a[2] += max(0, int(r["Start_Timestamp"]) - prev_end)
    prev_end = int(r["End_Timestamp"])
with open(out + "/step_timeline.txt", "w") as fo:
    for n, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        line = f"{n:58s} x{c:3d}  busy {d / 1e3:8.1f} us   idle before {g / 1e3:7.1f} us"
        print(line); fo.write(line + "\n")
PY
