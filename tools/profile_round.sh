#!/bin/bash
# Round profile on the GPU box: tools/profile_round.sh <tag>  -> gpurun_out/<tag>/ (bench lines, kernel stats, PMC FETCH/WRITE passes)
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
# (PROFILE_ONLY=1: only the rocprofv3 passes -- the bench lines of a tag already exist)
if [ -z "$PROFILE_ONLY" ]; then
python3 bench.py > $OUT/bench_train.json 2> $OUT/bench_train.err
python3 bench.py --mode forward --no-cpu --no-ar > $OUT/bench_forward.json 2> $OUT/bench_forward.err
python3 bench.py --dtype fp16 --no-cpu --no-ar --no-fp32 > $OUT/bench_fp16.json 2> $OUT/bench_fp16.err
# BASELINE configs 3 and 5 (the two 8-GPU configurations), the per-GPU shard of each: the full line (roofline family, roofline_step, cpu_baseline)
python3 bench.py --config c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 bench.py --config c5 --steps 10 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
fi
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-ar --no-fp32 --no-sub > $OUT/profiled_line.json 2> $OUT/stats.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fwd -- python3 $ROOT/bench.py --mode forward --steps 10 --warmup 3 --no-cpu --no-ar > $OUT/profiled_forward_line.json 2> $OUT/stats_fwd.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-ar --no-fp32 --no-sub > $OUT/pmc_$c.log 2>&1
done
cd $ROOT
python3 - <<PY
import csv, glob, collections, json, sys
sys.path.insert(0, "$ROOT")
import bench
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.OrderedDict()
    for f in glob.glob("$OUT/pmc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            key = r["Kernel_Name"].split("(")[0]
            if "glu_fwd_static" in key:      # rocprofv3 garbles these symbols' template arguments: the entry point's NAME tells training
                key = key.split("<")[0].strip()   # (glu_fwd_static_z_kernel) from inference launches; mean over a step's 24 launches
            agg.setdefault(key, []).append(float(r["Counter_Value"]))
    with open("$OUT/pmc_%s_per_kernel_mean_kb.csv" % c, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "mean_kb"])
        for k, v in agg.items():
            w.writerow([k, len(v), "%.1f" % (sum(v) / len(v))])
            res.setdefault(k, {})[c] = sum(v) / len(v)
out = {k: {"fetch_kb_raw": v.get("FETCH_SIZE", 0.0), "write_kb": v.get("WRITE_SIZE", 0.0),
           "hbm_bytes_per_launch": (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024} for k, v in res.items()}
json.dump({"csrc_hash": bench.csrc_hash(), "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-ar --no-fp32 --no-sub (two separate passes)",
           "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 1/2 of the bytes of 16-B/lane streaming reads on gfx950 -> doubled; WRITE_SIZE exact; unit KB -> x1024",
           "kernels": out}, open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
# per-layer (per-dilation) durations of the fused layer kernel, inference and training launches, from the two kernel traces
python3 - <<PY
import csv, glob, collections, sys
sys.path.insert(0, "$ROOT")
import bench
from wavenet_autoencoders_amd import Geometry
dil = Geometry.from_cfg(bench.C2).dilations
rows = []
for tag, d in (("inference", "$OUT/stats_fwd"), ("training", "$OUT/stats")):
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))
                     if "glu_fwd" in r["Kernel_Name"]), key=lambda e: e[0])
        want = "glu_fwd_static_z_kernel" if tag == "training" else "glu_fwd_static_kernel"
        sel = [e for e in ev if (want + "<") in e[2] or (want + "(") in e[2] or "glu_fwd_kernel" in e[2]]
        L = len(dil)
        per = collections.defaultdict(list)
        for i, (s0, e0, name) in enumerate(sel[len(sel) % L:] if len(sel) % L else sel):
            per[i % L].append((e0 - s0) / 1e3)
        for l in range(L):
            if per[l]:
                v = sorted(per[l])
                rows.append((tag, l, dil[l], len(v), sum(v) / len(v), v[0], v[len(v) // 2], v[-1]))
with open("$OUT/glu_layer_durations.csv", "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["launch_kind", "layer", "dilation", "launches", "mean_us", "min_us", "median_us", "max_us"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], r[3]] + ["%.2f" % x for x in r[4:]])
PY
for f in $OUT/stats/*/*kernel_stats.csv; do cp $f $OUT/kernel_stats.csv; done
for f in $OUT/stats_fwd/*/*kernel_stats.csv; do cp $f $OUT/forward_kernel_stats.csv; done
rm -rf $OUT/stats $OUT/stats_fwd $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
