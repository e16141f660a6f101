#!/bin/bash
# Round profile on the GPU box: tools/profile_round.sh <tag>  -> gpurun_out/<tag>/ (bench lines, kernel stats, PMC FETCH/WRITE passes)
TAG=${1:-rXX}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
python3 bench.py > $OUT/bench_train.json 2> $OUT/bench_train.err
python3 bench.py --mode forward --no-cpu --no-ar > $OUT/bench_forward.json 2> $OUT/bench_forward.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/bench.py --steps 10 --warmup 3 --no-cpu --no-ar > $OUT/profiled_line.json 2> $OUT/stats.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fwd -- python3 $ROOT/bench.py --mode forward --steps 10 --warmup 3 --no-cpu --no-ar > $OUT/profiled_forward_line.json 2> $OUT/stats_fwd.err
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-ar > $OUT/pmc_$c.log 2>&1
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
            agg.setdefault(r["Kernel_Name"].split("(")[0], []).append(float(r["Counter_Value"]))
    with open("$OUT/pmc_%s_per_kernel_mean_kb.csv" % c, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches", "mean_kb"])
        for k, v in agg.items():
            w.writerow([k, len(v), "%.1f" % (sum(v) / len(v))])
            res.setdefault(k, {})[c] = sum(v) / len(v)
out = {k: {"fetch_kb_raw": v.get("FETCH_SIZE", 0.0), "write_kb": v.get("WRITE_SIZE", 0.0),
           "hbm_bytes_per_launch": (2 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024} for k, v in res.items()}
json.dump({"csrc_hash": bench.csrc_hash(), "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-ar (two separate passes)",
           "correction": "MI355X_MICROARCH.md section HBM: FETCH_SIZE counts 1/2 of the bytes of 16-B/lane streaming reads on gfx950 -> doubled; WRITE_SIZE exact; unit KB -> x1024",
           "kernels": out}, open("$OUT/pmc_traffic.json", "w"), indent=1)
PY
for f in $OUT/stats/*/*kernel_stats.csv; do cp $f $OUT/kernel_stats.csv; done
for f in $OUT/stats_fwd/*/*kernel_stats.csv; do cp $f $OUT/forward_kernel_stats.csv; done
rm -rf $OUT/stats $OUT/stats_fwd $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
