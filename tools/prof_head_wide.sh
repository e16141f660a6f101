#!/bin/bash
# kernel stats of the C2 inference forward with the head run as wae_gemm_tm launches (WAE_HEAD_WIDE=1) and fused
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
for w in 0 1; do
  export WAE_HEAD_WIDE=$w
  rm -rf /tmp/hw$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hw$w -- python3 $ROOT/bench.py --mode forward --steps 10 --warmup 3 --no-cpu --no-ar > /tmp/hw$w.log 2>&1
  echo "== WAE_HEAD_WIDE=$w"
  python3 - /tmp/hw$w <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_stats.csv")[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:8]:
    print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:8.1f}')
PY
done
