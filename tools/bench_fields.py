#!/usr/bin/env python
"""bench.py under a variant library (WAE_LIB_PATH), a few fields of its line: tools/bench_fields.py <lib.so> [bench args]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.abspath(sys.argv[1])
out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-ar", "--no-fp32", "--steps", "20"] + sys.argv[2:],
                     env=dict(os.environ, WAE_LIB_PATH=lib), capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(os.path.basename(lib), "FAILED", out.stderr[-800:])
    sys.exit(1)
d = json.loads(line[-1])
f = lambda k: ("%.1f" % (d[k]["avg_launch_ms"] * 1e3)) if k in d else "-"
print(f"{os.path.basename(lib):28s} step {d['ms_per_step']:.3f} ms  pair {f('roofline_bwd_pair')} us  wgrad {f('roofline_wgrad')} us  glu_z {f('roofline_glu_fwd_z')} us  "
      f"fwd {d.get('forward_inference', {}).get('ms_per_step', 0):.3f} ms  loss {d['loss']:.4f}", flush=True)
