#!/usr/bin/env python
"""Same-process A/B of the static weight-gradient launch's team pacing over the C2 train step: tools/ab_tq_pace.py [windows...]
(the pacing window is an argument of wae_gemm_tn_static; the product launches with window 0 unless StaticStreamTable says otherwise)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
wins = [int(a) for a in sys.argv[1:]] or [0, 8]
dev = torch.device("cuda:0")
x, lat, g = bench.synth_inputs(0, dev)
xi = x.to(torch.int32)
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16", device="cuda:0")
eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
eng.init_optimizer()
for _ in range(3):
    eng.train_step(xi, lat, g)
st = eng._ws[("bwd", 8, 8000)]["stream"]
orig_launch = st.launch
st.launch = lambda: (st.pace.zero_(), orig_launch())[1]      # the team words must start at zero in every paced launch
ev = []
eng._tn_events = ev
for rnd in range(3):
    for w in wins:
        st.window = st.window_cond = w
        del ev[:]
        eng.train_step(xi, lat, g)
        del ev[:]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            r = eng.train_step(xi, lat, g)
        e1.record()
        torch.cuda.synchronize()
        tn = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        print(f"round {rnd} window {w}: step {e0.elapsed_time(e1) / 20:.3f} ms, weight-gradient launch {tn:.3f} ms, loss {float(r['loss']):.4f}", flush=True)
