#!/usr/bin/env python
"""Diagnostic: per-workgroup pacing counters of gemm_tn_static (holds, spins, time-outs) on the C2 launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
dev = torch.device("cuda:0")
x, lat, g = bench.synth_inputs(0, dev)
xi = x.to(torch.int32)
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16", device="cuda:0")
eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
eng.init_optimizer()
eng.train_step(xi, lat, g, lengths=None)
torch.cuda.synchronize()
st = eng._ws[("bwd",) + tuple(xi.shape)]["stream"]
st.stamps = torch.zeros(st.nwg * 8, dtype=torch.int64, device=dev)
st.launch()
torch.cuda.synchronize()
s = st.stamps.cpu().numpy().reshape(-1, 8)
s = s[s[:, 7] > 0]
names = {0: "tap 0", 1: "tap 1", 2: "tap 2", 3: "cond", 4: "out+skip"}
print("window", st.window, " workgroups:", len(s), " time-outs total", s[:, 2].sum())
for m in range(st.team_size):
    sel = s[s[:, 6] == m]
    life = sel[:, 7] / 100.0
    print(f" member {m} ({names.get(m, '?'):8s}): life us median {np.median(life):7.1f} p10 {np.percentile(life, 10):7.1f} max {life.max():7.1f}   holds {np.median(sel[:, 0]):6.0f} "
          f"spins {np.median(sel[:, 1]):7.0f} time-outs {sel[:, 2].sum():3d}")
for r in s[s[:, 2] > 0][:6]:
    w = int(r[4]) & 0xffffffff
    print("  time-out: team", r[5], "member", r[6], "pos", int(r[3]) >> 32, "pub", int(r[3]) & 0xffffffff, "word fields", w & 1023, (w >> 10) & 1023, (w >> 20) & 1023)
