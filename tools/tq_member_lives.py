#!/usr/bin/env python
"""Lives of the weight-gradient launch's workgroups by team member, PRODUCT build (gemm_tn_static writes [team, member, life in 10-ns
ticks] of every workgroup when the launch is handed a stamps buffer): which job kind ends the launch.
Usage: python tools/tq_member_lives.py [c2|c3]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine

conf = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
dev = torch.device("cuda:0")
eng = WaeEngine(Geometry.from_cfg(conf["cfg"]), dtype=conf["dtype"], device=str(dev))
eng.load_state_dict(O.make_state_dict(dict(conf["cfg"]), salt=conf["salt"], with_encoder=conf["encoder"]))
x, lat, g = bench.synth_inputs(0, dev, conf)
xi = x.to(torch.int32)
eng.init_optimizer()
for _ in range(3):
    eng.train_step(xi, lat, g)
st = eng._ws[("bwd", conf["B"], conf["T"])]["stream"]
st.stamps = torch.zeros(st.nwg * 8, dtype=torch.int64, device=dev)
eng.train_step(xi, lat, g)
torch.cuda.synchronize()
s = st.stamps.cpu().numpy().reshape(st.nwg, 8)
st.stamps = None
names = {k: n for k, n in enumerate(["tap 0", "tap 1", "tap 2", "cond", "out+skip", "member 5"])}
print(f"{st.nwg} workgroups, {st.nteams} teams of {st.team_size}")
for m in range(st.team_size):
    v = s[s[:, 6] == m][:, 7] / 100.0
    if len(v):
        print(f"  member {m} ({names.get(m, '?'):8s}): life median {np.median(v):7.1f} us, min {v.min():7.1f}, max {v.max():7.1f}  ({len(v)} workgroups)")
print(f"  launch ends with the slowest workgroup: {s[:, 7].max() / 100.0:.1f} us")
