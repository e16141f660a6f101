#!/usr/bin/env python
"""Time the C2 train step (forward + backward + clip/Adam/EMA) and its parts."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wae_oracle as O  # closed-form weights only
from wavenet_autoencoders_amd import Geometry  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

C2 = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5], cin_pad=0)
B, T = 8, 8000
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
eng = WaeEngine(Geometry.from_cfg(C2), dtype=dtype)
eng.load_state_dict(O.make_state_dict(dict(C2), salt=5, with_encoder=False))
x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
lat = torch.randn(B, 64, T // 320, device="cuda")
g = torch.randint(0, 153, (B,), device="cuda")
for _ in range(3):
    r = eng.train_step(x, lat, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    r = eng.train_step(x, lat, g)
host = (time.perf_counter() - t0) / steps      # the host's issue time per step (it runs ahead of the GPU when this is the smaller)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"host issue time {host * 1e3:.2f} ms per step")
print(f"train step {dtype}: {dt * 1e3:.2f} ms -> {B * T / dt / 1e6:.2f} M samples/s; loss {float(r['loss']):.4f} gnorm {float(r['grad_norm']):.4f}")
print("peak memory GB", torch.cuda.max_memory_allocated() / 2**30)
