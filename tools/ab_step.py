#!/usr/bin/env python
"""Same-box A/B of one debug switch of libwae_hip.so over the C2 train step: ab_step.py <setter> <valA> <valB> [rounds]."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wae_oracle as O  # noqa: E402
from wavenet_autoencoders_amd import Geometry  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

C2 = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5], cin_pad=0)
B, T = 8, 8000
setter, va, vb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
eng = WaeEngine(Geometry.from_cfg(C2), dtype="bf16")
eng.load_state_dict(O.make_state_dict(dict(C2), salt=5, with_encoder=False))
eng.init_optimizer()
x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
lat = torch.randn(B, 64, T // 320, device="cuda")
g = torch.randint(0, 153, (B,), device="cuda")
fn = getattr(eng.lib, setter)
fn.argtypes = [ctypes.c_int]
fn.restype = None
for v in (va, vb):
    fn(v)
    for _ in range(3):
        res = eng.train_step(x, lat, g)
    print(f"{setter}({v}): loss {float(res['loss']):.5f}")
for r in range(rounds):
    for v in (va, vb):
        fn(v)
        eng.train_step(x, lat, g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            eng.train_step(x, lat, g)
        e1.record()
        torch.cuda.synchronize()
        print(f"round {r} {setter}({v}): {e0.elapsed_time(e1) / 10:.3f} ms/step", flush=True)
