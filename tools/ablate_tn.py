#!/usr/bin/env python
"""Timing-only ablation of the per-layer weight-gradient launch (gemm_tn_kernel) at the C2 shape."""
import ctypes
import os
import sys

import torch

os.environ["WAE_TN_STREAM"] = "0"   # this tool times the per-layer 128x128-tile launches (the fp32 path)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry, backward as BW  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

C2 = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5], cin_pad=0)
B, T = 8, 8000
eng = WaeEngine(Geometry.from_cfg(C2), dtype="bf16")
eng.load_state_dict(O.make_state_dict(dict(C2), salt=5, with_encoder=False))
x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
lat = torch.randn(B, 64, T // 320, device="cuda")
g = torch.randint(0, 153, (B,), device="cuda")
eng.train_step(x, lat, g)
ws = eng._ws[("bwd", B, T)]
lib = eng.lib
lib.wae_debug_set_tn.argtypes = [ctypes.c_int]
tt = ws["tt_layer"][5]
print("tiles", tt.n, "splits", tt.splits)
for sp in (3, 5, 6, 8):
    tt.splits = sp
    for name, bits in (("full", 0), ("no_loads", 1), ("no_mfma", 2), ("no_atomics", 4), ("loads_only", 6), ("mfma_only", 5)):
        lib.wae_debug_set_tn(bits)
        for _ in range(3):
            tt.launch(B, T)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            tt.launch(B, T)
        e1.record()
        torch.cuda.synchronize()
        print(f"splits {sp} {name:12s} {e0.elapsed_time(e1) / 10 * 1e3:8.1f} us")
lib.wae_debug_set_tn(0)
