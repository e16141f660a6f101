#!/usr/bin/env python
"""Run wae_glu_layer_fwd N times at the C2 shape (for rocprofv3 counter passes).  Usage: run_glu.py [nw] [slots] [save_z] [n]"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_autoencoders_amd import Geometry, _lib as L  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

C2 = dict(layers=2, stacks=1, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None)
B, T = 8, 8000
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 4
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 0
save_z = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10
eng = WaeEngine(Geometry.from_cfg(C2), dtype="bf16")
torch.manual_seed(0)
eng.params.normal_(0, 0.05)
eng.prepare_weights()
g = eng.g
x = (torch.randn(B, T, g.Rp, device="cuda") * 0.5).to(eng.tdtype)
c = (torch.randn(B, T, g.Ccp, device="cuda") * 0.5).to(eng.tdtype)
xo = torch.zeros_like(x)
ubuf = torch.zeros(B, T, g.Hp, device="cuda").to(eng.tdtype)
zsave = torch.zeros(B, T, 2 * g.Hp, device="cuda").to(eng.tdtype)
zb = torch.zeros(B, 2 * g.Hp, device="cuda")
st = eng.stream()
pass  # shape: eng.glu_flags (wae_glu_desc.flags)
eng.lib.wae_debug_set_glu_slots.argtypes = [ctypes.c_int]
eng.glu_flags = 8 if nw == 4 else 0
eng.lib.wae_debug_set_glu_slots(slots)
for i in range(n):
    desc = L.GluDesc(eng.dt, B, T, g.Rp, g.Ccp, g.Hp, g.k, 4, 2 if save_z else 0)
    L.check(eng.lib.wae_glu_layer_fwd(ctypes.byref(desc), L.ptr(x), L.ptr(xo), L.ptr(c), L.ptr(ubuf), g.Hp, L.ptr(zb), 0,
                                      L.ptr(zsave) if save_z else None, L.ptr(eng.w_glu), L.ptr(eng.b_glu), st))
torch.cuda.synchronize()
