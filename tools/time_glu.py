#!/usr/bin/env python
"""Time wae_glu_layer_fwd at the C2 shape for both workgroup shapes (4 waves x 2 workgroups/CU, 8 waves x 1)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_autoencoders_amd import Geometry, _lib as L  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

C2 = dict(layers=2, stacks=1, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None)
B, T = 8, 8000
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
eng = WaeEngine(Geometry.from_cfg(C2), dtype=dtype)
torch.manual_seed(0)
eng.params.normal_(0, 0.05)
eng.prepare_weights()
g = eng.g
x = (torch.randn(B, T, g.Rp, device="cuda") * 0.5).to(eng.tdtype)
c = (torch.randn(B, T, g.Ccp, device="cuda") * 0.5).to(eng.tdtype)
ubuf = torch.zeros(B, T, g.Hp, device="cuda").to(eng.tdtype)
zsave = torch.zeros(B, T, 2 * g.Hp, device="cuda").to(eng.tdtype)
zb = torch.zeros(B, 2 * g.Hp, device="cuda")
st = eng.stream()
eng.lib.wae_debug_set_glu_waves.argtypes = [ctypes.c_int]


def run(xo, flags, d):
    desc = L.GluDesc(eng.dt, B, T, g.Rp, g.Ccp, g.Hp, g.k, d, flags)
    L.check(eng.lib.wae_glu_layer_fwd(ctypes.byref(desc), L.ptr(x), L.ptr(xo), L.ptr(c), L.ptr(ubuf), g.Hp, L.ptr(zb), 0,
                                      L.ptr(zsave) if flags & 2 else None, L.ptr(eng.w_glu), L.ptr(eng.b_glu), st))


eng.lib.wae_debug_set_glu_slots.argtypes = [ctypes.c_int]
outs = {}
for nw, ns in ((4, 0), (8, 2), (8, 3), (8, 0)):
    eng.lib.wae_debug_set_glu_waves(nw)
    eng.lib.wae_debug_set_glu_slots(ns)
    xo = torch.zeros_like(x)
    run(xo, 2, 4)
    torch.cuda.synchronize()
    outs[(nw, ns)] = (xo.float().clone(), ubuf.float().clone(), zsave.float().clone())
ref = outs[(4, 0)]
for k, v in outs.items():
    print("config", k, "max |diff| vs (4,0): x'", (v[0] - ref[0]).abs().max().item(), "u", (v[1] - ref[1]).abs().max().item(),
          "z", (v[2] - ref[2]).abs().max().item())
xo = torch.zeros_like(x)
for rnd in range(3):
    for nw, ns in ((4, 0), (8, 2), (8, 3), (8, 4), (8, 0)):
        eng.lib.wae_debug_set_glu_waves(nw)
        eng.lib.wae_debug_set_glu_slots(ns)
        for flags in (0, 2):
            for d in (1, 64):
                for _ in range(3):
                    run(xo, flags, d)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(xo, flags, d)
                e1.record()
                torch.cuda.synchronize()
                print(f"round {rnd} NW={nw} slots={ns} save_z={flags >> 1} d={d:3d}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
eng.lib.wae_debug_set_glu_waves(4)
eng.lib.wae_debug_set_glu_slots(0)
