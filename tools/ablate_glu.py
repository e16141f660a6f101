#!/usr/bin/env python
"""Timing-only ablation of wae_glu_layer_fwd at the C2 shape (run on the GPU box).
Needs the library built with the ablation bits compiled in:
    touch wavenet_autoencoders_amd/csrc/glu_fwd.hip && make -C wavenet_autoencoders_amd/csrc EXTRA=-DWAE_GLU_ABLATE
Usage: ablate_glu.py [bf16|fp32] [waves per workgroup: 4|8]
Interleaved rounds in one process (cdna_hip_programming.md 5.4 rule 24)."""
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
xo = torch.empty_like(x)
ubuf = torch.zeros(B, T, g.Hp, device="cuda").to(eng.tdtype)
zb = torch.zeros(B, 2 * g.Hp, device="cuda")
st = eng.stream()
pass  # shape: eng.glu_flags (wae_glu_desc.flags)
eng.glu_flags = 8 if (int(sys.argv[2]) if len(sys.argv) > 2 else 4) == 4 else 0
variants = {"full": 0, "no_dma": 0x100, "no_bload": 0x200, "no_epi": 0x400, "no_gate": 0x800,
            "no_dma_bload": 0x300, "coalesced_b": 0x1000, "coalesced_b_no_dma": 0x1100, "mfma_only": 0xF00}


def run(flags, d):
    desc = L.GluDesc(eng.dt, B, T, g.Rp, g.Ccp, g.Hp, g.k, d, flags)
    L.check(eng.lib.wae_glu_layer_fwd(ctypes.byref(desc), L.ptr(x), L.ptr(xo), L.ptr(c), L.ptr(ubuf), g.Hp, L.ptr(zb), 0, None,
                                      L.ptr(eng.w_glu), L.ptr(eng.b_glu), st))


res = {k: [] for k in variants}
for d in (1,):
    for rnd in range(5):
        for name, fl in variants.items():
            for _ in range(2):
                run(fl, d)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(fl, d)
            e1.record()
            torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"dilation {d}:")
    for name, v in res.items():
        v = sorted(v)
        print(f"  {name:14s} median {v[len(v)//2]:8.1f} us   min {v[0]:8.1f} us")
    res = {k: [] for k in variants}
