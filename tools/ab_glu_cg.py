#!/usr/bin/env python
"""A/B of wae_glu_layer_fwd shapes at the C2 layer (8 x 8000 samples): 8 waves x 32 columns (default), 4 waves x 32 (two
workgroups per CU), 4 waves x 64 columns (WAE_GLU_CG2: every A fragment feeds two MFMAs).  Outputs are compared bitwise."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_autoencoders_amd import Geometry, _lib as L  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
G = int(sys.argv[2]) if len(sys.argv) > 2 else 368
R = int(sys.argv[3]) if len(sys.argv) > 3 else 256
C2 = dict(layers=2, stacks=1, R=R, G=G, S=R, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None)
B, T = 8, 8000
eng = WaeEngine(Geometry.from_cfg(C2), dtype=dtype)
torch.manual_seed(0)
eng.params.normal_(0, 0.05)
eng.prepare_weights()
g = eng.g
x = (torch.randn(B, T, g.Rp, device="cuda") * 0.5).to(eng.tdtype)
c = (torch.randn(B, T, g.Ccp, device="cuda") * 0.5).to(eng.tdtype)
zb = torch.randn(B, 2 * g.Hp, device="cuda") * 0.1
st = eng.stream()


def run(xo, ub, zs, flags, d):
    desc = L.GluDesc(eng.dt, B, T, g.Rp, g.Ccp, g.Hp, g.k, d, flags)
    L.check(eng.lib.wae_glu_layer_fwd(ctypes.byref(desc), L.ptr(x), L.ptr(xo), L.ptr(c), L.ptr(ub), g.Hp, L.ptr(zb), 2 * g.Hp,
                                      L.ptr(zs) if flags & 2 else None, L.ptr(eng.w_glu), L.ptr(eng.b_glu), st))


shapes = {"8x32": 0, "4x32": L.GLU_WAVES4, "4x64": L.GLU_CG2}
if len(sys.argv) > 4:
    shapes = {k: v for k, v in shapes.items() if k in sys.argv[4].split(",")}
outs = {}
for name, fl in shapes.items():
    for d in (1, 512):
        xo, ub, zs = torch.zeros_like(x), torch.zeros(B, T, g.Hp, device="cuda").to(eng.tdtype), torch.zeros(B, T, 2 * g.Hp, device="cuda").to(eng.tdtype)
        print("running", name, d, flush=True)
        run(xo, ub, zs, fl | 2, d)
        torch.cuda.synchronize()
        outs[(name, d)] = (xo, ub, zs)
for name in shapes:
    for d in (1, 512):
        ref, got = outs[(list(shapes)[0], d)], outs[(name, d)]
        print(f"{name} d={d}: bitwise equal to 8x32: x' {torch.equal(ref[0], got[0])} u {torch.equal(ref[1], got[1])} z {torch.equal(ref[2], got[2])}"
              f"  max|dx'| {(ref[0].float() - got[0].float()).abs().max().item():.3e}")
xo, ub, zs = torch.zeros_like(x), torch.zeros(B, T, g.Hp, device="cuda").to(eng.tdtype), torch.zeros(B, T, 2 * g.Hp, device="cuda").to(eng.tdtype)
for rnd in range(3):
    for name, fl in shapes.items():
        for flags in (0, 2):
            for d in (1, 64):
                for _ in range(3):
                    run(xo, ub, zs, fl | flags, d)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    run(xo, ub, zs, fl | flags, d)
                e1.record()
                torch.cuda.synchronize()
                print(f"round {rnd} {name} save_z={bool(flags)} d={d}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
