#!/usr/bin/env python
"""Same-process A/B of the two per-layer wae_gemm_tm launches of the backward pass at the C2 shape: the default two-workgroup register-operand shape against the
8-wave LDS-staged operand shape (WAE_TM_BLDS): outputs bitwise, then us/launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import wae_oracle as O  # noqa: E402
from wavenet_autoencoders_amd import Geometry, _lib as L, backward as BW  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402
import bench  # noqa: E402

B, T = 8, 8000
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype=os.environ.get("AB_DTYPE", "bf16"))
eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
lat = torch.randn(B, 64, T // 320, device="cuda")
gid = torch.randint(0, 153, (B,), device="cuda")
eng.init_optimizer()
eng.train_step(x, lat, gid)
ws = eng._ws[("bwd", B, T)]
g = eng.g


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for l in (0, 3, 9, 11, 14):
    d = g.dilations[l]
    outs = {}
    for name, fl in (("regs", 0), ("blds", L.TM_BLDS)):
        ku, kx = BW._debug_kernels(eng, B, T, l, fl, fl)
        ku()
        torch.cuda.synchronize()
        dz = ws["dz"].clone()
        kx()
        torch.cuda.synchronize()
        gx = ws["gx"][l % len(ws["gx"])].clone()
        outs[name] = (dz, gx, timeit(ku), timeit(kx))
    same = [bool(torch.equal(a.view(torch.int16), b_.view(torch.int16))) for a, b_ in zip(outs["regs"][:2], outs["blds"][:2])]
    print(f"layer {l:2d} d={d:4d}: dz / gx bitwise equal {same};  du/dz {outs['regs'][2]:6.1f} -> {outs['blds'][2]:6.1f} us   "
          f"dx {outs['regs'][3]:6.1f} -> {outs['blds'][3]:6.1f} us", flush=True)
