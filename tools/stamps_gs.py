#!/usr/bin/env python
"""Diagnostic: phase timeline of one glu_fwd_static workgroup (s_memtime stamps; libwae_gsstamps.so = glu_fwd.hip built with
-DWAE_DEBUG_KNOBS and glu_fwd_static.hip with -DWAE_GLU_STAMPS -DWAE_GS_CONT=0).  Never quote run times from this build."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_autoencoders_amd import Geometry, _lib as L  # noqa: E402
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), sys.argv[1] if len(sys.argv) > 1 else "libwae_gsstamps.so")
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

C2 = dict(layers=2, stacks=1, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None)
B, T = 8, 8000
eng = WaeEngine(Geometry.from_cfg(C2), dtype="bf16")
torch.manual_seed(0)
eng.params.normal_(0, 0.05)
eng.prepare_weights()
g = eng.g
x = (torch.randn(B, T, g.Rp, device="cuda") * 0.5).to(eng.tdtype)
c = (torch.randn(B, T, g.Ccp, device="cuda") * 0.5).to(eng.tdtype)
xo = torch.empty_like(x)
ubuf = torch.zeros(B, T, g.Hp, device="cuda").to(eng.tdtype)
zs = torch.zeros(B, T, 2 * g.Hp, device="cuda").to(eng.tdtype)
zb = torch.zeros(B, 2 * g.Hp, device="cuda")
st = eng.stream()
nwg = B * ((T + 255) // 256)
stamps = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
lib = eng.lib
lib.wae_debug_set_stamps.argtypes = [ctypes.c_void_p]


def run(fl, d):
    desc = L.GluDesc(eng.dt, B, T, g.Rp, g.Ccp, g.Hp, g.k, d, fl)
    L.check(lib.wae_glu_layer_fwd(ctypes.byref(desc), L.ptr(x), L.ptr(xo), L.ptr(c), L.ptr(ubuf), g.Hp, L.ptr(zb), 0,
                                  L.ptr(zs) if fl & 2 else None, L.ptr(eng.w_glu), L.ptr(eng.b_glu), st))


for fl in (0, 2):
    for d in (1, 64):
        for _ in range(20):
            run(fl, d)
        lib.wae_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
        for _ in range(3):
            run(fl, d)
        torch.cuda.synchronize()
        lib.wae_debug_set_stamps(None)
        full = stamps.cpu().numpy().reshape(nwg, 64)
        s = full[:, :16].astype(np.int64)
        names = ["prologue (tables, first requests)", "GEMM1 pass 0", "z/gate/u-store pass 0", "GEMM1 pass 1", "z/gate/u-store pass 1",
                 "GEMM2: counted wait", "GEMM2: barrier", "GEMM2: MFMAs", "GEMM2: residual + x' stores"]
        idx = [0, 1, 2, 3, 4, 5, 9, 10, 11, 8]
        if not s[:, 4].any():            # the one-pass form (Hp = 192 at C2) has no second pass: stamps 4 and 5 are never written
            names = names[:3] + names[5:]
            idx = [0, 1, 2, 3, 9, 10, 11, 8]
        print(f"--- flags={fl} dilation={d}: s_memtime ticks per phase of wave 0, median / p10 / p90 over {nwg} workgroups")
        for i, nme in enumerate(names):
            v = np.sort(s[:, idx[i + 1]] - s[:, idx[i]])
            print(f"  {nme:38s} {int(np.median(v)):8d} {int(v[len(v)//10]):8d} {int(v[len(v)*9//10]):8d}")
        tot = s[:, 8] - s[:, 0]
        rt = s[:, 15] - s[:, 14]
        print(f"  total ticks median {int(np.median(tot))}; workgroup life {np.median(rt) / 100.0:.2f} us; clock {np.median(tot / np.maximum(rt, 1)) * 100:.0f} MHz; "
              f"launch span {(s[:, 15].max() - s[:, 14].min()) / 100.0:.2f} us, last start {(s[:, 14].max() - s[:, 14].min()) / 100.0:.2f} us after first")
        pw = full[:, 16:16 + 32].reshape(nwg, 8, 4).astype(np.int64)
        print("  per wave (median): vmcnt wait / barrier wait / GEMM-1 total (26 chunks) / life")
        for w in range(8):
            print(f"    wave {w}: " + "  ".join(f"{int(np.median(pw[:, w, i])):7d}" for i in range(4)))
