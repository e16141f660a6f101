#!/usr/bin/env python
"""Diagnostic: where one glu_fwd workgroup spends its cycles (s_memtime stamps; never quote run times from this)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wavenet_autoencoders_amd import Geometry, _lib as L  # noqa: E402
if os.path.exists(os.path.join(os.path.dirname(L.LIB_PATH), "libwae_stamps.so")):
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libwae_stamps.so")   # built with -DWAE_DEBUG_KNOBS -DWAE_GLU_STAMPS
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

C2 = dict(layers=2, stacks=1, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None)
B, T = 8, 8000
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
flags = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
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
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if NW == 4:
    flags |= L.GLU_WAVES4
nwg = B * ((T + 32 * NW - 1) // (32 * NW))
stamps = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
lib = eng.lib
lib.wae_debug_set_stamps.argtypes = [ctypes.c_void_p]


def run(fl, d=4):
    desc = L.GluDesc(eng.dt, B, T, g.Rp, g.Ccp, g.Hp, g.k, d, fl)
    L.check(lib.wae_glu_layer_fwd(ctypes.byref(desc), L.ptr(x), L.ptr(xo), L.ptr(c), L.ptr(ubuf), g.Hp, L.ptr(zb), 0, None,
                                  L.ptr(eng.w_glu), L.ptr(eng.b_glu), st))


for _ in range(5):
    run(flags)
lib.wae_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
for _ in range(3):
    run(flags)
torch.cuda.synchronize()
lib.wae_debug_set_stamps(None)
full = stamps.cpu().numpy().reshape(nwg, 64)
s = full[:, :16]
names = ["init(zb, first dma/B issue)", "GEMM1 pass 0", "z-save+gate+u-store (+ later passes)", "GEMM2+epilogue"]
d = np.diff(s[:, :5].astype(np.int64), axis=1)
print(f"flags={flags:#x} workgroups={nwg}; s_memtime ticks per phase (median / p10 / p90) over workgroups:")
for i, nme in enumerate(names):
    v = np.sort(d[:, i])
    print(f"  {nme:38s} {int(np.median(v)):8d} {int(v[len(v)//10]):8d} {int(v[len(v)*9//10]):8d}")
tot = (s[:, 4] - s[:, 0]).astype(np.int64)
rt = (s[:, 5] - s[:, 14]).astype(np.int64)   # 100 MHz ticks over the same interval
print(f"  total ticks median {int(np.median(tot))}; workgroup life {np.median(rt) / 100.0:.2f} us; "
      f"s_memtime rate {np.median(tot / np.maximum(rt, 1)) * 100:.0f} MHz")
print("  GEMM-1 chunk loop, wave 0, ticks summed over chunks (median): vmcnt wait", int(np.median(s[:, 6])), " barrier", int(np.median(s[:, 7])),
      " DMA+B issue", int(np.median(s[:, 8])), " ds_read+MFMA", int(np.median(s[:, 9])))
end = s[:, 5].astype(np.int64)
start = s[:, 14].astype(np.int64)
print(f"  launch span {(end.max() - start.min()) / 100.0:.2f} us; starts (us after first): median "
      f"{(np.median(start) - start.min()) / 100.0:.2f}, max {(start.max() - start.min()) / 100.0:.2f}")

pw = full[:, 16:16 + 4 * NW].reshape(nwg, NW, 4).astype(np.int64)
print("  per wave (median over workgroups): vmcnt wait / barrier / issue / ds_read+MFMA")
for w in range(NW):
    print(f"    wave {w}: " + "  ".join(f"{int(np.median(pw[:, w, i])):7d}" for i in range(4)))
