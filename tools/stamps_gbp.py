#!/usr/bin/env python
"""Diagnostic: where a workgroup of the fused residual(l) + gate(l-1) launch (csrc/glu_bwd.hip: glu_bwd_pair_kernel) spends its clocks
(libwae_gbpstamps.so = glu_bwd.hip built with -DWAE_GBP_STAMPS: tools/build_variant.sh gbpstamps glu_bwd.hip "-DWAE_GBP_STAMPS").
C2, bf16, the last fused launch of a train step (layers 1 + 0).  Never quote run times from this build."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["WAE_BWD_FUSED"] = "1"
from wavenet_autoencoders_amd import _lib as L  # noqa: E402
L.LIB_PATH = os.path.join(ROOT, "wavenet_autoencoders_amd", "libwae_gbpstamps.so")
from oracle import wae_oracle as O  # noqa: E402
from wavenet_autoencoders_amd import Geometry  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402
import bench  # noqa: E402

B, T = 8, 8000
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16")
eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
lat = torch.randn(B, 64, T // 320, device="cuda")
gid = torch.randint(0, 153, (B,), device="cuda")
eng.init_optimizer()
for _ in range(3):
    eng.train_step(x, lat, gid)
assert eng.fused_bwd
nwg = B * ((T + 127) // 128)
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
eng.lib.wae_debug_set_gbp_stamps.argtypes = [ctypes.c_void_p]
eng.lib.wae_debug_set_gbp_stamps(ctypes.c_void_p(buf.data_ptr()))
eng.train_step(x, lat, gid)
torch.cuda.synchronize()
eng.lib.wae_debug_set_gbp_stamps(None)
s = buf.cpu().numpy().reshape(nwg, 8).astype(np.float64)
s = s[s[:, 7] > 0]
med = lambda v: float(np.median(v))  # noqa: E731
names = ["prologue", "phase A (18 chunks)", "epilogue A", "phase B1 (W_out^T, operand in registers)", "phase B2 (W_skip^T dS)", "epilogue B (gate)"]
tot = sum(med(s[:, i]) for i in range(6))
print(f"{len(s)} workgroups; life {med(s[:, 6]) / 100:.1f} us = {tot:.0f} clocks ({tot / med(s[:, 6]) * 100:.0f} MHz)")
for i, n in enumerate(names):
    print(f"  {n:44s} {med(s[:, i]):8.0f} clocks  ({100 * med(s[:, i]) / tot:4.1f} %)")
