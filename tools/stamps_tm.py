#!/usr/bin/env python
"""Diagnostic: where a gemm_tm workgroup spends its clocks (libwae_tmstamps.so = gemm_tm.hip built with -DWAE_TM_STAMPS); the two
per-layer backward launches of layer 5 at C2.  Never quote run times from this build."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavenet_autoencoders_amd import _lib as L  # noqa: E402
L.LIB_PATH = os.path.join(ROOT, "wavenet_autoencoders_amd", "libwae_tmstamps.so")
from oracle import wae_oracle as O  # noqa: E402
from wavenet_autoencoders_amd import Geometry, backward as BW  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402
import bench  # noqa: E402

B, T = 8, 8000
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16")
eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
lat = torch.randn(B, 64, T // 320, device="cuda")
gid = torch.randint(0, 153, (B,), device="cuda")
eng.init_optimizer()
eng.train_step(x, lat, gid)
nwg = B * ((T + 127) // 128)      # upper bound (the 8-wave shape launches half as many)
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
eng.lib.wae_debug_set_tm_stamps.argtypes = [ctypes.c_void_p]
ku, kx = BW._debug_kernels(eng, B, T, 5)
for name, k in (("du/dz (GATE_BWD)", ku), ("dx (RESIDUAL)", kx)):
    for _ in range(10):
        k()
    eng.lib.wae_debug_set_tm_stamps(ctypes.c_void_p(buf.data_ptr()))
    k()
    torch.cuda.synchronize()
    eng.lib.wae_debug_set_tm_stamps(None)
    s = buf.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    s = s[s[:, 6] > 0]
    med = lambda v: float(np.median(v))  # noqa: E731
    print(f"{name}: {len(s)} workgroups; life {med(s[:, 1]) / 100:.1f} us = {med(s[:, 0]):.0f} clocks ({med(s[:, 0] / s[:, 1]) * 100:.0f} MHz); "
          f"chunk loop {med(s[:, 2]):.0f} ({med(s[:, 2] / s[:, 6]):.0f} per chunk x {int(med(s[:, 6]))}), epilogue {med(s[:, 3]):.0f}; "
          f"wave 0 per chunk: counted wait {med(s[:, 4] / s[:, 6]):.0f}, barrier {med(s[:, 5] / s[:, 6]):.0f}")
