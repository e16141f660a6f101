#!/usr/bin/env python
"""Diagnostic: where the skip contraction of the head (wae_gemm_tm mode 3, K = Ku) spends its clocks, per launch shape
(libwae_tmstamps.so = gemm_tm.hip built with -DWAE_TM_STAMPS).  Never quote run times from this build."""
import ctypes, math, os, sys
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
x, lat, gid = bench.synth_inputs(0, torch.device("cuda:0"))
eng.prepare_weights()
eng.decoder_forward(x, lat, gid, targets=x, lengths=None, want_logits=False, train=False)
ws = eng._ws[(B, T, False)]
g = eng.g
nwg = B * ((T + 127) // 128)
buf = torch.zeros(nwg * 8, dtype=torch.int64, device="cuda")
eng.lib.wae_debug_set_tm_stamps.argtypes = [ctypes.c_void_p]
for name, flags in (("two workgroups per CU", 0), ("one workgroup per CU", L.TM_ONE_WG)):
    k = lambda: BW._tm(eng, B, T, g.Sp, 3, math.sqrt(1.0 / g.layers), [(ws["u"].data_ptr(), g.Ku, g.Ku, 0)], eng.w_head.data_ptr(),
                       ws["h0"].data_ptr(), g.Sp, eng.b_head.data_ptr(), 0, flags=flags)
    for _ in range(5):
        k()
    buf.zero_()
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
    st0 = s[:, 7] - s[:, 7].min()
    late = st0 > 0.25 * med(s[:, 1])
    print(f"    launch span (first start .. last end) {(st0 + s[:, 1]).max() / 100:.1f} us; life p10 / p90 {np.percentile(s[:, 1], 10) / 100:.1f} / "
          f"{np.percentile(s[:, 1], 90) / 100:.1f} us; {int(late.sum())} workgroups start more than a quarter life after the first "
          f"(median start of those {np.median(st0[late]) / 100 if late.any() else 0:.1f} us)")
    full = buf.cpu().numpy().reshape(nwg, 8).astype(np.float64)
    ids = np.nonzero(full[:, 6] > 0)[0]
    print("    median life by XCD (flat workgroup id mod 8): " + " ".join(f"{np.median(full[ids[ids % 8 == k], 1]) / 100:.0f}" for k in range(8)))
    order = np.argsort(full[ids, 1])
    print("    slowest ten workgroups (flat id: life us): " + ", ".join(f"{ids[i]}: {full[ids[i], 1] / 100:.0f}" for i in order[-10:]))
    print("    fastest ten: " + ", ".join(f"{ids[i]}: {full[ids[i], 1] / 100:.0f}" for i in order[:10]))
