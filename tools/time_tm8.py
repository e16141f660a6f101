#!/usr/bin/env python
"""Times the head's skip contraction (wae_gemm_tm mode 3, K = Ku) at the C2 / C3 shapes: the static 8-wave kernel (csrc/gemm_tm8.hip)
against the generic one (WAE_TM_ONE_WG off = two workgroups per CU; on = one).  HIP events around 100 back-to-back launches."""
import ctypes, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavenet_autoencoders_amd import _lib as L  # noqa: E402
if len(sys.argv) > 1:
    L.LIB_PATH = os.path.abspath(sys.argv[1])
lib = L.lib()
dev = torch.device("cuda:0")
for name, B, T, K, M in (("C2", 8, 8000, 4608, 256), ("C3", 8, 5120, 2560, 256), ("C5 (two output slices)", 16, 5120, 12288, 512)):
    u = (torch.randn(B, T, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn((K // 64) * (M // 32) * 4 * 64 * 8, device=dev) / K ** 0.5).to(torch.bfloat16)
    bias = torch.randn(M, device=dev)
    out = torch.empty(B, T, M, dtype=torch.bfloat16, device=dev)
    ptrs = (ctypes.c_void_p * 1)(u.data_ptr()); strides = (ctypes.c_int64 * 1)(K)
    cols = (ctypes.c_int32 * 1)(K); shifts = (ctypes.c_int32 * 1)(0)
    res = []
    for label, flags in (("static 8-wave (tm8)", 0), ("generic, one workgroup per CU", L.TM_ONE_WG))[:1 if len(sys.argv) > 2 else 2]:
        d = L.TmDesc(L.WAE_BF16, B, T, M, 1, 3, 0.2, flags)
        k = lambda: L.check(lib.wae_gemm_tm(ctypes.byref(d), ptrs, strides, cols, shifts, L.ptr(w), L.ptr(out), M, L.ptr(bias), 0, None))
        for _ in range(5):
            k()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            k()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 100 * 1e3
        gb = (B * T * K * 2 + B * T * M * 2) / 1e9
        print(f"{name} {label:32s} {us:7.1f} us  {gb / us * 1e6 / 1e3:.2f} TB/s on {gb * 1e3:.0f} MB", flush=True)
