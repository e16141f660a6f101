#!/usr/bin/env python
"""The backward sweep's residual / gate launches (wae_gemm_tm modes 1 and 2) on the 8-wave schedule of csrc/gemm_tm8.hip (gemm_tm8x_kernel)
against the generic kernel (WAE_TM_ONE_WG): microseconds per launch over random operands, HIP events around 50 launches.
    python tools/time_tm8x.py            # the C2, hps/vqwae.json and C5 shapes"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavenet_autoencoders_amd import _lib as L  # noqa: E402
from wavenet_autoencoders_amd import packing as P  # noqa: E402

lib = L.lib()
dev = torch.device("cuda:0")


def tm(dt, B, T, M, mode, alpha, srcs, w, out, out_stride, aux, aux_stride, flags):
    d = L.TmDesc(dt, B, T, M, len(srcs), mode, alpha, flags)
    n = len(srcs)
    ptrs = (ctypes.c_void_p * n)(*[s[0].data_ptr() for s in srcs])
    strides = (ctypes.c_int64 * n)(*[s[1] for s in srcs])
    cols = (ctypes.c_int32 * n)(*[s[2] for s in srcs])
    shifts = (ctypes.c_int32 * n)(*[s[3] for s in srcs])
    L.check(lib.wae_gemm_tm(ctypes.byref(d), ptrs, strides, cols, shifts, L.ptr(w), L.ptr(out), out_stride, L.ptr(aux), aux_stride, None), "gemm_tm")


def timed(fn, n=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    td, dt = torch.bfloat16, L.WAE_BF16
    for name, B, T, R, S, Hp in (("c2", 8, 8000, 256, 256, 192), ("c3", 8, 5120, 256, 256, 128), ("c5", 16, 5120, 512, 512, 256),
                                 ("c5 half batch", 8, 5120, 512, 512, 256)):
        Z2 = 2 * Hp
        dz = (torch.randn(B, T, Z2, device=dev) * 0.5).to(td)
        gn = (torch.randn(B, T, R, device=dev) * 0.5).to(td)
        ds = (torch.randn(B, T, S, device=dev) * 0.5).to(td)
        z = (torch.randn(B, T, Z2, device=dev)).to(td)
        out_r = torch.empty(B, T, R, device=dev, dtype=td)
        out_g = torch.empty(B, T, Z2, device=dev, dtype=td)
        w_r = (torch.randn(3 * (Z2 // 64) * (R // 32) * 4 * 512, device=dev) * 0.02).to(td)
        w_g = (torch.randn((R // 64 + S // 64) * (Hp // 32) * 4 * 512, device=dev) * 0.02).to(td)
        srcs_r = [(dz, Z2, Z2, (2 - tap) * 4) for tap in range(3)]
        srcs_g = [(gn, R, R, 0), (ds, S, S, 0)]
        res = {}
        for tag, fl in (("generic", L.TM_ONE_WG), ("8-wave", 0)):
            res[tag] = (timed(lambda: tm(dt, B, T, R, 1, 0.7071, srcs_r, w_r, out_r, R, gn, R, P.TM_INTERLEAVE | fl)),
                        timed(lambda: tm(dt, B, T, Hp, 2, 1.0, srcs_g, w_g, out_g, Z2, z, Z2, fl)))
        print(f"{name:14s} residual: generic {res['generic'][0]:6.1f} us, 8-wave {res['8-wave'][0]:6.1f} us   gate: generic {res['generic'][1]:6.1f} us, "
              f"8-wave {res['8-wave'][1]:6.1f} us", flush=True)


if __name__ == "__main__":
    main()
