"""The wave-specialised form of the time-major GEMM (csrc/gemm_tm8.hip: 8 consumer + 4 loader waves; the head's skip contraction,
wavenet.py:204-209) against the generic kernel it replaces (csrc/gemm_tm.hip, forced by WAE_TM_ONE_WG): same packed weight stream, same
accumulation order, so the outputs are compared BITWISE; the generic kernel is the one the oracle comparisons of
tests/test_gpu_parity.py / test_gpu_wide.py pin.  (K = 192: an odd chunk count, which the new kernel leaves to the generic one.)"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib, L, dt, B, T, K, M, u, w, bias, alpha, flags):
    """wae_gemm_tm mode 3: out (B, T, M) = relu(alpha * (bias + W u)); M > 256 runs in slices of 256 rows (the wide head of config C5)"""
    out = torch.full((B, T, M), float("nan"), dtype=u.dtype, device=u.device)
    d = L.TmDesc(dt, B, T, M, 1, 3, alpha, flags)
    ptrs = (ctypes.c_void_p * 1)(u.data_ptr())
    strides = (ctypes.c_int64 * 1)(K)
    cols = (ctypes.c_int32 * 1)(K)
    shifts = (ctypes.c_int32 * 1)(0)
    L.check(lib.wae_gemm_tm(ctypes.byref(d), ptrs, strides, cols, shifts, L.ptr(w), L.ptr(out), M, L.ptr(bias), 0, None), "gemm_tm")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T,K,M", [(1, 1000, 256, 256), (3, 777, 2560, 256), (2, 8000, 4608, 256), (1, 33, 512, 256), (2, 300, 384, 256),
                                     (1, 257, 192, 256), (2, 900, 1024, 512)])
def test_skip_contraction_static_schedule_is_bitwise_the_generic_kernel(dtype, B, T, K, M):
    from wavenet_autoencoders_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda:0")
    td, dt = (torch.bfloat16, L.WAE_BF16) if dtype == "bf16" else (torch.float16, L.WAE_F16)
    gen = torch.Generator(device="cpu").manual_seed(B * 100003 + T * 17 + K)
    u = (torch.randn(B, T, K, generator=gen) * 0.5).to(td).to(dev)
    w = (torch.randn((K // 64) * (M // 32) * 4 * 64 * 8, generator=gen) * (1.0 / K ** 0.5)).to(td).to(dev)   # packed fragment stream
    bias = torch.randn(M, generator=gen).to(dev)
    old = _run(lib, L, dt, B, T, K, M, u, w, bias, 0.25, L.TM_ONE_WG)
    assert old.float().abs().max() > 0
    # (repeated: the kernel's loader and consumer waves meet only at barriers -- a missing one shows as an occasional stale tile, which is
    #  what the first build of the wave-specialised form had at its very first operand tile)
    for _ in range(12):
        new = _run(lib, L, dt, B, T, K, M, u, w, bias, 0.25, 0)
        assert not torch.isnan(new.float()).any()
        assert torch.equal(new.view(torch.int16), old.view(torch.int16))


def _tm(lib, L, dt, B, T, M, mode, alpha, srcs, w, out, out_stride, aux, aux_stride, flags):
    d = L.TmDesc(dt, B, T, M, len(srcs), mode, alpha, flags)
    n = len(srcs)
    ptrs = (ctypes.c_void_p * n)(*[s[0].data_ptr() for s in srcs])
    strides = (ctypes.c_int64 * n)(*[s[1] for s in srcs])
    cols = (ctypes.c_int32 * n)(*[s[2] for s in srcs])
    shifts = (ctypes.c_int32 * n)(*[s[3] for s in srcs])
    L.check(lib.wae_gemm_tm(ctypes.byref(d), ptrs, strides, cols, shifts, L.ptr(w), L.ptr(out), out_stride, L.ptr(aux), aux_stride, None), "gemm_tm")
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T,cols,M,dil,taps", [(2, 1000, 384, 256, 3, 3), (1, 777, 256, 256, 64, 3), (2, 2300, 512, 512, 16, 3),
                                                 (1, 300, 128, 256, 1, 3), (3, 515, 256, 256, 512, 2), (1, 257, 192, 256, 2, 3)])
def test_residual_backward_on_the_8_wave_schedule_is_bitwise_the_generic_kernel(dtype, B, T, cols, M, dil, taps):
    """wae_gemm_tm mode 1 with TM_INTERLEAVE (dx-hat of a layer: the taps of the transposed dilated convolution over shifted rows of
    dz -- rows past the clip's end are zeros -- plus the residual, autograd of modules.py:131,159-161): csrc/gemm_tm8.hip's
    gemm_tm8x_kernel against the generic kernel (WAE_TM_ONE_WG).  dz is a column slice of a wider array, as in the sweep; M = 512 runs as
    two slices of 256 output rows; (192 columns x 3 taps: an odd chunk count, left to the generic kernel)."""
    from wavenet_autoencoders_amd import _lib as L
    from wavenet_autoencoders_amd import packing as P
    lib = L.lib()
    dev = torch.device("cuda:0")
    td, dt = (torch.bfloat16, L.WAE_BF16) if dtype == "bf16" else (torch.float16, L.WAE_F16)
    gen = torch.Generator(device="cpu").manual_seed(B * 100003 + T * 17 + cols + dil)
    stride = cols + 128
    dz = (torch.randn(B, T, stride, generator=gen) * 0.5).to(td).to(dev)
    res = (torch.randn(B, T, M, generator=gen) * 0.5).to(td).to(dev)
    nq = taps * (cols // 64)
    w = (torch.randn(nq * (M // 32) * 4 * 512, generator=gen) * (1.0 / (taps * cols) ** 0.5)).to(td).to(dev)
    dzv = dz[:, :, 64:]                                      # the layer's columns inside the wider array
    srcs = [(dzv, stride, cols, (taps - 1 - tap) * dil) for tap in range(taps)]
    outs = []
    for flags in (P.TM_INTERLEAVE | L.TM_ONE_WG, P.TM_INTERLEAVE):
        for _ in range(1 if flags & L.TM_ONE_WG else 8):
            out = torch.full((B, T, M), float("nan"), dtype=td, device=dev)
            _tm(lib, L, dt, B, T, M, 1, 0.70710678, srcs, w, out, M, res, M, flags)
            assert not torch.isnan(out.float()).any()
            outs.append(out)
    assert outs[0].float().abs().max() > 0
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16))


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T,R,S,Hp", [(2, 1000, 256, 256, 192), (1, 777, 256, 256, 128), (2, 1300, 512, 512, 256), (2, 600, 0, 256, 192),
                                        (1, 290, 256, 128, 128)])
def test_gate_backward_on_the_8_wave_schedule_is_bitwise_the_generic_kernel(dtype, B, T, R, S, Hp):
    """wae_gemm_tm mode 2 (du = W_out^T dx-hat + W_skip^T dskip, then the gate derivative from the saved pre-activations: autograd of
    modules.py:154-157): gemm_tm8x_kernel against the generic kernel.  R = 0: the top layer's launch (dskip alone); the output is a column
    slice of the all-layers dz array."""
    from wavenet_autoencoders_amd import _lib as L
    lib = L.lib()
    dev = torch.device("cuda:0")
    td, dt = (torch.bfloat16, L.WAE_BF16) if dtype == "bf16" else (torch.float16, L.WAE_F16)
    gen = torch.Generator(device="cpu").manual_seed(B * 100003 + T * 17 + R + Hp)
    gn = (torch.randn(B, T, max(R, 64), generator=gen) * 0.5).to(td).to(dev)
    ds = (torch.randn(B, T, S, generator=gen) * 0.5).to(td).to(dev)
    z = (torch.randn(B, T, 2 * Hp, generator=gen) * 1.5).to(td).to(dev)
    nq = R // 64 + S // 64
    w = (torch.randn(nq * (Hp // 32) * 4 * 512, generator=gen) * (1.0 / (R + S) ** 0.5)).to(td).to(dev)
    srcs = ([(gn, R, R, 0)] if R else []) + [(ds, S, S, 0)]
    ostride = 3 * 2 * Hp
    outs = []
    for flags in (L.TM_ONE_WG, 0):
        for _ in range(1 if flags else 8):
            out = torch.zeros(B, T, ostride, dtype=td, device=dev)
            _tm(lib, L, dt, B, T, Hp, 2, 1.0, srcs, w, out[:, :, 2 * Hp:], ostride, z, 2 * Hp, flags)
            outs.append(out)
    assert outs[0].float().abs().max() > 0
    for o in outs[1:]:
        assert torch.equal(o.view(torch.int16), outs[0].view(torch.int16))
