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
