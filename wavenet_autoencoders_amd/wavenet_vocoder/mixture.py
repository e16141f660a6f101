"""Discretized mixture of logistics: loss and sampler (reference: wavenet_vocoder/mixture.py:26-156) on the HIP kernels
of csrc/loss.hip.  The Gaussian-mixture variants of the reference file are never selected by any preset and are not
provided."""
import ctypes

import torch

from .. import _lib as L


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _DMoLFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_hat, y, num_classes, log_scale_min):
        lib = L.lib()
        B, C, T = y_hat.shape
        yh = y_hat.contiguous().float()
        yt = y.reshape(B, T).contiguous().float()
        nll = torch.empty(B, T, dtype=torch.float32, device=y_hat.device)
        dyh = torch.empty_like(yh)
        L.check(lib.wae_dmol_loss_fwd(L.ptr(yh), L.ptr(yt), L.ptr(nll), L.ptr(dyh), B, C // 3, T, int(num_classes),
                                      float(log_scale_min), 0, _stream(yh)), "dmol_loss")
        ctx.save_for_backward(dyh)
        return nll

    @staticmethod
    def backward(ctx, dnll):
        (dyh,) = ctx.saved_tensors
        return dyh * dnll.unsqueeze(1), None, None, None


def discretized_mix_logistic_loss(y_hat, y, num_classes=256, log_scale_min=-7.0, reduce=True):
    """y_hat (B, 3*M, T), y (B, T, 1) in [-1, 1] -> scalar sum (reduce) or (B, T, 1) losses  (mixture.py:26-106)."""
    assert y_hat.dim() == 3 and y_hat.size(1) % 3 == 0
    if not y_hat.is_cuda:
        raise L.WaeError("discretized_mix_logistic_loss has no CPU implementation here: pass ROCm tensors")
    nll = _DMoLFn.apply(y_hat, y, num_classes, log_scale_min)
    return nll.sum() if reduce else nll.unsqueeze(-1)


def sample_from_discretized_mix_logistic(y, log_scale_min=-7.0, clamp_log_scale=False):
    """y (B, 3*M, T) -> samples (B, T) in [-1, 1]; uniforms drawn with torch's device RNG (mixture.py:118-156)."""
    assert y.size(1) % 3 == 0
    if not y.is_cuda:
        raise L.WaeError("sample_from_discretized_mix_logistic has no CPU implementation here: pass ROCm tensors")
    B, C, T = y.shape
    M = C // 3
    u_mix = torch.empty(B, T, M, device=y.device).uniform_(1e-5, 1.0 - 1e-5)
    u_log = torch.empty(B, T, device=y.device).uniform_(1e-5, 1.0 - 1e-5)
    yc = y.contiguous().float()
    out = torch.empty(B, T, dtype=torch.float32, device=y.device)
    L.check(L.lib().wae_dmol_sample(L.ptr(yc), L.ptr(u_mix), L.ptr(u_log), L.ptr(out), B, M, T, float(log_scale_min),
                                    int(bool(clamp_log_scale)), _stream(yc)), "dmol_sample")
    torch.cuda.current_stream(y.device).synchronize()      # inputs are locals: keep them alive until the kernel ran
    return out
