"""Drop-in ``VectorQuantize`` (reference: vector_quantization.py:10-49) on the MI355X kernels.

The sliced / EMA quantizers of the reference file are defined there but never instantiated; they
are not part of the hot path and are not provided."""
import ctypes

import torch
from torch import nn

from . import _lib as L


class _VQFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, lat, emb, beta):
        lib = L.lib()
        B, D, Tq = lat.shape
        K = emb.shape[0]
        lat = lat.contiguous().float()
        embc = emb.contiguous().float()
        idx = torch.empty(B * Tq, dtype=torch.int64, device=lat.device)
        quant = torch.empty_like(lat)
        stats = torch.empty(2, dtype=torch.float32, device=lat.device)
        hist = torch.empty(K + 1, dtype=torch.int32, device=lat.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(lat.device).cuda_stream)
        L.check(lib.wae_vq_nearest(L.ptr(lat), L.ptr(embc), L.ptr(idx), L.ptr(quant), L.ptr(stats), L.ptr(hist), B, D, Tq, K,
                                   float(beta), st), "vq_nearest")
        ctx.save_for_backward(lat, quant, idx)
        ctx.beta, ctx.K = float(beta), K
        ctx.mark_non_differentiable(idx)
        return quant, stats[0], stats[1], idx

    @staticmethod
    def backward(ctx, dquant, dloss, dperp, _):
        lib = L.lib()
        lat, quant, idx = ctx.saved_tensors
        B, D, Tq = lat.shape
        dlat = torch.empty_like(lat)
        demb = torch.zeros(ctx.K, D, dtype=torch.float32, device=lat.device)
        dq = dquant.contiguous().float() if dquant is not None else None
        scale = float(dloss) if dloss is not None else 0.0
        st = ctypes.c_void_p(torch.cuda.current_stream(lat.device).cuda_stream)
        L.check(lib.wae_vq_bwd(L.ptr(lat), L.ptr(quant), L.ptr(idx), L.ptr(dq), L.ptr(dlat), L.ptr(demb), B, D, Tq, ctx.beta, scale,
                               st), "vq_bwd")
        return dlat, demb, None


class VectorQuantize(nn.Module):
    """forward(inputs (B, D, T)) -> (quant (B, D, T) with straight-through gradient, vq_loss, perplexity)."""

    def __init__(self, K, D, beta=0.25):
        super().__init__()
        self.K, self.D, self.beta = K, D, beta
        self.embedding = nn.Embedding(K, D)
        self.embedding.weight.data.uniform_(-1.0 / K, 1.0 / K)          # vector_quantization.py:16

    def forward(self, inputs):
        if not inputs.is_cuda:
            raise L.WaeError("VectorQuantize has no CPU implementation: move inputs and module to a ROCm GPU")
        quant, loss, perp, idx = _VQFn.apply(inputs, self.embedding.weight, self.beta)
        self.last_indices = idx
        return quant, loss, perp
