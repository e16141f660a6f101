"""Drop-in quantizers of the reference's vector_quantization.py on the MI355X kernels: ``VectorQuantize`` (:10-49, the one
the hot path uses) and -- SURVEY section 8(f) rank 3 -- ``SlicedVectorQuantize`` (:51-128), ``SlicedVectorQuantizeEMA``
(:132-235) and ``VectorQuantizeEMA`` (:239-306), same constructor arguments, parameter/buffer names and return values.
No CPU implementation: every forward raises on a non-GPU tensor."""
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


class _SlicedFn(torch.autograd.Function):
    """One launch group per channel slice [d0, d0+D) with its own codebook; with EMA buffers and training=True the
    codebook is updated between the search and the gather (vector_quantization.py:190-218, :275-292)."""

    @staticmethod
    def forward(ctx, lat, spec, *embs):
        lib = L.lib()
        B, Dtot, Tq = lat.shape
        lat = lat.contiguous().float()
        dev = lat.device
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        quant = torch.empty_like(lat)
        stats = torch.empty(len(embs), 2, dtype=torch.float32, device=dev)
        idxs, d0 = [], 0
        for i, emb in enumerate(embs):
            K, D = emb.shape
            assert emb.is_contiguous() and emb.dtype == torch.float32
            idx = torch.empty(B * Tq, dtype=torch.int64, device=dev)
            hist = torch.empty(K + 1, dtype=torch.int32, device=dev)
            c_loss = spec["c_loss"] * D / Dtot
            args = (L.ptr(lat), L.ptr(emb), L.ptr(idx), L.ptr(quant), L.ptr(stats[i]), L.ptr(hist), B, Dtot, d0, D, Tq, K, c_loss)
            if spec["ema"] is not None and spec["training"]:
                n, w = spec["ema"][i]
                L.check(lib.wae_vq_slice(*args, 1, st), "vq_slice")
                L.check(lib.wae_vq_ema_update(L.ptr(lat), L.ptr(idx), L.ptr(hist), L.ptr(n), L.ptr(w), L.ptr(emb), B, Dtot, d0, D,
                                              Tq, K, float(spec["decay"]), st), "vq_ema_update")
                L.check(lib.wae_vq_slice(*args, 2, st), "vq_slice")
            else:
                L.check(lib.wae_vq_slice(*args, 0, st), "vq_slice")
            idxs.append(idx)
            d0 += D
        assert d0 == Dtot, "codebook widths do not add up to the latent width"
        ctx.save_for_backward(lat, quant, *idxs)
        ctx.spec, ctx.shapes = spec, [tuple(e.shape) for e in embs]
        ctx.mark_non_differentiable(*idxs)
        return (quant, stats[:, 0].sum(), stats[:, 1].sum(), *idxs)

    @staticmethod
    def backward(ctx, dquant, dloss, dperp, *_):
        lib = L.lib()
        lat, quant, *idxs = ctx.saved_tensors
        B, Dtot, Tq = lat.shape
        spec = ctx.spec
        st = ctypes.c_void_p(torch.cuda.current_stream(lat.device).cuda_stream)
        dlat = torch.empty_like(lat)
        dq = dquant.contiguous().float() if dquant is not None else None
        scale = (float(dloss) if dloss is not None else 0.0) * 2.0 / (B * Dtot * Tq)
        dembs, d0 = [], 0
        for idx, (K, D) in zip(idxs, ctx.shapes):
            demb = torch.zeros(K, D, dtype=torch.float32, device=lat.device) if spec["c_emb"] is not None else None
            L.check(lib.wae_vq_slice_bwd(L.ptr(lat), L.ptr(quant), L.ptr(idx), L.ptr(dq), L.ptr(dlat), L.ptr(demb), B, Dtot, d0, D,
                                         Tq, scale * spec["c_lat"], scale * (spec["c_emb"] or 0.0), st), "vq_slice_bwd")
            dembs.append(demb)
            d0 += D
        return (dlat, None, *dembs)


def _need_gpu(x, who):
    if not x.is_cuda:
        raise L.WaeError(f"{who} has no CPU implementation: move inputs and module to a ROCm GPU")


class SlicedVectorQuantize(nn.Module):
    """vector_quantization.py:51-128.  Two halves of the channel axis, K and K1 codes.  The loss wiring is the reference's:
    weight 1 on the encoder-side term, beta on the codebook-side term (:113-118); perp = perp1 + perp2 (:125-127).
    ``dropout`` / ``dropout_rate`` / ``decay`` are stored and unused, as in the reference."""

    def __init__(self, K, D, beta=0.25, decay=0.99, n_d=2, dropout=False, dropout_rate=0.25, K1=None):
        super().__init__()
        self.K, self.K1, self.D, self.sub_D = K, (K1 if K1 is not None else K), D, D // n_d
        self.embedding1 = nn.Embedding(K, self.sub_D)
        self.embedding1.weight.data.uniform_(-1.0 / K, 1.0 / K)
        self.embedding2 = nn.Embedding(self.K1, self.sub_D)
        self.embedding2.weight.data.uniform_(-1.0 / self.K1, 1.0 / self.K1)
        self.decay, self.beta, self.dropout, self.dropout_rate = decay, beta, dropout, dropout_rate

    def forward(self, x):
        _need_gpu(x, "SlicedVectorQuantize")
        assert x.size(1) == self.D == 2 * self.sub_D
        spec = dict(c_loss=1.0 + self.beta, c_lat=1.0, c_emb=self.beta, ema=None, training=self.training, decay=self.decay)
        quant, loss, perp, i1, i2 = _SlicedFn.apply(x, spec, self.embedding1.weight, self.embedding2.weight)
        self.last_indices = (i1, i2)
        return quant, loss, perp


class SlicedVectorQuantizeEMA(nn.Module):
    """vector_quantization.py:132-235: two slices, codebooks moved by exponential moving averages while training;
    vq_loss = beta * mse(sg[q], x) (:220), no codebook gradient."""

    def __init__(self, K, D, beta=0.25, decay=0.99, n_d=2):
        super().__init__()
        self.K, self.D, self.sub_D = K, D, D // n_d
        self.embedding1 = nn.Embedding(K, self.sub_D)
        self.embedding1.weight.data.uniform_(-1.0 / K, 1.0 / K)
        self.embedding2 = nn.Embedding(K, self.sub_D)
        self.embedding2.weight.data.uniform_(-1.0 / K, 1.0 / K)
        self.register_buffer("ema_cluster_size1", torch.zeros(K))
        self.register_buffer("ema_w1", torch.zeros(K, self.sub_D))
        self.register_buffer("ema_cluster_size2", torch.zeros(K))
        self.register_buffer("ema_w2", torch.zeros(K, self.sub_D))
        self.decay, self.beta = decay, beta

    def forward(self, x):
        _need_gpu(x, "SlicedVectorQuantizeEMA")
        assert x.size(1) == self.D == 2 * self.sub_D
        ema = [(self.ema_cluster_size1, self.ema_w1), (self.ema_cluster_size2, self.ema_w2)]
        spec = dict(c_loss=self.beta, c_lat=self.beta, c_emb=None, ema=ema, training=self.training, decay=self.decay)
        quant, loss, perp, i1, i2 = _SlicedFn.apply(x, spec, self.embedding1.weight.data, self.embedding2.weight.data)
        self.last_indices = (i1, i2)
        return quant, loss, perp


class VectorQuantizeEMA(nn.Module):
    """vector_quantization.py:239-306: one codebook over the full width, EMA updates while training."""

    def __init__(self, K, D, beta=0.25, decay=0.99):
        super().__init__()
        self.K, self.D = K, D
        self.embedding = nn.Embedding(K, D)
        self.embedding.weight.data.uniform_(-1.0 / K, 1.0 / K)
        self.register_buffer("ema_cluster_size", torch.zeros(K))
        self.register_buffer("ema_w", torch.zeros(K, D))
        self.decay, self.beta = decay, beta

    def forward(self, x):
        _need_gpu(x, "VectorQuantizeEMA")
        assert x.size(1) == self.D
        spec = dict(c_loss=self.beta, c_lat=self.beta, c_emb=None, ema=[(self.ema_cluster_size, self.ema_w)],
                    training=self.training, decay=self.decay)
        quant, loss, perp, idx = _SlicedFn.apply(x, spec, self.embedding.weight.data)
        self.last_indices = idx
        return quant, loss, perp
