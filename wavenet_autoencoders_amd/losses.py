"""Masked training losses with the reference's calling convention (vqwae_train.py:324-401)."""
import ctypes

import torch
from torch import nn

from . import _lib as L
from .wavenet_vocoder.mixture import discretized_mix_logistic_loss


def sequence_mask(sequence_length, max_len=None):
    """(B, max_len) float mask of valid positions (vqwae_train.py:324-334); pure index arithmetic."""
    if max_len is None:
        max_len = int(sequence_length.max())
    r = torch.arange(0, max_len, device=sequence_length.device).unsqueeze(0)
    return (r < sequence_length.unsqueeze(1)).float()


def _stream(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


class _MaskedMeanFn(torch.autograd.Function):
    """sum(losses * mask) / sum(mask) (vqwae_train.py:379) by wae_weighted_mean; d/dlosses = mask / sum(mask)."""

    @staticmethod
    def forward(ctx, losses, mask):
        lf, mf = losses.contiguous().float(), mask.contiguous().float()
        out = torch.empty(2, dtype=torch.float32, device=lf.device)
        L.check(L.lib().wae_weighted_mean(L.ptr(lf), L.ptr(mf), lf.numel(), L.ptr(out), _stream(lf)), "weighted_mean")
        ctx.save_for_backward(mf, out)
        ctx.shape = losses.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        mf, out = ctx.saved_tensors
        return (mf * (g / out[1])).view(ctx.shape), None


_ERR_WORDS = {}


def _dev_key(device):
    device = torch.device(device)
    return (device.type, device.index if device.index is not None else torch.cuda.current_device())


def _err_word(device):
    key = _dev_key(device)
    if key not in _ERR_WORDS:
        _ERR_WORDS[key] = dict(word=torch.zeros(1, dtype=torch.int32, device=device), pending=None,
                               host=torch.zeros(1, dtype=torch.int32).pin_memory())
    return _ERR_WORDS[key]


def _arm(err, stream):
    """Behind a launch that may set the sticky word: an asynchronous copy of it to pinned host memory, then an event.  Once the
    event has fired the HOST copy is current -- reading it needs no device synchronisation (an `.item()` on the device word would
    block the host until everything queued on that stream, the model forward included, has finished)."""
    err["host"].copy_(err["word"], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(stream)
    err["pending"] = ev


def _raise_if_flagged(err, sync):
    """Look at the host copy of the sticky word once the launch that may have set it is known to be complete (sync=False: only if
    its event has already fired -- no host wait on the hot path)."""
    ev = err["pending"]
    if ev is None or (not sync and not ev.query()):
        return
    if sync:
        ev.synchronize()
    err["pending"] = None
    if int(err["host"][0]):
        err["word"].zero_()
        err["host"].zero_()
        raise IndexError("Target out of bounds")              # what nn.CrossEntropyLoss raises


def check_target_errors(device=None):
    """Raise the IndexError of an earlier MaskedCrossEntropyLoss call whose targets left [0, C) (synchronises with that call).
    Contract: nn.CrossEntropyLoss raises on the spot; here the kernel clamps the target, computes on, and the error surfaces at the
    next loss call or backward on that device, or when this function is called (call it before trusting the last loss of a run)."""
    want = None if device is None else _dev_key(device)
    for key, err in list(_ERR_WORDS.items()):
        if want is None or key == want:
            _raise_if_flagged(err, sync=True)


class _CELogitsFn(torch.autograd.Function):
    """per-position cross-entropy of (B, C, T) logits by wae_ce_logits_fwd / _bwd"""

    @staticmethod
    def forward(ctx, logits, target):
        lg = logits.contiguous().float()
        tg = target.contiguous().to(torch.int64)
        B, C, T = lg.shape
        nll = torch.empty(B, T, dtype=torch.float32, device=lg.device)
        lse = torch.empty(B, T, dtype=torch.float32, device=lg.device)
        # an out-of-range target: the kernel clamps it and sets a sticky per-device word; the IndexError nn.CrossEntropyLoss raises on
        # the spot is raised by check_target_errors() -- at the latest by the NEXT loss call on that device (reading the word here
        # every time was one host synchronisation per loss evaluation)
        err = _err_word(lg.device)
        _raise_if_flagged(err, sync=False)
        L.check(L.lib().wae_ce_logits_fwd(L.ptr(lg), L.ptr(tg), L.ptr(nll), L.ptr(lse), B, C, T, L.ptr(err["word"]), _stream(lg)), "ce_logits")
        _arm(err, torch.cuda.current_stream(lg.device))
        ctx.save_for_backward(lg, tg, lse)
        return nll

    @staticmethod
    def backward(ctx, dnll):
        lg, tg, lse = ctx.saved_tensors
        _raise_if_flagged(_err_word(lg.device), sync=False)      # a gradient of clamped targets is not handed on silently if known by now
        B, C, T = lg.shape
        w = dnll.contiguous().float()
        dl = torch.empty_like(lg)
        L.check(L.lib().wae_ce_logits_bwd(L.ptr(lg), L.ptr(tg), L.ptr(lse), L.ptr(w), L.ptr(dl), B, C, T, _stream(lg)), "ce_logits_bwd")
        return dl, None


class MaskedCrossEntropyLoss(nn.Module):
    """forward(input (B, C, T, 1), target (B, T, 1), lengths=None, mask=None, max_len=None) -> sum(CE*mask)/sum(mask)
    (vqwae_train.py:363-379) on explicit logits: wae_ce_logits_fwd / _bwd and wae_weighted_mean.  (WaeEngine.train_step uses the
    CE fused into the head kernel instead, which never materialises the logits.)"""

    def forward(self, input, target, lengths=None, mask=None, max_len=None):
        if lengths is None and mask is None:
            raise RuntimeError("Should provide either lengths or mask")
        if not input.is_cuda:
            raise L.WaeError("MaskedCrossEntropyLoss has no CPU implementation here: pass ROCm tensors")
        if mask is None:
            mask = sequence_mask(lengths, max_len).unsqueeze(-1)
        mask_ = mask.expand_as(target)
        B, C = input.shape[0], input.shape[1]
        losses = _CELogitsFn.apply(input.reshape(B, C, -1), target.reshape(B, -1))
        return _MaskedMeanFn.apply(losses.view(target.shape), mask_)


class DiscretizedMixturelogisticLoss(nn.Module):
    """vqwae_train.py:382-401; num_classes / log_scale_min come from the constructor instead of a module-global."""

    def __init__(self, num_classes=256, log_scale_min=-7.0):
        super().__init__()
        self.num_classes, self.log_scale_min = num_classes, log_scale_min

    def forward(self, input, target, lengths=None, mask=None, max_len=None):
        if lengths is None and mask is None:
            raise RuntimeError("Should provide either lengths or mask")
        if mask is None:
            mask = sequence_mask(lengths, max_len).unsqueeze(-1)
        mask_ = mask.expand_as(target)
        losses = discretized_mix_logistic_loss(input, target, num_classes=self.num_classes, log_scale_min=self.log_scale_min,
                                               reduce=False)
        assert losses.size() == target.size()
        return _MaskedMeanFn.apply(losses, mask_)


class ExponentialMovingAverage(object):
    """shadow -= (1 - decay) * (shadow - x) (vqwae_train.py:339-350); WaeEngine.train_step fuses this into the optimizer."""

    def __init__(self, decay):
        self.decay, self.shadow = decay, {}

    def register(self, name, val):
        self.shadow[name] = val.clone()

    def update(self, name, x):
        assert name in self.shadow
        self.shadow[name] -= (1.0 - self.decay) * (self.shadow[name] - x)


class _SoftmaxBCT(torch.autograd.Function):
    """F.softmax(x, dim=1) on (B, C, T) logits (WaveNet.forward(softmax=True), wavenet.py:214; VQVAE.forward, vqvae_model.py:79-80) through
    wae_softmax_bct_fwd / _bwd: the drop-in modules form no arithmetic with torch operators."""

    @staticmethod
    def forward(ctx, x):
        from . import _lib as L
        xf = x.contiguous().float()
        B, C, T = xf.shape
        p = torch.empty_like(xf)
        L.check(L.lib().wae_softmax_bct_fwd(L.ptr(xf), L.ptr(p), B, C, T, _stream(xf)), "softmax_bct_fwd")
        ctx.save_for_backward(p)
        return p

    @staticmethod
    def backward(ctx, dp):
        from . import _lib as L
        (p,) = ctx.saved_tensors
        B, C, T = p.shape
        dpf = dp.contiguous().float()
        dx = torch.empty_like(p)
        L.check(L.lib().wae_softmax_bct_bwd(L.ptr(p), L.ptr(dpf), L.ptr(dx), B, C, T, _stream(p)), "softmax_bct_bwd")
        return dx


def softmax_bct(x: torch.Tensor) -> torch.Tensor:
    """softmax over dim 1 of (B, C, T) logits on the HIP path (differentiable)."""
    return _SoftmaxBCT.apply(x)
