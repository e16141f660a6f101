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


class MaskedCrossEntropyLoss(nn.Module):
    """forward(input (B, C, T, 1), target (B, T, 1), lengths=None, mask=None, max_len=None) -> sum(CE*mask)/sum(mask)
    (vqwae_train.py:363-379).  The CE itself is torch's device op on the logits the engine produced; the fused
    head+CE kernel (WaeEngine.train_step) is the fast path that never materialises the logits."""

    def forward(self, input, target, lengths=None, mask=None, max_len=None):
        if lengths is None and mask is None:
            raise RuntimeError("Should provide either lengths or mask")
        if mask is None:
            mask = sequence_mask(lengths, max_len).unsqueeze(-1)
        mask_ = mask.expand_as(target)
        losses = torch.nn.functional.cross_entropy(input, target, reduction="none")
        return (losses * mask_).sum() / mask_.sum()


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
        return (losses * mask_).sum() / mask_.sum()


class ExponentialMovingAverage(object):
    """shadow -= (1 - decay) * (shadow - x) (vqwae_train.py:339-350); WaeEngine.train_step fuses this into the optimizer."""

    def __init__(self, decay):
        self.decay, self.shadow = decay, {}

    def register(self, name, val):
        self.shadow[name] = val.clone()

    def update(self, name, x):
        assert name in self.shadow
        self.shadow[name] -= (1.0 - self.decay) * (self.shadow[name] - x)
