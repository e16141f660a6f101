"""Shared plumbing of the drop-in modules: parameters with the reference's names live as views of the engine's flat
arena, so ``state_dict()`` / ``load_state_dict()`` / ``torch.optim`` see ordinary tensors while every kernel reads one
contiguous buffer."""
import math

import torch
from torch import nn

from .. import packing as P


class _Holder(nn.Module):
    """Pure parameter container (a node of the reference's module tree, e.g. ``conv_layers.3.conv``)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: call the owning model")


def _leaf(root, dotted):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Holder())
        m = m._modules[p]
    return m, parts[-1]


def register_params(root: nn.Module, geom: P.Geometry, strip: str = "", skip=None):
    """Create nn.Parameters for every entry of packing.param_specs (names relative to ``strip``) with the reference's
    initialisation: kaiming-normal conv weights under weight norm (g = ||v||), zero biases (modules.py:13-19),
    N(0, 0.1) speaker embedding (wavenet.py:143-147, modules.py:21-24), 1/(2s+1) smoothing FIRs (upsample.py:42-44),
    default Conv1d/Linear init for conv_in and the encoder, U(-1/K, 1/K) codebook (vector_quantization.py:16)."""
    names = []
    for name, shape, is_v in P.param_specs(geom):
        if not name.startswith(strip):
            continue
        rel = name[len(strip):]
        if skip is not None and skip(rel):
            continue                # no such parameter in this module: its arena slot stays at zero and is never trained
        t = torch.zeros(shape)
        if rel.endswith("weight_v"):
            if "upsample" in rel:
                t.fill_(1.0 / shape[-1])
            else:
                nn.init.kaiming_normal_(t, nonlinearity="relu")
        elif rel.endswith("embed_speakers.weight"):
            t.normal_(0, 0.1)
        elif rel.endswith("embedding.weight"):
            t.uniform_(-1.0 / shape[0], 1.0 / shape[0])
        elif rel.endswith(".weight"):
            fan_in = int(torch.tensor(shape[1:]).prod())
            bound = 1.0 / math.sqrt(fan_in)
            t.uniform_(-bound, bound)
        elif rel.endswith(".bias") and ("encoder" in name):
            wshape = dict((n, s) for n, s, _ in P.param_specs(geom))[name[:-4] + "weight"]
            bound = 1.0 / math.sqrt(int(torch.tensor(wshape[1:]).prod()))
            t.uniform_(-bound, bound)
        m, leaf = _leaf(root, rel)
        m.register_parameter(leaf, nn.Parameter(t))
        names.append(rel)
    # weight_g = ||v|| per output row (nn.utils.weight_norm initialisation)
    sd = dict(root.named_parameters())
    for rel in names:
        if rel.endswith("weight_g"):
            v = sd[rel[:-1] + "v"].data
            sd[rel].data.copy_(v.reshape(v.shape[0], -1).norm(dim=1).view(sd[rel].shape))
    return names


class ArenaModel(nn.Module):
    """Base of WaveNet / VQVAE: owns a WaeEngine once the parameters sit on a GPU."""

    def _init_arena(self, geom: P.Geometry, strip: str, dtype="fp32", skip=None):
        self.geom = geom
        self._strip = strip
        self._compute_dtype = dtype
        self._engine = None
        self._pnames = register_params(self, geom, strip, skip)

    def set_compute_dtype(self, dtype: str):
        """'fp32' (exact, default), 'bf16' or 'fp16' (16-bit storage of activations and packed weights, fp32 accumulate)."""
        if dtype != self._compute_dtype:
            if self._engine is not None:
                sd = {k: v.detach().clone() for k, v in self.state_dict().items()}
                self._engine = None
                for k, p in self.named_parameters():
                    p.data = sd[k]
            self._compute_dtype = dtype
        return self

    def engine(self):
        p0 = next(self.parameters())
        if not p0.is_cuda:
            from .._lib import WaeError
            raise WaeError("this model has no CPU implementation: move it to a ROCm GPU first (model.cuda())")
        if self._engine is None or self._engine.device != p0.device:
            from ..engine import WaeEngine
            eng = WaeEngine(self.geom, dtype=self._compute_dtype, device=str(p0.device), dropout=float(getattr(self, "dropout", 0.0)))
            params = dict(self.named_parameters())
            for rel in self._pnames:
                full = self._strip + rel
                off, n = eng.lay.off(full), eng.lay.numel(full)
                view = eng.params[off:off + n].view(eng.lay.shapes[full])
                view.copy_(params[rel].data)
                params[rel].data = view                      # the Parameter now aliases the arena
            eng.weights_dirty = True
            self._engine = eng
        self._engine.weights_dirty = True                    # optimizers write through the aliases
        return self._engine

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._engine = None                                  # storage moved: rebind lazily
        return out

    def _grad_views(self, eng):
        params = dict(self.named_parameters())
        out = []
        for rel in self._pnames:
            full = self._strip + rel
            off, n = eng.lay.off(full), eng.lay.numel(full)
            out.append(eng.grads[off:off + n].view(eng.lay.shapes[full]))
        return [params[r] for r in self._pnames], out
