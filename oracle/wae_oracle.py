"""CPU oracle for the WaveNet-autoencoder hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a *functional* restatement (plain ``torch`` CPU ops over a flat
``state_dict``) of the reference's encoder -> VQ -> upsample -> gated dilated
stack -> head -> loss path.  It is NOT part of the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline.  The product path
(``wavenet_autoencoders_amd``) never imports anything from ``oracle/``.

Pinning: the reference ships no tests for this path, so
the oracle is pinned against outputs of the reference itself: the generator
``tests/golden/make_golden.py`` imports ``/root/reference`` in the build
container, runs the reference modules on closed-form weights/inputs and commits
the resulting vectors under ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against them.  mu-law companding has no importable
reference here (nnmnkwii absent) and is pinned by hand-computed values only.

Every function cites the reference file:line it follows.
State-dict key names are the reference's.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# weight norm (reference: nn.utils.weight_norm applied in
# wavenet_vocoder/modules.py:18 and wavenet_vocoder/upsample.py:44)
# --------------------------------------------------------------------------
def weight_norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """w = g * v / ||v||, norm over every dim but 0 (old-style weight_g/weight_v)."""
    dims = tuple(range(1, v.dim()))
    return v * (g / v.pow(2).sum(dims, keepdim=True).sqrt())


def eff_weight(sd: SD, prefix: str) -> torch.Tensor:
    """Effective conv weight for ``prefix`` whether or not weight-norm was removed
    (wavenet.py:358-364 make_generation_fast_ leaves a plain ``.weight``)."""
    if prefix + ".weight_v" in sd:
        return weight_norm(sd[prefix + ".weight_v"], sd[prefix + ".weight_g"])
    return sd[prefix + ".weight"]


# --------------------------------------------------------------------------
# a1: encoder  (vqvae_model.py:9-23 ConvReLURes, :27-51 Encoder)
# --------------------------------------------------------------------------
ENCODER_BLOCKS = [(3, 1), (3, 1), (5, 2), (5, 2), (3, 1), (3, 1), (1, 1), (1, 1), (1, 1), (1, 1)]


def encoder_forward(sd: SD, c: torch.Tensor, prefix: str = "encoder.") -> torch.Tensor:
    """c (B, c_in, F) -> latents (B, Cc, F').  y = relu(conv(x)); residual add only
    when stride == 1 and Cin == Cout (vqvae_model.py:17-23)."""
    x = c
    for i, (k, s) in enumerate(ENCODER_BLOCKS):
        w = sd[f"{prefix}net.{i}.conv.weight"]
        b = sd[f"{prefix}net.{i}.conv.bias"]
        y = F.relu(F.conv1d(x, w, b, stride=s, padding=k // 2))
        if s == 1 and w.shape[0] == w.shape[1]:
            y = y + x
        x = y
    # Linear over the channel dim (vqvae_model.py:50)
    lat = F.linear(x.permute(0, 2, 1), sd[prefix + "lin.weight"], sd[prefix + "lin.bias"])
    return lat.permute(0, 2, 1).contiguous()


# --------------------------------------------------------------------------
# a2: vector quantizer (vector_quantization.py:21-49)
# --------------------------------------------------------------------------
def vq_forward(emb: torch.Tensor, lat: torch.Tensor, beta: float = 0.25):
    """lat (B, D, T') -> (quant (B,D,T') with straight-through grad, vq_loss, perp, idx (B*T',)).

    distance = ||e||^2 + ||x||^2 - 2 x.e via addmm (:27-30), first-minimum argmin (:31),
    vq_loss = beta * mse(sg[q], x) + mse(q, sg[x]) (:41-43), perplexity over the batch (:47-48).
    """
    x = lat.permute(0, 2, 1).contiguous()
    B, T, D = x.shape
    flat = x.view(-1, D)
    in_sqr = (flat ** 2).sum(1, keepdim=True)
    e_sqr = (emb ** 2).sum(1)
    dis = torch.addmm(e_sqr + in_sqr, flat, emb.t(), alpha=-2.0, beta=1.0)
    idx = torch.argmin(dis, dim=1)
    q = emb[idx].view(B, T, D)
    vq = ((q.detach() - x) ** 2).mean()
    commit = ((q - x.detach()) ** 2).mean()
    vq_loss = beta * vq + commit
    q_st = x + (q - x).detach()
    K = emb.shape[0]
    avg = torch.bincount(idx, minlength=K).to(x.dtype) / idx.numel()
    perp = torch.exp(-(avg * torch.log(avg + 1e-10)).sum())
    return q_st.permute(0, 2, 1).contiguous(), vq_loss, perp, idx


def _vq_assign(emb: torch.Tensor, flat: torch.Tensor):
    """The per-slice search shared by every quantizer class (vector_quantization.py:84-97, :163-176, :262-270):
    argmax(-dis) == first minimum of ||e||^2 + ||x||^2 - 2 x.e."""
    dis = torch.addmm((emb ** 2).sum(1) + (flat ** 2).sum(1, keepdim=True), flat, emb.t(), alpha=-2.0, beta=1.0)
    return torch.argmax(-1.0 * dis, dim=1)


def _perplexity(idx: torch.Tensor, K: int, dtype) -> torch.Tensor:
    avg = torch.bincount(idx, minlength=K).to(dtype) / idx.numel()
    return torch.exp(-(avg * torch.log(avg + 1e-10)).sum())


def sliced_vq_forward(emb1: torch.Tensor, emb2: torch.Tensor, lat: torch.Tensor, beta: float = 0.25):
    """SlicedVectorQuantize.forward (vector_quantization.py:75-128): two channel halves, one codebook each (K and K1 codes).
    NB the loss wiring differs from VectorQuantize: the encoder-side term has weight 1 and the codebook-side term
    weight beta (:113-118); perplexity is the SUM of the two slice perplexities (:125-127).
    -> (quant (B,D,T) straight-through, vq_loss, perp, (idx1, idx2))."""
    x = lat.permute(0, 2, 1).contiguous()
    B, T, D = x.shape
    sub = emb1.shape[1]
    flat = x.view(-1, D)
    i1, i2 = _vq_assign(emb1, flat[:, :sub]), _vq_assign(emb2, flat[:, sub:])
    q = torch.cat([emb1[i1].view(B, T, sub), emb2[i2].view(B, T, D - sub)], dim=2)
    vq_loss = ((q.detach() - x) ** 2).mean() + beta * ((q - x.detach()) ** 2).mean()
    q_st = x + (q - x).detach()
    perp = _perplexity(i1, emb1.shape[0], x.dtype) + _perplexity(i2, emb2.shape[0], x.dtype)
    return q_st.permute(0, 2, 1), vq_loss, perp, (i1, i2)


def _ema_update(emb, ema_n, ema_w, flat, idx, decay):
    """Training branch of the EMA classes (vector_quantization.py:190-215, :275-290); returns the new (emb, ema_n, ema_w)."""
    K = emb.shape[0]
    counts = torch.bincount(idx, minlength=K).to(flat.dtype)
    ema_n = ema_n * decay + (1.0 - decay) * counts
    n = ema_n.sum()
    ema_n = (ema_n + 1e-5) / (n + K * 1e-5) * n
    dw = torch.zeros_like(ema_w).index_add_(0, idx, flat)
    ema_w = ema_w * decay + (1 - decay) * dw
    return ema_w / ema_n.unsqueeze(1), ema_n, ema_w


def vq_ema_forward(state: dict, lat: torch.Tensor, beta: float = 0.25, decay: float = 0.99, training: bool = True):
    """VectorQuantizeEMA.forward (vector_quantization.py:255-306).  state: embedding (K,D), ema_cluster_size (K), ema_w (K,D);
    returns (quant, vq_loss, perp, idx, new_state).  The codes are gathered from the UPDATED codebook (:290-292) with the
    assignments made against the old one; vq_loss = beta * mse(sg[q], x) only (:294)."""
    x = lat.permute(0, 2, 1).contiguous()
    B, T, D = x.shape
    flat = x.view(-1, D)
    emb = state["embedding"]
    idx = _vq_assign(emb, flat)
    new = dict(state)
    if training:
        emb, n, w = _ema_update(emb, state["ema_cluster_size"], state["ema_w"], flat.detach(), idx, decay)
        new.update(embedding=emb, ema_cluster_size=n, ema_w=w)
    q = emb[idx].view(B, T, D)
    vq_loss = beta * ((q.detach() - x) ** 2).mean()
    q_st = x + (q - x).detach()
    return q_st.permute(0, 2, 1), vq_loss, _perplexity(idx, emb.shape[0], x.dtype), idx, new


def sliced_vq_ema_forward(state: dict, lat: torch.Tensor, beta: float = 0.25, decay: float = 0.99, training: bool = True):
    """SlicedVectorQuantizeEMA.forward (vector_quantization.py:157-235).  state: embedding1/2, ema_cluster_size1/2, ema_w1/2."""
    x = lat.permute(0, 2, 1).contiguous()
    B, T, D = x.shape
    sub = state["embedding1"].shape[1]
    flat = x.view(-1, D)
    parts = (flat[:, :sub], flat[:, sub:])
    new, qs, idxs, perp = dict(state), [], [], 0.0
    for s, f in zip(("1", "2"), parts):
        emb = state["embedding" + s]
        idx = _vq_assign(emb, f)
        if training:
            emb, n, w = _ema_update(emb, state["ema_cluster_size" + s], state["ema_w" + s], f.detach(), idx, decay)
            new.update({"embedding" + s: emb, "ema_cluster_size" + s: n, "ema_w" + s: w})
        qs.append(emb[idx].view(B, T, -1))
        idxs.append(idx)
        perp = perp + _perplexity(idx, emb.shape[0], x.dtype)
    q = torch.cat(qs, dim=2)
    vq_loss = beta * ((q.detach() - x) ** 2).mean()
    q_st = x + (q - x).detach()
    return q_st.permute(0, 2, 1), vq_loss, perp, tuple(idxs), new


def vq_distances(emb: torch.Tensor, lat: torch.Tensor) -> torch.Tensor:
    """(N, K) distance matrix in the reference's formulation; used for top-2 margins."""
    flat = lat.permute(0, 2, 1).reshape(-1, lat.shape[1])
    return torch.addmm((emb ** 2).sum(1) + (flat ** 2).sum(1, keepdim=True), flat, emb.t(), alpha=-2.0)


# --------------------------------------------------------------------------
# a3: conditioning upsample net (upsample.py:12-21, :29-66, :69-85)
# --------------------------------------------------------------------------
def upsample_forward(sd: SD, c: torch.Tensor, scales: Sequence[int], prefix: str = "wavenet.upsample_net.",
                     cin_pad: int = 0, conv_in: bool = True, act: str = "none", act_slope: float = 0.01) -> torch.Tensor:
    """ConvInUpsampleNetwork (upsample.py:69-85): conv_in (plain Conv1d, k = 2*cin_pad+1, no bias, no weight-norm; :77-78) then per
    scale s: nearest stretch x s (:19-21) + one shared 1-channel FIR of 2s+1 taps, zero padded (:39-46).
    conv_in=False: the plain UpsampleNetwork (upsample.py:29-66): the stages alone (keys `up_layers.N`, no `.upsample`), then
    `indent = cin_pad * prod(scales)` samples trimmed at either end (:64-65).
    act != "none": upsample_activation (:44-46), an element-wise module behind every stage's FIR (ReLU, LeakyReLU(act_slope), Tanh,
    Sigmoid here); the ModuleList then holds three modules per stage, the FIRs at up_layers.{3 i + 1}."""
    x = F.conv1d(c, sd[prefix + "conv_in.weight"]) if conv_in else c
    x = x.unsqueeze(1)
    per = 2 if act == "none" else 3
    for i, s in enumerate(scales):
        x = F.interpolate(x, scale_factor=(1, s), mode="nearest")
        w = eff_weight(sd, f"{prefix}{'upsample.' if conv_in else ''}up_layers.{per * i + 1}")
        x = F.conv2d(x, w, padding=(0, s))
        if act == "ReLU":
            x = torch.relu(x)
        elif act == "LeakyReLU":
            x = F.leaky_relu(x, act_slope)
        elif act == "Tanh":
            x = torch.tanh(x)
        elif act == "Sigmoid":
            x = torch.sigmoid(x)
        else:
            assert act == "none", act
    x = x.squeeze(1)
    indent = 0 if conv_in else cin_pad * int(np.prod(list(scales)))
    return x[:, :, indent:x.shape[-1] - indent] if indent > 0 else x


# --------------------------------------------------------------------------
# a6: one gated residual layer (modules.py:115-163)
# --------------------------------------------------------------------------
def dropout_keep(seed: int, B: int, R: int, T: int, p: float) -> torch.Tensor:
    """The keep mask (B, R, T) of the engine's dropout (csrc/misc.hip: dropout_keep): a counter-based hash of (finalised seed, element
    index in the engine's (B, T, Rp) layout, Rp = R rounded up to 128); keep iff the top 24 bits of the mix >= p * 2^24.
    The reference draws its mask from torch's RNG (modules.py:127-128: F.dropout); the engine cannot reproduce that stream, so
    parity is checked mask-for-mask: the oracle applies THIS mask where the reference applies torch's."""
    Rp = (R + 127) // 128 * 128
    e = (np.arange(B * T, dtype=np.uint64)[:, None] * np.uint64(Rp) + np.arange(R, dtype=np.uint64)[None, :])
    M = 0xFFFFFFFFFFFFFFFF
    z = ((seed & M) + 0x9E3779B97F4A7C15) & M               # csrc/misc.hip: dropout_key -- the seed is finalised before it meets e
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    key = z ^ (z >> 31)
    with np.errstate(over="ignore"):
        h = (e ^ np.uint64(key)) * np.uint64(0x9E3779B97F4A7C15)
        h ^= h >> np.uint64(32)
        h *= np.uint64(0xD6E8FEB86659FD93)
        h ^= h >> np.uint64(32)
    keep = (h >> np.uint64(40)).astype(np.int64) >= int(p * 16777216.0 + 0.5)
    return torch.from_numpy(keep.reshape(B, T, R)).permute(0, 2, 1).contiguous()


def glu_layer_forward(sd: SD, prefix: str, x: torch.Tensor, c: Optional[torch.Tensor],
                      g: Optional[torch.Tensor], dilation: int, keep: Optional[torch.Tensor] = None, p: float = 0.0,
                      causal: bool = True):
    """x (B,R,T), c (B,Cc,T) | None, g (B,Cg,T) or (B,Cg,1) | None -> (x' (B,R,T), s (B,S,T)).
    causal=False (modules.py:82-88,134-136): the convolution pads (k-1)//2 * d on BOTH sides and nothing is trimmed (odd k).

    residual = x; x = F.dropout(x, p, training) (:126-128: `keep` (B,R,T) bool is the mask in training, None = eval);
    causal dilated conv with pad (k-1)*d, tail trimmed (:134-136); split a|b (:138); add 1x1(c),
    1x1(g) halves (:141-152); tanh(a)*sigmoid(b) (:154); skip and out 1x1 (:157-160);
    (out + residual) * sqrt(0.5) (:162).
    """
    T = x.shape[-1]
    w = eff_weight(sd, prefix + "conv")
    k = w.shape[-1]
    xc = x if keep is None else x * keep.to(x.dtype) / (1.0 - p)
    if causal:
        z = F.conv1d(xc, w, sd.get(prefix + "conv.bias"), padding=(k - 1) * dilation, dilation=dilation)[:, :, :T]
    else:
        z = F.conv1d(xc, w, sd.get(prefix + "conv.bias"), padding=(k - 1) // 2 * dilation, dilation=dilation)
    if c is not None:
        z = z + F.conv1d(c, eff_weight(sd, prefix + "conv1x1c"))
    if g is not None:
        z = z + F.conv1d(g, eff_weight(sd, prefix + "conv1x1g"))
    a, b = z.split(z.shape[1] // 2, dim=1)
    u = torch.tanh(a) * torch.sigmoid(b)
    s = F.conv1d(u, eff_weight(sd, prefix + "conv1x1_skip"), sd.get(prefix + "conv1x1_skip.bias"))
    o = F.conv1d(u, eff_weight(sd, prefix + "conv1x1_out"), sd.get(prefix + "conv1x1_out.bias"))
    return (o + x) * math.sqrt(0.5), s


def glu_layer_gate(sd: SD, prefix: str, x, c, g, dilation: int) -> torch.Tensor:
    """The gated activation u = tanh(a)*sigmoid(b) of one layer (modules.py:134-154), i.e. the tensor both 1x1
    output convs consume; the HIP path stores it per layer and contracts the skip 1x1 of all layers at once."""
    T = x.shape[-1]
    w = eff_weight(sd, prefix + "conv")
    k = w.shape[-1]
    z = F.conv1d(x, w, sd.get(prefix + "conv.bias"), padding=(k - 1) * dilation, dilation=dilation)[:, :, :T]
    if c is not None:
        z = z + F.conv1d(c, eff_weight(sd, prefix + "conv1x1c"))
    if g is not None:
        z = z + F.conv1d(g, eff_weight(sd, prefix + "conv1x1g"))
    a, b = z.split(z.shape[1] // 2, dim=1)
    return torch.tanh(a) * torch.sigmoid(b)


def layer_dilations(layers: int, stacks: int) -> List[int]:
    """wavenet.py:117,126: d = 2 ** (layer % (layers // stacks))."""
    assert layers % stacks == 0
    per = layers // stacks
    return [2 ** (i % per) for i in range(layers)]


def receptive_field_size(total_layers: int, num_cycles: int, kernel_size: int) -> int:
    """wavenet.py:42-60."""
    return (kernel_size - 1) * sum(layer_dilations(total_layers, num_cycles)) + 1


# --------------------------------------------------------------------------
# a4..a8: decoder forward (wavenet.py:164-216)
# --------------------------------------------------------------------------
def wavenet_forward(sd: SD, cfg: dict, x: torch.Tensor, c: Optional[torch.Tensor] = None,
                    g: Optional[torch.Tensor] = None, softmax: bool = False, prefix: str = "wavenet.",
                    return_intermediates: bool = False, dropout: Optional[Tuple[float, Sequence[int]]] = None):
    """x (B,C,T) one-hot or scalar (C=1); c (B,Cc,Tc); g (B,) int64 speaker ids or (B,Cg,1) floats.

    cfg keys: layers, stacks, upsample_scales (or None), cin_pad.
    """
    B, _, T = x.shape
    gb = None
    if g is not None:
        if prefix + "embed_speakers.weight" in sd:
            gb = sd[prefix + "embed_speakers.weight"][g.view(B, -1)].transpose(1, 2)  # (B,Cg,1) :185-190
        else:
            gb = g if g.dim() == 3 else g.unsqueeze(-1)
        gb = gb.expand(B, -1, T)  # :194
    if c is not None and cfg.get("upsample_scales"):
        c = upsample_forward(sd, c, cfg["upsample_scales"], prefix + "upsample_net.", cfg.get("cin_pad", 0), cfg.get("conv_in", True),
                             cfg.get("up_act", "none"), cfg.get("up_act_slope", 0.01))
        if c.shape[-1] != T:
            raise Exception("upsampled c length != T")  # :198-200
    h = F.conv1d(x, eff_weight(sd, prefix + "first_conv"), sd[prefix + "first_conv.bias"])  # :203
    skips = 0
    inter = []
    for i, d in enumerate(layer_dilations(cfg["layers"], cfg["stacks"])):
        keep = dropout_keep(dropout[1][i], B, h.shape[1], T, dropout[0]) if dropout is not None else None   # training mode
        h, s = glu_layer_forward(sd, f"{prefix}conv_layers.{i}.", h, c, gb, d, keep, dropout[0] if dropout is not None else 0.0)
        skips = skips + s
        if return_intermediates:
            inter.append((h, s))
    skips = skips * math.sqrt(1.0 / cfg["layers"])  # :208
    y = F.relu(skips)
    y = F.conv1d(y, eff_weight(sd, prefix + "last_conv_layers.1"), sd[prefix + "last_conv_layers.1.bias"])
    y = F.relu(y)
    y = F.conv1d(y, eff_weight(sd, prefix + "last_conv_layers.3"), sd[prefix + "last_conv_layers.3.bias"])
    if softmax:
        y = F.softmax(y, dim=1)
    if return_intermediates:
        return y, c, inter
    return y


# --------------------------------------------------------------------------
# a13: VQVAE composition (vqvae_model.py:66-84)
# --------------------------------------------------------------------------
def vqvae_forward(sd: SD, cfg: dict, x, c, g, softmax: bool = False):
    lat = encoder_forward(sd, c)
    quant, vq_loss, perp, idx = vq_forward(sd["vq.embedding.weight"], lat, cfg.get("beta", 0.25))
    y_hat = wavenet_forward(sd, cfg, x, quant, g, softmax)
    return y_hat, vq_loss, perp, dict(latents=lat, idx=idx, quant=quant)


def vqvae_encode(sd: SD, c: torch.Tensor) -> torch.Tensor:
    """vqvae_model.py:80-84."""
    with torch.no_grad():
        lat = encoder_forward(sd, c)
        return vq_forward(sd["vq.embedding.weight"], lat)[0]


# --------------------------------------------------------------------------
# a9: masked CE (vqwae_train.py:324-334 sequence_mask, :363-379 loss, :764 shift)
# --------------------------------------------------------------------------
def sequence_mask(lengths: torch.Tensor, max_len: Optional[int] = None) -> torch.Tensor:
    if max_len is None:
        max_len = int(lengths.max())
    r = torch.arange(0, max_len).long().unsqueeze(0)
    return (r < lengths.unsqueeze(1)).float()


def masked_ce_loss(y_hat: torch.Tensor, y: torch.Tensor, lengths: torch.Tensor) -> torch.Tensor:
    """y_hat (B,O,T) logits, y (B,T,1) int64 targets, lengths (B,).  Predict y[t+1] from
    y_hat[t]; mask from lengths, dropped first column; sum(mask*CE)/sum(mask)."""
    T = y_hat.shape[-1]
    mask = sequence_mask(lengths, T).unsqueeze(-1)[:, 1:, :]
    losses = F.cross_entropy(y_hat[:, :, :-1].unsqueeze(-1), y[:, 1:, :], reduction="none")
    return (losses * mask).sum() / mask.sum()


# --------------------------------------------------------------------------
# a10: discretized mixture of logistics loss (mixture.py:17-23, :26-106)
# --------------------------------------------------------------------------
def dmol_loss(y_hat: torch.Tensor, y: torch.Tensor, num_classes: int = 256, log_scale_min: float = -7.0,
              reduce: bool = True) -> torch.Tensor:
    """y_hat (B, 3*M, T): [logit pi | mu | log s]; y (B, T, 1) in [-1, 1]."""
    assert y_hat.dim() == 3 and y_hat.shape[1] % 3 == 0
    M = y_hat.shape[1] // 3
    p = y_hat.transpose(1, 2)
    logit_pi, mu = p[..., :M], p[..., M:2 * M]
    log_s = torch.clamp(p[..., 2 * M:], min=log_scale_min)          # :53
    yy = y.expand_as(mu)
    cen = yy - mu
    inv = torch.exp(-log_s)
    half_bin = 1.0 / (num_classes - 1)
    plus_in = inv * (cen + half_bin)
    min_in = inv * (cen - half_bin)
    cdf_delta = torch.sigmoid(plus_in) - torch.sigmoid(min_in)     # :60-75
    log_cdf_plus = plus_in - F.softplus(plus_in)                   # :67
    log_one_minus_cdf_min = -F.softplus(min_in)                    # :71
    mid_in = inv * cen
    log_pdf_mid = mid_in - log_s - 2.0 * F.softplus(mid_in)        # :79
    c3 = (cdf_delta > 1e-5).float()                                # :91
    inner = c3 * torch.log(torch.clamp(cdf_delta, min=1e-12)) + (1 - c3) * (log_pdf_mid - np.log((num_classes - 1) / 2))
    c2 = (yy > 0.999).float()
    mid = c2 * log_one_minus_cdf_min + (1 - c2) * inner
    c1 = (yy < -0.999).float()
    lp = c1 * log_cdf_plus + (1 - c1) * mid
    lp = lp + F.log_softmax(logit_pi, -1)                          # :101
    m = lp.max(-1, keepdim=True)[0]
    lse = m.squeeze(-1) + torch.log(torch.exp(lp - m).sum(-1))     # :17-23
    if reduce:
        return -lse.sum()
    return -lse.unsqueeze(-1)


def masked_dmol_loss(y_hat, y, lengths, num_classes=256, log_scale_min=-7.0):
    """vqwae_train.py:382-401 with the :766 shift."""
    T = y_hat.shape[-1]
    mask = sequence_mask(lengths, T).unsqueeze(-1)[:, 1:, :]
    losses = dmol_loss(y_hat[:, :, :-1], y[:, 1:, :], num_classes, log_scale_min, reduce=False)
    return (losses * mask).sum() / mask.sum()


# --------------------------------------------------------------------------
# a11: DMoL sampler with *explicit* uniforms (mixture.py:118-156)
# --------------------------------------------------------------------------
def dmol_sample(y: torch.Tensor, u_mix: torch.Tensor, u_log: torch.Tensor, log_scale_min: float = -7.0,
                clamp_log_scale: bool = False) -> torch.Tensor:
    """y (B, 3M, T); u_mix (B, T, M) and u_log (B, T) are U(1e-5, 1-1e-5) draws the reference
    makes with .uniform_ (:138,:151).  Gumbel-max (:139-140), select (:143-146), logistic (:151-152),
    clamp [-1,1] (:154).  Returns (B, T)."""
    M = y.shape[1] // 3
    p = y.transpose(1, 2)
    t = p[..., :M] - torch.log(-torch.log(u_mix))
    am = t.argmax(-1, keepdim=True)
    mu = p[..., M:2 * M].gather(-1, am).squeeze(-1)
    ls = p[..., 2 * M:].gather(-1, am).squeeze(-1)
    if clamp_log_scale:
        ls = torch.clamp(ls, min=log_scale_min)
    x = mu + torch.exp(ls) * (torch.log(u_log) - torch.log(1.0 - u_log))
    return torch.clamp(x, -1.0, 1.0)


# --------------------------------------------------------------------------
# a12: incremental (autoregressive) decoder (conv.py:17-62, wavenet.py:218-346)
# --------------------------------------------------------------------------
def incremental_forward(sd: SD, cfg: dict, c_up: Optional[torch.Tensor], g: Optional[torch.Tensor], T: int,
                        test_inputs: Optional[torch.Tensor] = None, initial_input: Optional[torch.Tensor] = None,
                        mode: str = "logits", uniforms: Optional[torch.Tensor] = None,
                        prefix: str = "wavenet.", u_mix: Optional[torch.Tensor] = None, u_log: Optional[torch.Tensor] = None,
                        log_scale_min: float = -7.0) -> torch.Tensor:
    """Sample-by-sample decode with per-layer history, restated with an O(1) ring lookup instead of
    the reference's O(d) buffer shift (conv.py:39) -- same arithmetic: tap j reads x[t-(k-1-j)d]
    (zero before t=0, conv.py:35-36) through the linearized (G, k*R) weight (:51-62).

    c_up: (B, Cc, T) ALREADY upsampled (wavenet.py:276-280 upsamples everything up-front).
    mode: "logits"  -> raw outputs (softmax=False, quantize=False): teacher-forced while t < test_inputs' length, then the
                       output vector of step t-1 is the input of step t (wavenet.py:299-305)
          "probs"   -> the same with softmax=True, quantize=False: the probability vector is fed back
          "argmax"  -> greedy one-hot feedback (deterministic stand-in for OneHotCategorical, :335-338)
          "sample"  -> inverse-CDF categorical draw from ``uniforms`` (B, T)
    x inputs are one-hot (B, O, T) for test_inputs / (B, O, 1) for initial_input.  Returns (B, O, T).
    Scalar-input models (first_conv has one input channel, wavenet.py:284-285,325-333): test_inputs (B, 1, T), the start
    value is 0; "logits" returns the mixture parameters (B, 3M, T) under teacher forcing, "sample" draws every step with
    sample_from_discretized_mix_logistic on the explicit uniforms u_mix (B, T, M), u_log (B, T) and returns (B, 1, T).
    """
    L, stacks = cfg["layers"], cfg["stacks"]
    dil = layer_dilations(L, stacks)
    O = sd[prefix + "last_conv_layers.3.bias"].shape[0]
    B = c_up.shape[0] if c_up is not None else (test_inputs.shape[0] if test_inputs is not None else 1)
    gb = None
    if g is not None:
        if prefix + "embed_speakers.weight" in sd:
            gb = sd[prefix + "embed_speakers.weight"][g.view(B, -1)].squeeze(1)  # (B, Cg)
        else:
            gb = g.view(B, -1)
    W = {}
    for i in range(L):
        p = f"{prefix}conv_layers.{i}."
        w = eff_weight(sd, p + "conv")                              # (G, R, k)
        W[i] = dict(conv=w.transpose(1, 2).contiguous().view(w.shape[0], -1),  # (G, k*R)  conv.py:55-61
                    bias=sd[p + "conv.bias"],
                    c=eff_weight(sd, p + "conv1x1c").squeeze(-1) if c_up is not None else None,
                    g=eff_weight(sd, p + "conv1x1g").squeeze(-1) if gb is not None else None,
                    out=eff_weight(sd, p + "conv1x1_out").squeeze(-1), out_b=sd[p + "conv1x1_out.bias"],
                    skip=eff_weight(sd, p + "conv1x1_skip").squeeze(-1), skip_b=sd[p + "conv1x1_skip.bias"])
        W[i]["k"] = w.shape[-1]
    wf = eff_weight(sd, prefix + "first_conv").squeeze(-1)
    bf = sd[prefix + "first_conv.bias"]
    w1 = eff_weight(sd, prefix + "last_conv_layers.1").squeeze(-1)
    b1 = sd[prefix + "last_conv_layers.1.bias"]
    w3 = eff_weight(sd, prefix + "last_conv_layers.3").squeeze(-1)
    b3 = sd[prefix + "last_conv_layers.3.bias"]
    R = wf.shape[0]
    scalar = wf.shape[1] == 1
    hist = [torch.zeros(B, T, R) for _ in range(L)]   # layer inputs over time
    if initial_input is None:
        cur = torch.zeros(B, 1 if scalar else O)                   # wavenet.py:284-288
        if not scalar and (test_inputs is None or test_inputs.shape[-1] < 1):
            cur[:, 127] = 1.0                                      # wavenet.py:288
    else:
        cur = initial_input.view(B, -1)
    outs = []
    for t in range(T):
        if test_inputs is not None and t < test_inputs.shape[-1]:
            cur = test_inputs[:, :, t]
        elif t > 0:
            cur = outs[-1]
        x = F.linear(cur, wf, bf)
        skips = 0
        for i in range(L):
            w = W[i]
            hist[i][:, t] = x
            k, d = w["k"], dil[i]
            taps = []
            for j in range(k):
                tt = t - (k - 1 - j) * d
                taps.append(hist[i][:, tt] if tt >= 0 else torch.zeros(B, R))
            z = F.linear(torch.cat(taps, dim=1), w["conv"], w["bias"])
            if w["c"] is not None:
                z = z + F.linear(c_up[:, :, t], w["c"])
            if w["g"] is not None:
                z = z + F.linear(gb, w["g"])
            a, b = z.split(z.shape[1] // 2, dim=1)
            u = torch.tanh(a) * torch.sigmoid(b)
            skips = skips + F.linear(u, w["skip"], w["skip_b"])
            x = (F.linear(u, w["out"], w["out_b"]) + x) * math.sqrt(0.5)
        skips = skips * math.sqrt(1.0 / L)
        y = F.linear(F.relu(F.linear(F.relu(skips), w1, b1)), w3, b3)
        if scalar and mode != "logits":
            o = dmol_sample(y.unsqueeze(-1), u_mix[:, t:t + 1], u_log[:, t:t + 1], log_scale_min)   # (B, 1)
        elif mode == "logits":
            o = y
        elif mode == "probs":
            o = F.softmax(y, dim=1)
        else:
            prob = F.softmax(y, dim=1)
            if mode == "argmax":
                idx = prob.argmax(1)
            else:
                cdf = prob.double().cumsum(1)
                idx = (cdf < uniforms[:, t:t + 1].double() * cdf[:, -1:]).sum(1).clamp(max=O - 1)
            o = F.one_hot(idx, O).float()
        outs.append(o)
    return torch.stack(outs, dim=-1)


# --------------------------------------------------------------------------
# a15: optimizer side (vqwae_train.py:776-787, :339-350; lrschedule.py:14-17)
# --------------------------------------------------------------------------
def step_learning_rate_decay(init_lr, global_step, anneal_rate=0.98, anneal_interval=100000):
    return init_lr * anneal_rate ** (global_step // anneal_interval)


def clip_adam_ema_step(params: SD, grads: SD, m: SD, v: SD, shadow: Optional[SD], step: int, lr: float,
                       betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_thresh=100.0, ema_decay=0.9999):
    """One torch.optim.Adam step (no amsgrad) after clip_grad_norm_ (max_norm=clip_thresh, norm 2,
    coef = thresh/(norm+1e-6) clamped to 1) and the reference EMA update
    shadow -= (1-decay)*(shadow-p).  ``step`` is the 1-based Adam step count.  In place; returns grad norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = 1.0
    if clip_thresh > 0:
        coef = min(1.0, float(clip_thresh / (total + 1e-6)))
    b1, b2 = betas
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    for k, p in params.items():
        g = grads[k] * coef
        if weight_decay != 0:
            g = g + weight_decay * p
        m[k].mul_(b1).add_(g, alpha=1 - b1)
        v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[k].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m[k], denom, value=-lr / bc1)
        if shadow is not None:
            shadow[k].sub_((1.0 - ema_decay) * (shadow[k] - p))
    return total


# --------------------------------------------------------------------------
# mu-law companding (nnmnkwii.preprocessing.mulaw_quantize / inv_mulaw_quantize; library absent from
# the image -> restated from its published formula, pinned by hand-computed values only)
# --------------------------------------------------------------------------
def mulaw(x, mu=255):
    x = np.asarray(x, dtype=np.float64)
    return np.sign(x) * np.log1p(mu * np.abs(x)) / np.log1p(mu)


def mulaw_quantize(x, mu=255):
    y = mulaw(x, mu)
    return ((y + 1) / 2 * mu).astype(np.int64)


def inv_mulaw(y, mu=255):
    y = np.asarray(y, dtype=np.float64)
    return np.sign(y) * (1.0 / mu) * ((1.0 + mu) ** np.abs(y) - 1.0)


def inv_mulaw_quantize(y, mu=255):
    y = 2 * np.asarray(y, dtype=np.float64) / mu - 1
    return inv_mulaw(y, mu)


# --------------------------------------------------------------------------
# deterministic closed-form tensors shared by the golden generator and the GPU parity tests
# --------------------------------------------------------------------------
def hash_fill(shape, salt: int, scale: float = 1.0) -> torch.Tensor:
    """Seed-free pseudo-random fill in [-scale, scale): integer hash of the flat index."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64) + np.uint64((int(salt) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    i ^= i >> np.uint64(33)
    i *= np.uint64(0xFF51AFD7ED558CCD)
    i ^= i >> np.uint64(33)
    i *= np.uint64(0xC4CEB9FE1A85EC53)
    i ^= i >> np.uint64(33)
    u = (i >> np.uint64(11)).astype(np.float64) / float(1 << 53)
    return torch.from_numpy(((u * 2 - 1) * scale).astype(np.float32)).view(*shape)


def make_state_dict(cfg: dict, salt: int = 1, with_encoder: bool = True) -> SD:
    """Closed-form state_dict with the reference's key names and shapes.
    cfg: layers, stacks, R, G, S, O, Cc, Cg, k, n_speakers, upsample_scales, encoder_hid, c_in, K, scalar_input."""
    sd: SD = {}
    n = [salt * 1000]

    def fill(shape, fan_in, gain=1.0):
        n[0] += 1
        return hash_fill(shape, n[0], gain * math.sqrt(3.0 / max(fan_in, 1)))

    def wn_conv(prefix, cout, cin, k, bias=True):
        sd[prefix + ".weight_v"] = fill((cout, cin, k), cin * k)
        n[0] += 1
        sd[prefix + ".weight_g"] = (1.0 + 0.25 * hash_fill((cout, 1, 1), n[0])) * \
            sd[prefix + ".weight_v"].pow(2).sum((1, 2), keepdim=True).sqrt()
        if bias:
            sd[prefix + ".bias"] = fill((cout,), 1, 0.05)

    R, G, S, O, Cc, Cg, k = cfg["R"], cfg["G"], cfg["S"], cfg["O"], cfg["Cc"], cfg["Cg"], cfg.get("k", 3)
    H = G // 2
    in_ch = 1 if cfg.get("scalar_input") else O
    wn_conv("wavenet.first_conv", R, in_ch, 1)
    for i in range(cfg["layers"]):
        p = f"wavenet.conv_layers.{i}."
        wn_conv(p + "conv", G, R, k)
        if Cc > 0:
            wn_conv(p + "conv1x1c", G, Cc, 1, bias=False)
        if Cg > 0:
            wn_conv(p + "conv1x1g", G, Cg, 1, bias=False)
        wn_conv(p + "conv1x1_out", R, H, 1)
        wn_conv(p + "conv1x1_skip", S, H, 1)
    wn_conv("wavenet.last_conv_layers.1", S, S, 1)
    wn_conv("wavenet.last_conv_layers.3", O, S, 1)
    if Cg > 0 and cfg.get("n_speakers"):
        n[0] += 1
        sd["wavenet.embed_speakers.weight"] = hash_fill((cfg["n_speakers"], Cg), n[0], 0.3)
    if cfg.get("upsample_scales"):
        conv_in = cfg.get("conv_in", True)         # False: the plain UpsampleNetwork's keys (upsample.py:29-66)
        if conv_in:
            sd["wavenet.upsample_net.conv_in.weight"] = fill((Cc, Cc, 2 * cfg.get("cin_pad", 0) + 1), Cc, 1.5)
        for i, s in enumerate(cfg["upsample_scales"]):
            p = f"wavenet.upsample_net.{'upsample.' if conv_in else ''}up_layers.{(2 if cfg.get('up_act', 'none') == 'none' else 3) * i + 1}"
            n[0] += 1
            v = 1.0 / (2 * s + 1) + 0.02 * hash_fill((1, 1, 1, 2 * s + 1), n[0])
            sd[p + ".weight_v"] = v
            sd[p + ".weight_g"] = v.pow(2).sum((1, 2, 3), keepdim=True).sqrt() * 1.1
    if with_encoder:
        hid, cin = cfg["encoder_hid"], cfg["c_in"]
        dims = [(cin, hid)] + [(hid, hid)] * 9
        for i, ((ci, co), (kk, _)) in enumerate(zip(dims, ENCODER_BLOCKS)):
            sd[f"encoder.net.{i}.conv.weight"] = fill((co, ci, kk), ci * kk, 0.8)
            sd[f"encoder.net.{i}.conv.bias"] = fill((co,), 1, 0.05)
        sd["encoder.lin.weight"] = fill((Cc, hid), hid)
        sd["encoder.lin.bias"] = fill((Cc,), 1, 0.05)
        n[0] += 1
        sd["vq.embedding.weight"] = hash_fill((cfg.get("K", 256), Cc), n[0], 1.5)
    return sd
