#!/usr/bin/env python
"""Golden-vector generator.  Runs ONLY in the build container (needs /root/reference).

Imports the reference's own modules (never copied, never shipped), loads closed-form
weights (oracle.make_state_dict -- a seed-free index hash) into them, runs the reference
code and stores inputs + expected outputs as small .npz fixtures next to this script.
While doing so it also asserts that the oracle restatement agrees with the reference.

    python tests/golden/make_golden.py
"""
import json
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, REF)

import vqvae_model as ref_vqvae  # noqa: E402
import vector_quantization as ref_vq  # noqa: E402
import lrschedule as ref_lr  # noqa: E402
from wavenet_vocoder import WaveNet as RefWaveNet  # noqa: E402
from wavenet_vocoder import mixture as ref_mix  # noqa: E402
from wavenet_vocoder import upsample as ref_up  # noqa: E402
from wavenet_vocoder.modules import ResidualConv1dGLU as RefGLU  # noqa: E402

from oracle import wae_oracle as O  # noqa: E402

torch.manual_seed(0)
torch.set_num_threads(4)


# out-of-place residual add: the reference's in-place `out += x` on a ReLU output breaks autograd
# under torch >= 1.x.  Mathematically identical; reference files untouched.
def _crr_forward(self, x):
    out = self.relu(self.conv(x))
    if self.stride == 1 and self.dim_in == self.dim_out:
        out = out + x
    return out


ref_vqvae.ConvReLURes.forward = _crr_forward

CFG_A = dict(name="A", layers=4, stacks=2, R=32, G=48, S=32, O=64, Cc=16, Cg=8, k=3, n_speakers=5,
             upsample_scales=[4, 4, 8, 5], encoder_hid=32, c_in=39, K=32, cin_pad=0)
CFG_B = dict(name="B", layers=6, stacks=2, R=32, G=80, S=64, O=64, Cc=16, Cg=16, k=3, n_speakers=4,
             upsample_scales=[4, 4, 4, 5], encoder_hid=32, c_in=39, K=32, cin_pad=0)
CFG_S = dict(name="S", layers=4, stacks=2, R=32, G=64, S=32, O=30, Cc=16, Cg=8, k=3, n_speakers=5,
             upsample_scales=[4, 4, 8, 5], encoder_hid=32, c_in=39, K=32, cin_pad=0, scalar_input=True)


def build_ref_wavenet(cfg):
    return RefWaveNet(out_channels=cfg["O"], layers=cfg["layers"], stacks=cfg["stacks"],
                      residual_channels=cfg["R"], gate_channels=cfg["G"], skip_out_channels=cfg["S"],
                      kernel_size=cfg["k"], dropout=0.0, cin_channels=cfg["Cc"], gin_channels=cfg["Cg"],
                      n_speakers=cfg["n_speakers"], upsample_conditional_features=True,
                      upsample_net="ConvInUpsampleNetwork",
                      upsample_params=dict(upsample_scales=cfg["upsample_scales"], cin_channels=cfg["Cc"], cin_pad=0),
                      scalar_input=bool(cfg.get("scalar_input")), use_speaker_embedding=True,
                      output_distribution="Logistic", cin_pad=0)


def build_ref_vqvae(cfg, sd):
    wn = build_ref_wavenet(cfg)
    m = ref_vqvae.VQVAE(c_in=cfg["c_in"], hid=cfg["Cc"], K=cfg["K"], wavenet=wn, encoder_hid=cfg["encoder_hid"])
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m


def close(a, b, tol=2e-5, what=""):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    err = (a - b).abs().max().item()
    ref = b.abs().max().item() + 1e-12
    assert err <= tol * max(1.0, ref), f"oracle != reference for {what}: {err} (ref max {ref})"
    return err


def inputs_for(cfg, B, F, salt):
    """c (B, c_in, F) MFCC-like, x indices (B, T), g (B,), with T = F/4 * prod(scales)."""
    hop = int(np.prod(cfg["upsample_scales"]))
    Tq = ((F - 1) // 2 + 1 - 1) // 2 + 1
    T = Tq * hop
    c = O.hash_fill((B, cfg["c_in"], F), salt + 1, 1.7)
    if cfg.get("scalar_input"):
        x = O.hash_fill((B, 1, T), salt + 2, 0.98)
        xin = x
    else:
        x = ((O.hash_fill((B, T), salt + 2) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
        xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    g = ((O.hash_fill((B,), salt + 3) * 0.5 + 0.5) * cfg["n_speakers"]).long().clamp(0, cfg["n_speakers"] - 1)
    return c, x, xin, g, T


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def gen_model(cfg, salt):
    sd = O.make_state_dict(cfg, salt)
    B, F = 2, 8
    c, x, xin, g, T = inputs_for(cfg, B, F, salt * 10)
    model = build_ref_vqvae(cfg, sd).eval()
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    with torch.no_grad():
        lat = model.encoder(c)
        quant, vq_loss, perp = model.vq(lat)
        c_up = model.wavenet.upsample_net(quant)
        y_hat, vq2, perp2 = model(xin, c, g, False)
        y_sm = model.wavenet(xin, quant, g, True)
    # oracle agreement
    o_lat = O.encoder_forward(sd, c)
    close(o_lat, lat, what="latents")
    o_q, o_vq, o_perp, o_idx = O.vq_forward(sd["vq.embedding.weight"], lat)
    close(o_q, quant, what="quant")
    close(o_vq, vq_loss, what="vq_loss")
    close(o_perp, perp, what="perp")
    dist = O.vq_distances(sd["vq.embedding.weight"], lat)
    top2 = dist.topk(2, dim=1, largest=False)[0]
    margin = (top2[:, 1] - top2[:, 0])
    qf = quant.permute(0, 2, 1).reshape(-1, cfg["Cc"])
    ref_idx = ((qf[:, None, :] - sd["vq.embedding.weight"][None]) ** 2).sum(-1).argmin(1)   # x+(q-x) rounds, so nearest row
    assert torch.equal(ref_idx, o_idx)
    close(O.upsample_forward(sd, quant, cfg["upsample_scales"]), c_up, what="c_up")
    o_y, o_vq2, o_perp2, _ = O.vqvae_forward(sd, ocfg, xin, c, g)
    e = close(o_y, y_hat, what="logits")
    print(f"  cfg {cfg['name']}: T={T} logits max|oracle-ref| = {e:.2e}, min VQ margin = {margin.min().item():.3e}")
    close(O.wavenet_forward(sd, ocfg, xin, quant, g, softmax=True), y_sm, what="softmax")
    save(f"model_{cfg['name']}", cfg=json.dumps(cfg), salt=salt, c=c, x=x.numpy() if not cfg.get("scalar_input") else x,
         g=g, latents=lat, quant=quant, vq_idx=o_idx, vq_margin=margin, vq_loss=vq_loss, perp=perp, c_up=c_up,
         y_hat=y_hat, y_softmax_probe=y_sm[:, :, ::37])
    return sd, model, (c, x, xin, g, T), ocfg


def gen_layers(cfg, sd, model, ins):
    """(4) single GLU layers at several dilations, with and without conditioning."""
    c, x, xin, g, T = ins
    B = 2
    R, Cc, Cg = cfg["R"], cfg["Cc"], cfg["Cg"]
    xr = O.hash_fill((B, R, T), 901, 1.2)
    cr = O.hash_fill((B, Cc, T), 902, 1.0)
    gr = O.hash_fill((B, Cg, 1), 903, 0.5)
    probe_t = torch.cat([torch.arange(0, 48), torch.arange(48, T, 11)])
    out = dict(x_salt=901, c_salt=902, g_salt=903, x_scale=1.2, c_scale=1.0, g_scale=0.5, T=T, probe_t=probe_t)
    for d in (1, 2, 512):
        lay = RefGLU(R, cfg["G"], kernel_size=3, skip_out_channels=cfg["S"], cin_channels=Cc, gin_channels=Cg,
                     dropout=0.0, dilation=d, bias=True).eval()
        p = "wavenet.conv_layers.1."
        lsd = {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
        lay.load_state_dict(lsd, strict=True)
        with torch.no_grad():
            for tag, (cc, gg) in dict(cg=(cr, gr), none=(None, None)).items():
                xo, so = lay(xr, cc, None if gg is None else gg.expand(B, Cg, T).contiguous())
                oxo, oso = O.glu_layer_forward(sd, p, xr, cc, gg, d)
                close(oxo, xo, what=f"glu x d={d}")
                close(oso, so, what=f"glu s d={d}")
                out[f"xo_d{d}_{tag}"] = xo[:, :, probe_t]
                out[f"so_d{d}_{tag}"] = so[:, :, probe_t]
    save(f"glu_{cfg['name']}", **out)


def gen_noncausal_layer():
    """ResidualConv1dGLU(causal=False) (modules.py:82-88: symmetric padding (k-1)//2 * d, nothing trimmed): forward and the reference's
    own autograd gradients of a fixed scalar function of (x', s) at two dilations -> glu_noncausal.npz."""
    cfg = CFG_A
    sd = O.make_state_dict(dict(cfg), 1)
    R, Cc, Cg, G, S = cfg["R"], cfg["Cc"], cfg["Cg"], cfg["G"], cfg["S"]
    B, T = 2, 200
    p = "wavenet.conv_layers.1."
    lsd = {k[len(p):]: v for k, v in sd.items() if k.startswith(p)}
    out = dict(T=T, B=B, x_salt=921, c_salt=922, g_salt=923, wx_salt=924, ws_salt=925, scale=0.8)
    x0, c0 = O.hash_fill((B, R, T), 921, 0.8), O.hash_fill((B, Cc, T), 922, 0.8)
    gv = O.hash_fill((B, Cg, 1), 923, 0.8)
    wx, wsk = O.hash_fill((B, R, T), 924), O.hash_fill((B, S, T), 925)
    for d in (1, 8):
        lay = RefGLU(R, G, kernel_size=3, skip_out_channels=S, cin_channels=Cc, gin_channels=Cg, dropout=0.0, dilation=d,
                     causal=False, bias=True).eval()
        lay.load_state_dict({k: v.clone() for k, v in lsd.items()}, strict=True)
        xr, cr = x0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
        xo, so = lay(xr, cr, gv.expand(B, Cg, T).contiguous())
        assert xo.shape == (B, R, T) and so.shape == (B, S, T)
        ((xo * wx).sum() + (so * wsk).sum()).backward()
        psd = {p + k: v.clone().requires_grad_(True) for k, v in lsd.items()}
        xq, cq = x0.clone().requires_grad_(True), c0.clone().requires_grad_(True)
        oxo, oso = O.glu_layer_forward(psd, p, xq, cq, gv.expand(B, Cg, T), d, causal=False)
        ((oxo * wx).sum() + (oso * wsk).sum()).backward()
        close(oxo, xo, what=f"non-causal glu x d={d}")
        close(oso, so, what=f"non-causal glu s d={d}")
        close(xq.grad, xr.grad, tol=1e-4, what=f"non-causal dx d={d}")
        close(cq.grad, cr.grad, tol=1e-4, what=f"non-causal dc d={d}")
        out[f"xo_d{d}"], out[f"so_d{d}"], out[f"dx_d{d}"], out[f"dc_d{d}"] = xo, so, xr.grad, cr.grad
        for k, v in lay.named_parameters():
            close(psd[p + k].grad, v.grad, tol=1e-4, what=f"non-causal grad {k} d={d}")
            out[f"grad_d{d}:{k}"] = v.grad
    save("glu_noncausal", **out)


def gen_losses(cfg, sd, model, ins, ocfg):
    """(6) CE (masked, shifted) value + logits-gradient."""
    c, x, xin, g, T = ins
    lengths = torch.tensor([T, T - 137])
    y = x.unsqueeze(-1)
    y_hat = model(xin, c, g, False)[0].detach().requires_grad_(True)
    mask = O.sequence_mask(lengths, T).unsqueeze(-1)[:, 1:, :]
    crit = torch.nn.CrossEntropyLoss(reduction="none")
    losses = crit(y_hat[:, :, :-1].unsqueeze(-1), y[:, 1:, :])
    loss = (losses * mask).sum() / mask.sum()
    loss.backward()
    yo = y_hat.detach().clone().requires_grad_(True)
    lo = O.masked_ce_loss(yo, y, lengths)
    lo.backward()
    close(lo, loss, what="ce")
    close(yo.grad, y_hat.grad, what="ce grad", tol=1e-6)
    save(f"ce_{cfg['name']}", lengths=lengths, loss=loss.detach(), dlogits_probe=y_hat.grad[:, :, ::29])


def gen_dmol():
    """(6b) DMoL loss incl. edge cases and the sampler."""
    B, T, M = 2, 96, 10
    y_hat = O.hash_fill((B, 3 * M, T), 777, 3.0)
    y_hat[:, 2 * M:, :] = O.hash_fill((B, M, T), 778, 6.0) - 5.0      # log-scales in [-11, 1]
    y = O.hash_fill((B, T, 1), 779, 1.0)
    y[0, :6, 0] = torch.tensor([-1.0, -0.9995, 1.0, 0.9995, 0.999, -0.999])
    # force cdf_delta <= 1e-5: tiny scale and target far from every mean
    y_hat[1, 2 * M:, 5] = -12.0
    y_hat[1, M:2 * M, 5] = 0.9
    y[1, 5, 0] = -0.5
    yh = y_hat.clone().requires_grad_(True)
    out = {}
    for lsm in (-7.0, -9.0):
        if yh.grad is not None:
            yh.grad = None
        l_el = ref_mix.discretized_mix_logistic_loss(yh, y, num_classes=256, log_scale_min=lsm, reduce=False)
        l_sum = ref_mix.discretized_mix_logistic_loss(yh, y, num_classes=256, log_scale_min=lsm, reduce=True)
        l_sum.backward()
        yo = y_hat.clone().requires_grad_(True)
        o_el = O.dmol_loss(yo, y, 256, lsm, reduce=False)
        O.dmol_loss(yo, y, 256, lsm, reduce=True).backward()
        close(o_el, l_el, what="dmol")
        close(yo.grad, yh.grad, what="dmol grad", tol=1e-5)
        out[f"loss_el_{int(-lsm)}"] = l_el.detach()
        out[f"loss_sum_{int(-lsm)}"] = l_sum.detach()
        out[f"grad_{int(-lsm)}"] = yh.grad.clone()
    # 65536-class variant (raw 16-bit input default, hparams.py:21)
    l16 = ref_mix.discretized_mix_logistic_loss(y_hat, y, num_classes=65536, log_scale_min=-16.0, reduce=False)
    close(O.dmol_loss(y_hat, y, 65536, -16.0, reduce=False), l16, what="dmol16")
    out["loss_el_65536"] = l16
    # sampler: reproduce the reference's uniform_ draws from the global RNG
    torch.manual_seed(4242)
    s_ref = ref_mix.sample_from_discretized_mix_logistic(y_hat, log_scale_min=-7.0)
    torch.manual_seed(4242)
    u_mix = torch.empty(B, T, M).uniform_(1e-5, 1.0 - 1e-5)
    u_log = torch.empty(B, T).uniform_(1e-5, 1.0 - 1e-5)
    s_or = O.dmol_sample(y_hat, u_mix, u_log, -7.0)
    close(s_or, s_ref, what="dmol sample")
    save("dmol", y_hat=y_hat, y=y, u_mix=u_mix, u_log=u_log, sample=s_ref, **out)


def gen_train_step(cfg, sd, ins, ocfg):
    """(7) parameter gradients of one step, post-Adam weights, EMA shadow (vqwae_train.py:709-798)."""
    c, x, xin, g, T = ins
    model = build_ref_vqvae(cfg, {k: v.clone() for k, v in sd.items()}).train()
    lengths = torch.tensor([T, T])
    y = x.unsqueeze(-1)
    opt = torch.optim.Adam(model.parameters(), lr=4e-4, eps=1e-8, weight_decay=0.0)
    shadow = {n: p.data.clone() for n, p in model.named_parameters() if p.requires_grad}
    mask = O.sequence_mask(lengths, T).unsqueeze(-1)[:, 1:, :]
    opt.zero_grad()
    y_hat, vq_loss, perp = model(xin, c, g, False)
    crit = torch.nn.CrossEntropyLoss(reduction="none")
    ce = ((crit(y_hat[:, :, :-1].unsqueeze(-1), y[:, 1:, :]) * mask).sum()) / mask.sum()
    loss = ce + vq_loss.mean()
    loss.backward()
    none = [n for n, p in model.named_parameters() if p.grad is None]
    print("  params without grad in the reference step:", none)
    grads = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 100.0)
    opt.step()
    for n, p in model.named_parameters():
        d = shadow[n] - p.data
        shadow[n] -= (1.0 - 0.9999) * d
    # oracle: autograd through the functional restatement + own Adam/EMA
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    oy, ovq, operp, _ = O.vqvae_forward(psd, ocfg, xin, c, g)
    oloss = O.masked_ce_loss(oy, y, lengths) + ovq
    oloss.backward()
    close(oloss, loss, what="train loss")
    ograds = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in psd.items()}
    for k in grads:
        close(ograds[k], grads[k], what=f"grad {k}", tol=2e-4)
    params = {k: v.detach().clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    vv = {k: torch.zeros_like(v) for k, v in params.items()}
    osh = {k: v.clone() for k, v in params.items()}
    ogn = O.clip_adam_ema_step(params, {k: v.detach() for k, v in ograds.items()}, m, vv, osh, 1, 4e-4)
    close(ogn, gn, what="grad norm", tol=1e-4)
    new = dict(model.named_parameters())
    for k in params:
        close(params[k], new[k].data, what=f"adam {k}", tol=1e-5)
        close(osh[k], shadow[k], what=f"ema {k}", tol=1e-6)
    keys = ["wavenet.conv_layers.0.conv.weight_v", "wavenet.conv_layers.0.conv.weight_g",
            "wavenet.conv_layers.2.conv1x1c.weight_v", "wavenet.conv_layers.3.conv1x1_skip.bias",
            "wavenet.first_conv.weight_v", "wavenet.last_conv_layers.3.weight_v", "wavenet.embed_speakers.weight",
            "wavenet.upsample_net.conv_in.weight", "wavenet.upsample_net.upsample.up_layers.5.weight_v",
            "encoder.net.0.conv.weight", "encoder.net.3.conv.weight", "encoder.lin.weight", "vq.embedding.weight"]
    out = dict(loss=loss.detach(), ce=ce.detach(), vq_loss=vq_loss.detach(), perp=perp.detach(), grad_norm=gn)
    for k in keys:
        out["grad:" + k] = grads[k]
        out["new:" + k] = new[k].data
        out["ema:" + k] = shadow[k]
    out["grad_sq_by_key"] = json.dumps({k: float((v.double() ** 2).sum()) for k, v in grads.items()})
    save(f"train_{cfg['name']}", **out)


def gen_ar(cfg, sd, model, ins, ocfg):
    """(8) incremental_forward: teacher-forced (== forward) and a greedy roll-out."""
    c, x, xin, g, T = ins
    Tar = 96
    with torch.no_grad():
        lat = model.encoder(c)
        quant = model.vq(lat)[0]
        c_up = model.wavenet.upsample_net(quant)[:, :, :Tar].contiguous()
        # feed pre-upsampled c by disabling the upsample net on a fresh reference decoder
        wn = build_ref_wavenet(cfg).eval()
        wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items() if k.startswith("wavenet.")})
        wn.upsample_net = None
        init0 = torch.zeros(2, cfg["O"], 1)
        init0[:, cfg["O"] // 2 - 1, 0] = 1
        tf = wn.incremental_forward(init0, c=c_up, g=g, T=Tar, test_inputs=xin[:, :, :Tar].contiguous(),
                                    softmax=False, quantize=False)
        fwd = wn(xin[:, :, :Tar].contiguous(), c_up, g, False)
        close(tf, fwd, what="incremental == forward", tol=1e-5)
        wn.make_generation_fast_()
        fwd_fast = wn(xin[:, :, :Tar].contiguous(), c_up, g, False)
        assert (fwd_fast - fwd).abs().max().item() < 1e-5
    o_tf = O.incremental_forward(sd, ocfg, c_up, g, Tar, test_inputs=xin[:, :, :Tar], mode="logits")
    close(o_tf, tf, what="oracle AR teacher-forced", tol=1e-5)
    # greedy roll-out: reference has no argmax mode; drive its one-step API by feeding back argmax ourselves
    with torch.no_grad():
        wn2 = build_ref_wavenet(cfg).eval()
        wn2.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items() if k.startswith("wavenet.")})
        wn2.upsample_net = None
        cur = torch.zeros(2, cfg["O"], 1)
        cur[:, cfg["O"] // 2 - 1, 0] = 1
        init = cur.clone()
        seq = []
        hist = []
        for t in range(24):
            hist.append(cur)
            full = torch.cat(hist, dim=-1)
            out = wn2.incremental_forward(init, c=c_up[:, :, :t + 1].contiguous(), g=g, T=t + 1, test_inputs=full,
                                          softmax=False, quantize=False)
            idx = out[:, :, -1].argmax(1)
            seq.append(idx)
            cur = torch.nn.functional.one_hot(idx, cfg["O"]).float().unsqueeze(-1)
        greedy = torch.stack(seq, dim=1)
    o_g = O.incremental_forward(sd, ocfg, c_up[:, :, :24].contiguous(), g, 24, initial_input=init, mode="argmax")
    assert torch.equal(o_g.argmax(1), greedy), "greedy roll-out differs"
    # dense feedback (wavenet.py:299-305,335-338 with quantize=False): the softmax probabilities, or the raw logits, of step t
    # are the decoder input of step t+1; and partial teacher forcing (test_inputs shorter than T: forced, then free-running)
    Ts, nf = 40, 9
    with torch.no_grad():
        c40 = c_up[:, :, :Ts].contiguous()
        # (the reference takes the batch size from test_inputs, wavenet.py:256: the start vector goes in as a one-step test_inputs)
        soft = wn2.incremental_forward(init, c=c40, g=g, T=Ts, test_inputs=init, softmax=True, quantize=False)
        raw = wn2.incremental_forward(init, c=c40, g=g, T=Ts, test_inputs=init, softmax=False, quantize=False)
        part = wn2.incremental_forward(init, c=c40, g=g, T=Ts, test_inputs=xin[:, :, :nf].contiguous(), softmax=True,
                                       quantize=False)
    close(O.incremental_forward(sd, ocfg, c40, g, Ts, initial_input=init, mode="probs"), soft, what="oracle soft feedback", tol=1e-5)
    close(O.incremental_forward(sd, ocfg, c40, g, Ts, initial_input=init, mode="logits"), raw, what="oracle raw feedback", tol=1e-5)
    close(O.incremental_forward(sd, ocfg, c40, g, Ts, initial_input=init, test_inputs=xin[:, :, :nf], mode="probs"), part,
          what="oracle partial teacher forcing", tol=1e-5)
    save(f"ar_{cfg['name']}", c_up=c_up, tf_logits=tf, greedy=greedy, init=init, soft=soft, raw=raw, part=part, part_forced=nf)


def gen_ar_scalar(cfg, sd, model, ins, ocfg):
    """(8b) incremental_forward of a scalar-input decoder (wavenet.py:284-285,325-333).  The reference draws every step with
    sample_from_discretized_mix_logistic on torch's RNG; its draw is replaced here by the oracle's explicit-uniform restatement
    (itself pinned by dmol.npz), so that the reference's own incremental loop yields reproducible vectors: the mixture
    parameters of every step under teacher forcing, and a free-running roll-out."""
    import wavenet_vocoder.wavenet as ref_wn_mod
    c, x, xin, g, T = ins
    Tar, M = 64, cfg["O"] // 3
    with torch.no_grad():
        lat = model.encoder(c)
        quant = model.vq(lat)[0]
        c_up = model.wavenet.upsample_net(quant)[:, :, :Tar].contiguous()
    rng = np.random.default_rng(77)
    u_mix = torch.from_numpy(rng.uniform(1e-5, 1 - 1e-5, size=(2, Tar, M)).astype(np.float32))
    u_log = torch.from_numpy(rng.uniform(1e-5, 1 - 1e-5, size=(2, Tar)).astype(np.float32))
    rec, step = [], [0]

    def patched(y, log_scale_min=-7.0, clamp_log_scale=False):
        t = step[0]
        step[0] += 1
        rec.append(y.clone())
        return O.dmol_sample(y, u_mix[:, t:t + 1], u_log[:, t:t + 1], log_scale_min, clamp_log_scale)

    orig = ref_wn_mod.sample_from_discretized_mix_logistic
    ref_wn_mod.sample_from_discretized_mix_logistic = patched
    try:
        with torch.no_grad():
            wn = build_ref_wavenet(cfg).eval()
            wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items() if k.startswith("wavenet.")})
            wn.upsample_net = None
            tf_in = xin[:, :, :Tar].transpose(1, 2).contiguous()                 # (B, T, 1)
            wn.incremental_forward(None, c=c_up, g=g, T=Tar, test_inputs=tf_in, log_scale_min=-7.0)
            params_tf = torch.cat(rec, dim=-1)                                   # (B, 3M, Tar)
            fwd = wn(xin[:, :, :Tar].contiguous(), c_up, g, False)
            close(params_tf, fwd, what="scalar incremental parameters == forward", tol=1e-5)
            rec.clear()
            step[0] = 0
            # free-running from the zero start value; a one-step test_inputs only tells the reference the batch size
            roll = wn.incremental_forward(None, c=c_up[:, :, :24].contiguous(), g=g, T=24, test_inputs=torch.zeros(2, 1, 1),
                                          log_scale_min=-7.0)                    # (B, 1, 24)
    finally:
        ref_wn_mod.sample_from_discretized_mix_logistic = orig
    o_tf = O.incremental_forward(sd, ocfg, c_up, g, Tar, test_inputs=xin[:, :, :Tar], mode="logits")
    close(o_tf, params_tf, what="oracle scalar AR teacher-forced", tol=1e-5)
    o_roll = O.incremental_forward(sd, ocfg, c_up[:, :, :24].contiguous(), g, 24, mode="sample", u_mix=u_mix, u_log=u_log,
                                   log_scale_min=-7.0)
    close(o_roll, roll, what="oracle scalar AR roll-out", tol=1e-5)
    save(f"ar_{cfg['name']}", c_up=c_up, params_tf=params_tf, roll=roll, u_mix=u_mix, u_log=u_log)


def gen_misc():
    steps = [0, 1, 399999, 400000, 400001, 800000, 1200000]
    lr = [ref_lr.step_learning_rate_decay(4e-4, s, anneal_rate=0.5, anneal_interval=400000) for s in steps]
    assert lr == [O.step_learning_rate_decay(4e-4, s, 0.5, 400000) for s in steps]
    noam = [float(ref_lr.noam_learning_rate_decay(1e-3, s)) for s in steps]
    cyc = [float(ref_lr.cyclic_cosine_annealing(1e-3, s, 1000, 5)) for s in steps]
    from hparams import hparams as ref_hp
    defaults = dict(ref_hp.values())
    status = {}
    vq = None
    for f in sorted(os.listdir(os.path.join(REF, "hps"))):
        from hparams import hparams as hp
        import copy
        h = copy.deepcopy(hp)
        try:
            with open(os.path.join(REF, "hps", f)) as fh:
                h.parse_json(fh.read())
            status[f] = "ok"
            if f == "vqwae.json":
                h.parse("layers=24,batch_size=8,ema_decay=0.99")
                vq = dict(h.values())
        except Exception as e:  # noqa: BLE001
            status[f] = type(e).__name__
    rf = {f"{L}_{s}_{k}": RefWaveNetRF(L, s, k) for (L, s, k) in [(20, 2, 3), (24, 2, 3), (24, 4, 3), (48, 4, 3), (4, 2, 3)]}
    with open(os.path.join(HERE, "misc.json"), "w") as fh:
        json.dump(dict(lr_steps=steps, step_lr=lr, noam=noam, cyclic=cyc, hparams_defaults=defaults,
                       preset_status=status, vqwae_parsed_with_overrides=vq, receptive_field=rf), fh, indent=1, sort_keys=True)
    print("wrote misc.json", status)


def RefWaveNetRF(L, s, k):
    from wavenet_vocoder import receptive_field_size
    r = receptive_field_size(L, s, k)
    assert r == O.receptive_field_size(L, s, k)
    return r


def gen_vqwae_probe():
    """Full-size hps/vqwae.json model: sparse logits probe + sums (closed-form weights, short clip)."""
    cfg = dict(name="vqwae", layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153,
               upsample_scales=[4, 4, 8, 5], encoder_hid=256, c_in=39, K=256, cin_pad=0)
    sd = O.make_state_dict(cfg, 7)
    assert len(sd) == 302 and sum(v.numel() for v in sd.values()) == 7555218, (len(sd), sum(v.numel() for v in sd.values()))
    c, x, xin, g, T = inputs_for(cfg, 1, 16, 70)
    model = build_ref_vqvae(cfg, sd).eval()
    with torch.no_grad():
        y_hat, vq_loss, perp = model(xin, c, g, False)
    ocfg = dict(layers=20, stacks=2, upsample_scales=cfg["upsample_scales"], cin_pad=0)
    oy, ovq, operp, aux = O.vqvae_forward(sd, ocfg, xin, c, g)
    e = close(oy, y_hat, what="vqwae logits", tol=5e-5)
    print(f"  vqwae.json full-size: T={T}, max|oracle-ref| = {e:.2e}; keys={len(sd)}")
    ti = torch.arange(0, T, 41)
    save("model_vqwae_probe", cfg=json.dumps(cfg), salt=7, probe_t=ti, y_probe=y_hat[0][:, ti], vq_idx=aux["idx"],
         y_sum=y_hat.double().sum(), y_abs_sum=y_hat.double().abs().sum(), vq_loss=vq_loss, perp=perp,
         keys=json.dumps({k: list(v.shape) for k, v in sd.items()}))


def gen_wide_probe():
    """BASELINE config C5's widths (R = G = S = 512, Cc = 64, Cg = 32) on a short stack: the reference's own WaveNet on
    audio-rate conditioning -> sparse logits probe + sums; inputs are the closed-form fills of tests/test_gpu_wide.py."""
    cfg = dict(name="wide", layers=4, stacks=2, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=8,
               upsample_scales=None, cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
    B, T = 2, 777
    x = ((O.hash_fill((B, T), 21) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    c = O.hash_fill((B, cfg["Cc"], T), 22, 1.3)
    g = torch.tensor([1, 6])
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    model = RefWaveNet(out_channels=256, layers=4, stacks=2, residual_channels=512, gate_channels=512, skip_out_channels=512,
                       kernel_size=3, dropout=0.0, cin_channels=64, gin_channels=32, n_speakers=8,
                       upsample_conditional_features=False, scalar_input=False, use_speaker_embedding=True,
                       output_distribution="Logistic", cin_pad=0)
    missing = model.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.eval()
    with torch.no_grad():
        y_hat = model(xin, c, g, False)
        oy = O.wavenet_forward(sd, dict(layers=4, stacks=2, upsample_scales=None, cin_pad=0), xin, c, g)
    e = close(oy, y_hat, what="wide logits", tol=5e-5)
    print(f"  wide (512 channels): T={T}, max|oracle-ref| = {e:.2e}")
    ti = torch.arange(0, T, 13)
    save("model_wide_probe", cfg=json.dumps(cfg), salt=5, x_salt=21, c_salt=22, g=g, T=T, probe_t=ti, y_probe=y_hat[:, :, ti],
         y_sum=y_hat.double().sum(), y_abs_sum=y_hat.double().abs().sum())


def gen_quantizers():
    """Section 8(f) rank 3: SlicedVectorQuantize, SlicedVectorQuantizeEMA, VectorQuantizeEMA (vector_quantization.py:51-306)
    run on this CPU host.  The two EMA classes test the function object `torch.cuda.is_available` (:183, :277), which is
    always truthy, and then call .cuda(); Tensor.cuda is made the identity for the duration (reference files untouched).
    Two training steps then one eval step each; gradients of sum(quant * w) + vq_loss."""
    B, D, T, K, K1 = 3, 16, 37, 24, 20
    gen = torch.Generator().manual_seed(11)
    lats = [torch.randn(B, D, T, generator=gen) * 0.3 for _ in range(3)]
    w = torch.randn(B, D, T, generator=gen)
    e1 = (torch.rand(K, D // 2, generator=gen) - 0.5) * 0.8
    e2 = (torch.rand(K1, D // 2, generator=gen) - 0.5) * 0.8
    e2k = (torch.rand(K, D // 2, generator=gen) - 0.5) * 0.8
    ef = (torch.rand(K, D, generator=gen) - 0.5) * 0.8
    out = dict(lats=torch.stack(lats), w=w, e1=e1, e2=e2, e2k=e2k, ef=ef)
    keep = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        # --- SlicedVectorQuantize (K codes / K1 codes)
        m = ref_vq.SlicedVectorQuantize(K, D, beta=0.25, K1=K1)
        m.embedding1.weight.data.copy_(e1)
        m.embedding2.weight.data.copy_(e2)
        x = lats[0].clone().requires_grad_(True)
        q, loss, perp = m(x)
        (q * w).sum().add(loss).backward()
        oq, ol, op, (i1, i2) = O.sliced_vq_forward(e1, e2, lats[0], 0.25)
        close(oq, q, what="oracle sliced quant", tol=1e-6)
        close(ol, loss, what="oracle sliced loss", tol=1e-6)
        close(op, perp, what="oracle sliced perp", tol=1e-5)
        out.update(s_quant=q, s_loss=loss, s_perp=perp, s_dlat=x.grad, s_demb1=m.embedding1.weight.grad,
                   s_demb2=m.embedding2.weight.grad, s_idx1=i1, s_idx2=i2)
        # --- VectorQuantizeEMA
        m = ref_vq.VectorQuantizeEMA(K, D, beta=0.25, decay=0.9)
        m.embedding.weight.data.copy_(ef)
        st = dict(embedding=ef.clone(), ema_cluster_size=torch.zeros(K), ema_w=torch.zeros(K, D))
        for step in range(3):
            m.train(step < 2)
            x = lats[step].clone().requires_grad_(True)
            q, loss, perp = m(x)
            (q * w).sum().add(loss).backward()
            oq, ol, op, idx, st = O.vq_ema_forward(st, lats[step], 0.25, 0.9, training=step < 2)
            close(oq, q, what=f"oracle ema quant {step}", tol=1e-5)
            close(ol, loss, what=f"oracle ema loss {step}", tol=1e-5)
            close(st["embedding"], m.embedding.weight.data, what=f"oracle ema codebook {step}", tol=1e-5)
            close(st["ema_cluster_size"], m.ema_cluster_size, what=f"oracle ema sizes {step}", tol=1e-5)
            out.update({f"e{step}_quant": q, f"e{step}_loss": loss, f"e{step}_perp": perp, f"e{step}_dlat": x.grad,
                        f"e{step}_idx": idx, f"e{step}_emb": m.embedding.weight.data.clone(),
                        f"e{step}_n": m.ema_cluster_size.clone(), f"e{step}_w": m.ema_w.clone()})
        assert m.embedding.weight.grad is None or float(m.embedding.weight.grad.abs().sum()) == 0.0
        # --- SlicedVectorQuantizeEMA
        m = ref_vq.SlicedVectorQuantizeEMA(K, D, beta=0.25, decay=0.9)
        m.embedding1.weight.data.copy_(e1)
        m.embedding2.weight.data.copy_(e2k)
        st = dict(embedding1=e1.clone(), embedding2=e2k.clone(), ema_cluster_size1=torch.zeros(K), ema_cluster_size2=torch.zeros(K),
                  ema_w1=torch.zeros(K, D // 2), ema_w2=torch.zeros(K, D // 2))
        for step in range(3):
            m.train(step < 2)
            x = lats[step].clone().requires_grad_(True)
            q, loss, perp = m(x)
            (q * w).sum().add(loss).backward()
            oq, ol, op, idxs, st = O.sliced_vq_ema_forward(st, lats[step], 0.25, 0.9, training=step < 2)
            close(oq, q, what=f"oracle sliced ema quant {step}", tol=1e-5)
            close(ol, loss, what=f"oracle sliced ema loss {step}", tol=1e-5)
            close(op, perp, what=f"oracle sliced ema perp {step}", tol=1e-5)
            close(st["embedding2"], m.embedding2.weight.data, what=f"oracle sliced ema codebook {step}", tol=1e-5)
            out.update({f"se{step}_quant": q, f"se{step}_loss": loss, f"se{step}_perp": perp, f"se{step}_dlat": x.grad,
                        f"se{step}_idx1": idxs[0], f"se{step}_idx2": idxs[1], f"se{step}_emb1": m.embedding1.weight.data.clone(),
                        f"se{step}_emb2": m.embedding2.weight.data.clone(), f"se{step}_n1": m.ema_cluster_size1.clone(),
                        f"se{step}_n2": m.ema_cluster_size2.clone(), f"se{step}_w1": m.ema_w1.clone(),
                        f"se{step}_w2": m.ema_w2.clone()})
    finally:
        torch.Tensor.cuda = keep
    save("quantizers", **out)


CFG_VQWAE = dict(name="vqwae", layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153,
                 upsample_scales=[4, 4, 8, 5], encoder_hid=256, c_in=39, K=256, cin_pad=0)
CFG_C5 = dict(name="c5", layers=48, stacks=4, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=153,
              upsample_scales=[4, 4, 8, 5], cin_pad=0)


CFG_C2 = dict(name="c2", layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153,
              upsample_scales=[4, 4, 4, 5], cin_pad=0)


def gen_c2():
    """BASELINE config C2 -- the geometry bench.py times (hps/inae_hp.json decoder dims: 24 layers / 2 stacks, R 256, G 368, S 256,
    Cc 64, Cg 64, scales 4,4,4,5) -- one clip of 8000 samples through the reference's own WaveNet: sparse logits probe, per-step
    log-sum-exp, sums; and ONE reference train step on that clip (teacher-forced CE with the shift of vqwae_train.py:764, autograd,
    no clipping needed): loss and every decoder parameter's gradient (squared norm + probe values), beside the fp64 oracle's."""
    cfg = CFG_C2
    sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
    T = 8000
    lat = O.hash_fill((1, cfg["Cc"], T // 320), 601, 1.2)
    x = ((O.hash_fill((1, T), 602) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    g = torch.tensor([77])
    wn = build_ref_wavenet(cfg)
    wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items()})
    wn.eval()
    ocfg = dict(layers=24, stacks=2, upsample_scales=cfg["upsample_scales"], cin_pad=0)
    with torch.no_grad():
        y = wn(xin, lat, g, False)
        oy = O.wavenet_forward(sd, ocfg, xin, lat, g)
    e = close(oy, y, what="C2 logits", tol=5e-5)
    print(f"  C2 (24 layers, G 368): T={T}, max|oracle-ref| = {e:.2e}, |y|max {y.abs().max().item():.3f}")
    ti = torch.unique(torch.cat([torch.arange(0, T, 53), torch.arange(4090, 4106), torch.arange(T - 4, T)]))
    # one reference train step (decoder only; the reference module in train mode, dropout 0.0)
    wn.train()
    for p_ in wn.parameters():
        p_.grad = None
    y_t = wn(xin, lat, g, False)
    crit = torch.nn.CrossEntropyLoss(reduction="none")
    ce = crit(y_t[:, :, :-1].unsqueeze(-1), x[:, 1:].unsqueeze(-1)).mean()
    ce.backward()
    grads = {"wavenet." + n: (p_.grad.clone() if p_.grad is not None else torch.zeros_like(p_)) for n, p_ in wn.named_parameters()}
    psd = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    oy64 = O.wavenet_forward(psd, ocfg, xin.double(), lat.double(), g)
    oce = O.masked_ce_loss(oy64, x.unsqueeze(-1), torch.tensor([T]))
    oce.backward()
    close(oce, ce, what="C2 train loss", tol=1e-5)
    ograds = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in psd.items()}
    worst = max((ograds[k].float() - grads[k]).abs().max().item() / (grads[k].abs().max().item() + 1e-20) for k in grads)
    print(f"  C2 train step: loss {ce.item():.5f}, worst fp32-reference vs fp64-oracle gradient deviation {worst:.2e} of a tensor's max")
    names = list(grads.keys())
    pidx = {k: _probe_index(grads[k].numel()) for k in names}
    cat = lambda d, dt: torch.cat([d[k].reshape(-1)[pidx[k]].to(dt) for k in names])   # noqa: E731
    save("model_c2_probe", cfg=json.dumps(cfg), salt=5, lat_salt=601, x_salt=602, g=g, T=T, probe_t=ti, y_probe=y[0][:, ti],
         y_lse=torch.logsumexp(y, 1), y_sum=y.double().sum(), y_abs_sum=y.double().abs().sum(),
         loss=ce.detach(), names=json.dumps(names), probe_counts=np.array([len(pidx[k]) for k in names]),
         grad_probe=cat(grads, torch.float32), grad64_probe=cat(ograds, torch.float64),
         grad64_sq=np.array([float((ograds[k] ** 2).sum()) for k in names]),
         grad_max=np.array([float(grads[k].abs().max()) for k in names]))


CFG_P = dict(name="P", layers=4, stacks=2, R=32, G=48, S=32, O=64, Cc=16, Cg=8, k=3, n_speakers=5,
             upsample_scales=[4, 4, 8, 5], encoder_hid=32, c_in=39, K=32, cin_pad=1)


def gen_cin_pad():
    """cin_pad = 1 (upsample.py:69-85: conv_in has k = 2*cin_pad + 1 taps and NO padding, so it eats cin_pad frames on either side;
    vqwae_train.py:455-478 crops the features cin_pad frames wider than the audio): the reference's VQVAE with
    upsample_params cin_pad=1 -- c_up, logits and, for the decoder alone on (B, Cc, Tc) features, logits and the input gradient."""
    cfg = CFG_P
    sd = O.make_state_dict(cfg, 4)
    assert sd["wavenet.upsample_net.conv_in.weight"].shape[-1] == 3
    B, F = 2, 16                                   # encoder: 16 frames -> 4 latent frames; conv_in leaves 2 -> T = 2 * 640
    c = O.hash_fill((B, cfg["c_in"], F), 41, 1.7)
    hop = int(np.prod(cfg["upsample_scales"]))
    wn = RefWaveNet(out_channels=cfg["O"], layers=cfg["layers"], stacks=cfg["stacks"], residual_channels=cfg["R"],
                    gate_channels=cfg["G"], skip_out_channels=cfg["S"], kernel_size=cfg["k"], dropout=0.0,
                    cin_channels=cfg["Cc"], gin_channels=cfg["Cg"], n_speakers=cfg["n_speakers"],
                    upsample_conditional_features=True, upsample_net="ConvInUpsampleNetwork",
                    upsample_params=dict(upsample_scales=cfg["upsample_scales"], cin_channels=cfg["Cc"], cin_pad=1),
                    scalar_input=False, use_speaker_embedding=True, output_distribution="Logistic", cin_pad=1)
    model = ref_vqvae.VQVAE(c_in=cfg["c_in"], hid=cfg["Cc"], K=cfg["K"], wavenet=wn, encoder_hid=cfg["encoder_hid"])
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.eval()
    with torch.no_grad():
        lat = model.encoder(c)
        quant, vq_loss, perp = model.vq(lat)
        c_up = model.wavenet.upsample_net(quant)
    T = c_up.shape[-1]
    assert T == (lat.shape[-1] - 2) * hop, (T, lat.shape)
    x = ((O.hash_fill((B, T), 42) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    g = torch.tensor([1, 3])
    with torch.no_grad():
        y_hat, vq2, perp2 = model(xin, c, g, False)
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=1)
    close(O.upsample_forward(sd, quant, cfg["upsample_scales"], cin_pad=1), c_up, what="cin_pad c_up")
    oy, ovq, operp, aux = O.vqvae_forward(sd, ocfg, xin, c, g)
    e = close(oy, y_hat, what="cin_pad logits")
    # decoder alone with a gradient flowing back into the (B, Cc, Tc) features through conv_in
    feats = O.hash_fill((B, cfg["Cc"], lat.shape[-1]), 43, 1.1).requires_grad_(True)
    y_dec = model.wavenet(xin, feats, g, False)
    wsum = O.hash_fill(tuple(y_dec.shape), 44, 1.0)
    (y_dec * wsum).sum().backward()
    print(f"  cin_pad=1: T={T} from {lat.shape[-1]} latent frames, logits max|oracle-ref| = {e:.2e}")
    save("model_P", cfg=json.dumps(cfg), salt=4, c=c, x=x.numpy(), g=g, latents=lat, quant=quant, vq_idx=aux["idx"], vq_loss=vq_loss,
         perp=perp, c_up=c_up, y_hat=y_hat, feats=feats.detach(), y_dec_probe=y_dec.detach()[:, :, ::7], dfeats=feats.grad, w_salt=44)


CFG_U = dict(name="U", layers=4, stacks=2, R=32, G=48, S=32, O=64, Cc=16, Cg=8, k=3, n_speakers=5,
             upsample_scales=[4, 4, 8, 5], cin_pad=1, conv_in=False)


def gen_plain_upsample():
    """upsample_net = "UpsampleNetwork" (upsample.py:29-66: the stages without conv_in, the output trimmed by cin_pad * prod(scales)
    samples at either end; state_dict keys `upsample_net.up_layers.N`): the reference's WaveNet on (B, Cc, Tc) features with
    cin_pad = 1 and cin_pad = 0 -- c_up, logits, and the gradient of a weighted logit sum with respect to the features."""
    out = {}
    for pad in (1, 0):
        cfg = dict(CFG_U, cin_pad=pad)
        sd = O.make_state_dict(cfg, 6, with_encoder=False)
        assert "wavenet.upsample_net.conv_in.weight" not in sd and "wavenet.upsample_net.up_layers.1.weight_v" in sd
        wn = RefWaveNet(out_channels=cfg["O"], layers=cfg["layers"], stacks=cfg["stacks"], residual_channels=cfg["R"],
                        gate_channels=cfg["G"], skip_out_channels=cfg["S"], kernel_size=cfg["k"], dropout=0.0,
                        cin_channels=cfg["Cc"], gin_channels=cfg["Cg"], n_speakers=cfg["n_speakers"],
                        upsample_conditional_features=True, upsample_net="UpsampleNetwork",
                        upsample_params=dict(upsample_scales=cfg["upsample_scales"], cin_channels=cfg["Cc"], cin_pad=pad),
                        scalar_input=False, use_speaker_embedding=True, output_distribution="Logistic", cin_pad=pad)
        missing = wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items()}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        wn.eval()
        B, Tc = 2, 4
        hop = int(np.prod(cfg["upsample_scales"]))
        T = (Tc - 2 * pad) * hop
        feats = O.hash_fill((B, cfg["Cc"], Tc), 61 + pad, 1.1).requires_grad_(True)
        x = ((O.hash_fill((B, T), 63 + pad) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
        xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
        g = torch.tensor([1, 3])
        with torch.no_grad():
            c_up = wn.upsample_net(feats)
        assert c_up.shape[-1] == T, (c_up.shape, T)
        y = wn(xin, feats, g, False)
        wsum = O.hash_fill(tuple(y.shape), 65 + pad, 1.0)
        (y * wsum).sum().backward()
        ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=pad, conv_in=False)
        close(O.upsample_forward(sd, feats.detach(), cfg["upsample_scales"], cin_pad=pad, conv_in=False), c_up, what="plain upsample c_up")
        e = close(O.wavenet_forward(sd, ocfg, xin, feats.detach(), g), y.detach(), what="plain upsample logits")
        print(f"  UpsampleNetwork cin_pad={pad}: T={T} from {Tc} frames, logits max|oracle-ref| = {e:.2e}")
        out.update({f"feats{pad}": feats.detach(), f"x{pad}": x.numpy(), f"c_up_probe{pad}": c_up[:, :, ::3], f"y_probe{pad}": y.detach()[:, :, ::5],
                    f"dfeats{pad}": feats.grad, f"w_salt{pad}": 65 + pad})
    save("model_U", cfg=json.dumps(CFG_U), salt=6, g=torch.tensor([1, 3]), **out)


def gen_upsample_activation():
    """upsample_activation (upsample.py:44-46: an element-wise torch.nn module behind every stage's FIR; the conv keys move to
    up_layers.{3 i + 1}): the reference's WaveNet with ConvInUpsampleNetwork + LeakyReLU(0.2) and with the plain UpsampleNetwork + Tanh --
    c_up, logits and the gradient of a weighted logit sum with respect to the features."""
    out = {}
    for tag, net, act, params, slope in (("leaky", "ConvInUpsampleNetwork", "LeakyReLU", {"negative_slope": 0.2}, 0.2),
                                         ("tanh", "UpsampleNetwork", "Tanh", {}, 0.01), ("relu", "ConvInUpsampleNetwork", "ReLU", {}, 0.01),
                                         ("sigm", "ConvInUpsampleNetwork", "Sigmoid", {}, 0.01)):
        cfg = dict(CFG_U, cin_pad=0, conv_in=net == "ConvInUpsampleNetwork", up_act=act, up_act_slope=slope)
        sd = O.make_state_dict(cfg, 8, with_encoder=False)
        wn = RefWaveNet(out_channels=cfg["O"], layers=cfg["layers"], stacks=cfg["stacks"], residual_channels=cfg["R"],
                        gate_channels=cfg["G"], skip_out_channels=cfg["S"], kernel_size=cfg["k"], dropout=0.0,
                        cin_channels=cfg["Cc"], gin_channels=cfg["Cg"], n_speakers=cfg["n_speakers"],
                        upsample_conditional_features=True, upsample_net=net,
                        upsample_params=dict(upsample_scales=cfg["upsample_scales"], cin_channels=cfg["Cc"], cin_pad=0,
                                             upsample_activation=act, upsample_activation_params=params),
                        scalar_input=False, use_speaker_embedding=True, output_distribution="Logistic", cin_pad=0)
        missing = wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items()}, strict=True)
        assert not missing.missing_keys and not missing.unexpected_keys
        wn.eval()
        B, Tc = 2, 2
        T = Tc * int(np.prod(cfg["upsample_scales"]))
        feats = O.hash_fill((B, cfg["Cc"], Tc), 71, 1.1).requires_grad_(True)
        x = ((O.hash_fill((B, T), 72) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
        xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
        g = torch.tensor([1, 3])
        with torch.no_grad():
            c_up = wn.upsample_net(feats)
        y = wn(xin, feats, g, False)
        wsum = O.hash_fill(tuple(y.shape), 73, 1.0)
        (y * wsum).sum().backward()
        ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0, conv_in=cfg["conv_in"],
                    up_act=act, up_act_slope=slope)
        close(O.upsample_forward(sd, feats.detach(), cfg["upsample_scales"], conv_in=cfg["conv_in"], act=act, act_slope=slope), c_up,
              what="upsample_activation c_up")
        e = close(O.wavenet_forward(sd, ocfg, xin, feats.detach(), g), y.detach(), what="upsample_activation logits")
        print(f"  upsample_activation {act} on {net}: logits max|oracle-ref| = {e:.2e}")
        out.update({f"cfg_{tag}": json.dumps(cfg), f"c_up_probe_{tag}": c_up[:, :, ::3], f"y_probe_{tag}": y.detach()[:, :, ::5],
                    f"dfeats_{tag}": feats.grad})
    save("model_V", salt=8, g=torch.tensor([1, 3]), feats=O.hash_fill((2, CFG_U["Cc"], 2), 71, 1.1), x=x.numpy(), w_salt=73, **out)


def _ref_sampler_class():
    """The reference's PartialyRandomizedSimilarTimeLengthSampler, cut out of vqwae_train.py by its syntax tree (the file itself
    cannot be imported: docopt / nnmnkwii / librosa / tensorboardX are absent) and executed as is."""
    import ast
    import random as _random
    from torch.utils.data.sampler import Sampler
    src = open(os.path.join(REF, "vqwae_train.py")).read()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.ClassDef) and n.name == "PartialyRandomizedSimilarTimeLengthSampler")
    ns = {"Sampler": Sampler, "torch": torch, "np": np, "random": _random}
    exec(compile(ast.Module(body=[node], type_ignores=[]), "vqwae_train.py", "exec"), ns)
    return ns["PartialyRandomizedSimilarTimeLengthSampler"], _random


def gen_sampler():
    """Orders the reference's own sampler yields (three epochs each, `random.seed(1234)` first): unique lengths for N = 37, 64, 203
    with batch sizes 8 / 4, and tied lengths (torch.sort is not stable, so only the LENGTH at every position is pinned there)."""
    cls, rnd = _ref_sampler_class()
    out = {}
    cases = []
    for i, (N, bs, tied) in enumerate(((37, 8, False), (64, 8, False), (203, 4, False), (120, 8, True))):
        rs = np.random.RandomState(100 + i)
        lengths = (rs.permutation(N) * 7 + 300) if not tied else rs.randint(300, 320, size=N)
        rnd.seed(1234)
        smp = cls(lengths.tolist(), batch_size=bs)
        orders = np.stack([np.array([int(j) for j in smp]) for _ in range(3)])
        assert all(sorted(o.tolist()) == list(range(N)) for o in orders)
        out[f"lengths{i}"] = lengths.astype(np.int64)
        out[f"orders{i}"] = orders.astype(np.int64)
        cases.append(dict(N=N, batch_size=bs, tied=tied))
    save("sampler_ref", cases=json.dumps(cases), **out)



class _Pick:
    """Stands in for torch.distributions.OneHotCategorical inside the reference's own incremental loop (wavenet.py:335-338):
    the draw becomes reproducible -- argmax, or the inverse CDF on explicit uniforms (the oracle's and the engine's form) --
    and every step's probability vector is recorded."""
    mode, uniforms, rec, step = "argmax", None, [], 0

    def __init__(self, probs):
        self.p = probs

    def sample(self):
        p, t = self.p, _Pick.step
        _Pick.step += 1
        _Pick.rec.append(p.clone())
        if _Pick.mode == "argmax":
            idx = p.argmax(1)
        else:
            cdf = p.double().cumsum(1)
            idx = (cdf < _Pick.uniforms[:, t:t + 1].double() * cdf[:, -1:]).sum(1).clamp(max=p.shape[1] - 1)
        return torch.nn.functional.one_hot(idx, p.shape[1]).float()

    @classmethod
    def start(cls, mode, uniforms=None):
        cls.mode, cls.uniforms, cls.rec, cls.step = mode, uniforms, [], 0


def gen_ar_c4():
    """BASELINE config C4: the synthesis decoder of hps/vqwae.json (20 layers, R = G = S = 256, dilations to 512) over T = 2560
    samples -- long enough that every layer's history ring wraps (2 * 512 + 1 entries) at least twice.  The reference's own
    incremental_forward: teacher-forced logits, a greedy roll-out, an inverse-CDF sampled roll-out and partial teacher forcing."""
    cfg = CFG_VQWAE
    sd = O.make_state_dict(dict(cfg), salt=7, with_encoder=False)
    T, O_ = 2560, cfg["O"]
    lat = O.hash_fill((1, cfg["Cc"], T // 640), 401, 1.2)
    x = ((O.hash_fill((1, T), 402) * 0.5 + 0.5) * O_).long().clamp(0, O_ - 1)
    xin = torch.nn.functional.one_hot(x, O_).float().transpose(1, 2).contiguous()
    g = torch.tensor([17])
    u = O.hash_fill((1, T), 403) * 0.5 + 0.5
    wn = build_ref_wavenet(cfg).eval()
    wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items()})
    wn.make_generation_fast_()
    import time
    keep = torch.distributions.OneHotCategorical
    out = {}
    try:
        torch.distributions.OneHotCategorical = _Pick
        with torch.no_grad():
            t0 = time.time()
            tf = wn.incremental_forward(None, c=lat, g=g, T=T, test_inputs=xin, softmax=False, quantize=False)
            print(f"  C4 teacher-forced: {T / (time.time() - t0):.0f} samples/s on the reference")
            fwd = wn(xin, lat, g, False)
            close(tf, fwd, what="C4 incremental == forward", tol=2e-5)
            _Pick.start("argmax")
            gr = wn.incremental_forward(None, c=lat, g=g, T=T, softmax=True, quantize=True)
            p_gr = torch.stack(_Pick.rec, dim=-1)                     # (1, O, T)
            _Pick.start("cdf", u)
            sm = wn.incremental_forward(None, c=lat, g=g, T=T, softmax=True, quantize=True)
            p_sm = torch.stack(_Pick.rec, dim=-1)
            nf, Tp = 700, 1280                                        # forced up to the middle of the d = 512 ring's 2nd lap, then free
            _Pick.start("argmax")
            pt = wn.incremental_forward(None, c=lat[:, :, :Tp // 640].contiguous(), g=g, T=Tp, softmax=True, quantize=True,
                                        test_inputs=xin[:, :, :nf].contiguous())
            p_pt = torch.stack(_Pick.rec, dim=-1)
    finally:
        torch.distributions.OneHotCategorical = keep
    greedy = gr.argmax(1)
    top2 = p_gr[:, :, :].topk(2, dim=1)[0].clamp_min(1e-30).log()
    margin = (top2[:, 0] - top2[:, 1])                                # logit gap between the two best classes, (1, T)
    sampled = sm.argmax(1)
    partial = pt.argmax(1)
    top2p = p_pt.topk(2, dim=1)[0].clamp_min(1e-30).log()
    partial_margin = top2p[:, 0] - top2p[:, 1]
    cdf = p_sm.double().cumsum(1)
    cdf_gap = (cdf / cdf[:, -1:] - u[:, None, :].double()).abs().min(1)[0]   # distance of the uniform from the nearest CDF step
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    # the oracle on a prefix (its loop is the slow part of this generator)
    Tq = 640
    c_up = O.upsample_forward(sd, lat, cfg["upsample_scales"])
    o_tf = O.incremental_forward(sd, ocfg, c_up[:, :, :Tq].contiguous(), g, Tq, test_inputs=xin[:, :, :Tq], mode="logits")
    close(o_tf, tf[:, :, :Tq], what="oracle C4 teacher-forced prefix", tol=2e-5)
    probe_t = torch.unique(torch.cat([torch.arange(0, T, 13), torch.arange(1016, 1040), torch.arange(2040, 2064),
                                      torch.arange(T - 8, T)]))
    save("ar_c4", cfg=json.dumps(cfg), salt=7, lat=lat, x=x.to(torch.uint8), g=g, probe_t=probe_t, tf_probe=tf[0][:, probe_t],
         tf_argmax=tf.argmax(1).to(torch.uint8), tf_lse=torch.logsumexp(tf, 1), greedy=greedy.to(torch.uint8),
         greedy_margin=margin, sampled=sampled.to(torch.uint8), cdf_gap=cdf_gap.float(), u_salt=403,
         partial=partial.to(torch.uint8), partial_margin=partial_margin, partial_forced=nf)
    print(f"  C4: min greedy logit margin {margin.min().item():.3e}, min cdf gap {cdf_gap.min().item():.3e}")


def _probe_index(n, m=24):
    """m reproducible positions of a flat tensor of n values: the first 4 and a stride walk"""
    idx = np.unique(np.concatenate([np.arange(min(4, n)), (np.arange(m, dtype=np.int64) * 2654435761 + 12345) % n]))
    return torch.from_numpy(idx)


def gen_train_vqwae():
    """BASELINE configs C1 / C3 geometry: hps/vqwae.json in full (302 tensors, 7 555 218 values), B = 2 clips x 5120 samples,
    one complete reference train step (vqwae_train.py:709-798): loss terms, every parameter's gradient (squared norm + probe
    values), clip norm, post-Adam parameters and EMA shadow at the probes."""
    cfg = CFG_VQWAE
    sd = O.make_state_dict(dict(cfg), 7)
    c, x, xin, g, T = inputs_for(cfg, 2, 32, 310)
    assert T == 5120
    lengths = torch.tensor([T, T - 777])
    y = x.unsqueeze(-1)
    model = build_ref_vqvae(cfg, {k: v.clone() for k, v in sd.items()}).train()
    opt = torch.optim.Adam(model.parameters(), lr=4e-4, eps=1e-8, weight_decay=0.0)
    shadow = {n: p.data.clone() for n, p in model.named_parameters()}
    mask = O.sequence_mask(lengths, T).unsqueeze(-1)[:, 1:, :]
    opt.zero_grad()
    y_hat, vq_loss, perp = model(xin, c, g, False)
    crit = torch.nn.CrossEntropyLoss(reduction="none")
    ce = ((crit(y_hat[:, :, :-1].unsqueeze(-1), y[:, 1:, :]) * mask).sum()) / mask.sum()
    loss = ce + vq_loss.mean()
    loss.backward()
    grads = {n: (p.grad.clone() if p.grad is not None else torch.zeros_like(p)) for n, p in model.named_parameters()}
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 100.0)
    opt.step()
    for n, p in model.named_parameters():
        shadow[n] -= (1.0 - 0.9999) * (shadow[n] - p.data)
    new = {n: p.data for n, p in model.named_parameters()}
    # fp64 oracle of the same step: the tight yardstick for the engine's fp32 gradients
    psd = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    oy, ovq, operp, aux = O.vqvae_forward(psd, ocfg, xin.double(), c.double(), g)
    oloss = O.masked_ce_loss(oy, y, lengths) + ovq
    oloss.backward()
    close(oloss, loss, what="vqwae train loss", tol=1e-5)
    ograds = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in psd.items()}
    worst = 0.0
    for k in grads:
        e = (ograds[k].float() - grads[k]).abs().max().item() / (grads[k].abs().max().item() + 1e-20)
        worst = max(worst, e)
        assert e < 5e-3, (k, e)
    print(f"  vqwae.json train step: loss {loss.item():.5f}, |g| {float(gn):.4f}, worst fp32-reference vs fp64-oracle gradient "
          f"deviation {worst:.2e} of a tensor's max")
    names = list(grads.keys())
    pidx = {k: _probe_index(grads[k].numel()) for k in names}
    cat = lambda d, dt: torch.cat([d[k].reshape(-1)[pidx[k]].to(dt) for k in names])   # noqa: E731
    save("train_vqwae", cfg=json.dumps(cfg), salt=7, in_salt=310, lengths=lengths, loss=loss.detach(), ce=ce.detach(),
         vq_loss=vq_loss.detach(), perp=perp.detach(), grad_norm=gn, vq_idx=aux["idx"],
         names=json.dumps(names), probe_counts=np.array([len(pidx[k]) for k in names]),
         grad_probe=cat(grads, torch.float32), grad64_probe=cat(ograds, torch.float64), new_probe=cat(new, torch.float32),
         ema_probe=cat(shadow, torch.float32),
         grad_sq=np.array([float((grads[k].double() ** 2).sum()) for k in names]),
         grad64_sq=np.array([float((ograds[k] ** 2).sum()) for k in names]),
         grad_max=np.array([float(grads[k].abs().max()) for k in names]))


def gen_c5_probe():
    """BASELINE config C5 at FULL depth: 48 layers in 4 stacks (dilations to 2048, receptive field 32 761), R = G = S = 512,
    one clip of 5120 samples through the reference's WaveNet with its upsampling network: sparse logits probe + sums.
    (T = 5120 > 2 * 2048: both history taps of the widest layers reach real samples.)"""
    cfg = CFG_C5
    sd = O.make_state_dict(dict(cfg), salt=9, with_encoder=False)
    T = 5120
    lat = O.hash_fill((1, cfg["Cc"], T // 640), 501, 1.2)
    x = ((O.hash_fill((1, T), 502) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    g = torch.tensor([41])
    wn = build_ref_wavenet(cfg).eval()
    wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items()})
    with torch.no_grad():
        y = wn(xin, lat, g, False)
        ocfg = dict(layers=48, stacks=4, upsample_scales=cfg["upsample_scales"], cin_pad=0)
        oy = O.wavenet_forward(sd, ocfg, xin, lat, g)
    e = close(oy, y, what="C5 logits", tol=5e-5)
    print(f"  C5 full depth: T={T}, max|oracle-ref| = {e:.2e}, |y|max {y.abs().max().item():.3f}")
    ti = torch.unique(torch.cat([torch.arange(0, T, 37), torch.arange(4090, 4110), torch.arange(T - 4, T)]))
    save("model_c5_probe", cfg=json.dumps(cfg), salt=9, lat_salt=501, x_salt=502, g=g, T=T, probe_t=ti, y_probe=y[0][:, ti],
         y_lse=torch.logsumexp(y, 1), y_sum=y.double().sum(), y_abs_sum=y.double().abs().sum())


def gen_optim_state(cfg, salt):
    """Section 8(f) rank 2: what a reference checkpoint's "optimizer" entry holds (vqwae_train.py:881) -- torch.optim.Adam's
    state after one and after two steps of the reference model on one batch, flattened in named_parameters() order."""
    sd = O.make_state_dict(cfg, salt)
    c, x, xin, g, T = inputs_for(cfg, 2, 8, salt * 10)
    model = build_ref_vqvae(cfg, {k: v.clone() for k, v in sd.items()}).train()
    opt = torch.optim.Adam(model.parameters(), lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False)
    names = [n for n, _ in model.named_parameters()]
    lengths = torch.tensor([T, T])
    y = x.unsqueeze(-1)
    mask = O.sequence_mask(lengths, T).unsqueeze(-1)[:, 1:, :]
    crit = torch.nn.CrossEntropyLoss(reduction="none")
    out = {}

    def flat(d):
        return torch.cat([d[n].reshape(-1) for n in names])

    for step in (1, 2):
        opt.zero_grad()
        y_hat, vq_loss, perp = model(xin, c, g, False)
        loss = ((crit(y_hat[:, :, :-1].unsqueeze(-1), y[:, 1:, :]) * mask).sum()) / mask.sum() + vq_loss.mean()
        loss.backward()
        for p_ in model.parameters():
            if p_.grad is None:
                p_.grad = torch.zeros_like(p_)
        torch.nn.utils.clip_grad_norm_(model.parameters(), 100.0)
        opt.step()
        st = opt.state_dict()
        assert st["param_groups"][0]["params"] == list(range(len(names)))
        out[f"params{step}"] = flat({n: p_.data for n, p_ in model.named_parameters()})
        out[f"exp_avg{step}"] = flat({n: st["state"][i]["exp_avg"] for i, n in enumerate(names)})
        out[f"exp_avg_sq{step}"] = flat({n: st["state"][i]["exp_avg_sq"] for i, n in enumerate(names)})
        out[f"step{step}"] = np.array([float(st["state"][i]["step"]) for i in range(len(names))])
        out[f"loss{step}"] = loss.detach()
    group = {k: v for k, v in st["param_groups"][0].items() if k != "params"}
    save(f"optim_{cfg['name']}", names=json.dumps(names), param_group=json.dumps(group), **out)


def main():
    if sys.argv[1:] == ["optim"]:
        return gen_optim_state(CFG_A, 1)
    if sys.argv[1:] == ["c4"]:
        return gen_ar_c4()
    if sys.argv[1:] == ["train_vqwae"]:
        return gen_train_vqwae()
    if sys.argv[1:] == ["c5"]:
        return gen_c5_probe()
    if sys.argv[1:] == ["quantizers"]:
        return gen_quantizers()
    if sys.argv[1:] == ["wide"]:
        return gen_wide_probe()
    if sys.argv[1:] == ["c2"]:
        return gen_c2()
    if sys.argv[1:] == ["cin_pad"]:
        return gen_cin_pad()
    if sys.argv[1:] == ["plain_upsample"]:
        return gen_plain_upsample()
    if sys.argv[1:] == ["upsample_activation"]:
        return gen_upsample_activation()
    if sys.argv[1:] == ["sampler"]:
        return gen_sampler()
    if sys.argv[1:] == ["noncausal"]:
        return gen_noncausal_layer()
    gen_quantizers()
    gen_misc()
    gen_dmol()
    for cfg, salt in ((CFG_A, 1), (CFG_B, 2)):
        sd, model, ins, ocfg = gen_model(cfg, salt)
        gen_layers(cfg, sd, model, ins)
        gen_losses(cfg, sd, model, ins, ocfg)
        gen_ar(cfg, sd, model, ins, ocfg)
        if cfg is CFG_A:
            gen_train_step(cfg, sd, ins, ocfg)
            gen_optim_state(cfg, salt)
    # scalar-input (DMoL) decoder
    sd, model, ins, ocfg = gen_model(CFG_S, 3)
    gen_ar_scalar(CFG_S, sd, model, ins, ocfg)
    gen_vqwae_probe()
    gen_wide_probe()
    gen_ar_c4()
    gen_train_vqwae()
    gen_c5_probe()
    gen_c2()
    gen_cin_pad()
    gen_plain_upsample()
    gen_upsample_activation()
    gen_sampler()
    gen_noncausal_layer()


if __name__ == "__main__":
    main()
