"""GPU parity of the fp16 storage mode (WAE_F16; BASELINE config C5 names "fp16 + MFMA 1x1"): same kernels, packing and
fragment geometry as bf16 with v_mfma_f32_32x32x16_f16; the backward pass runs on loss-scaled 16-bit gradients
(engine.grad_scale) and hands unscaled fp32 weight gradients to the optimizer.

Stated tolerances (norm-wise: max|a-b| / max|b|): logits 1e-2 (fp16 keeps 11 significand bits against bf16's 8, whose bound is
5e-2), parameter gradients 4e-2 of each tensor's range (bf16: 8e-2)."""
import json

import numpy as np
import pytest
import torch

from helpers import golden_model, load_npz, rel_err
from oracle import wae_oracle as O

pytestmark = pytest.mark.gpu


def _engine(cfg, sd, dtype="fp16"):
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    eng.load_state_dict(sd)
    return eng


@pytest.mark.parametrize("name", ["A", "B"])
def test_forward_logits_latents_indices(name):
    cfg, sd, ins, z, ocfg = golden_model(name)
    eng = _engine(cfg, sd)
    out = eng.forward(ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda(), targets=ins["x"].cuda())
    torch.cuda.synchronize()
    assert np.array_equal(out["idx"].cpu().numpy(), z["vq_idx"])            # the front end is fp32 in every mode
    assert rel_err(out["latents"].cpu(), z["latents"]) < 1e-3
    assert rel_err(out["logits"].cpu(), z["y_hat"]) < 1e-2
    ce = O.masked_ce_loss(torch.from_numpy(z["y_hat"]), ins["x"].unsqueeze(-1), torch.full((2,), ins["x"].shape[1]))
    assert abs(float(out["loss"]) - float(ce)) < 5e-3 * float(ce)


def test_wide_head_and_incremental_decode():
    z = load_npz("model_wide_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    eng = _engine(cfg, sd)
    B, T = 2, int(z["T"])
    x = ((O.hash_fill((B, T), int(z["x_salt"])) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    c = O.hash_fill((B, cfg["Cc"], T), int(z["c_salt"]), 1.3)
    out = eng.decoder_forward(x.cuda(), c.cuda(), torch.from_numpy(z["g"]).cuda(), c_is_upsampled=True)
    torch.cuda.synchronize()
    pt = torch.from_numpy(z["probe_t"]).long()
    assert rel_err(out["logits"].cpu()[:, :, pt], z["y_probe"]) < 1e-2
    # autoregressive kernels (teacher-forced == the batch forward of the same engine), both the one-CU and the cooperative one
    cfg2, sd2, ins, zm, ocfg = golden_model("A")
    za = load_npz("ar_A")
    e2 = _engine(cfg2, sd2)
    c_up = torch.from_numpy(za["c_up"]).cuda()
    Tar = c_up.shape[-1]
    ar = e2.incremental_forward(c_up, ins["g"].cuda(), Tar, mode="logits", test_inputs=ins["x"][:, :Tar].cuda(), c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(ar["logits"].cpu(), za["tf_logits"]) < 1e-2


def test_c4_teacher_forced_fp16():
    z = load_npz("ar_c4")
    cfg = {k: v for k, v in json.loads(str(z["cfg"])).items() if k not in ("encoder_hid", "c_in", "K")}
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    eng = _engine(cfg, sd)
    T = z["x"].shape[1]
    out = eng.incremental_forward(torch.from_numpy(z["lat"]).cuda(), torch.from_numpy(z["g"]).cuda(), T, mode="logits",
                                  test_inputs=torch.from_numpy(z["x"].astype(np.int64)).cuda())
    torch.cuda.synchronize()
    pt = torch.from_numpy(z["probe_t"]).long()
    assert rel_err(out["logits"].cpu()[0][:, pt], z["tf_probe"]) < 1e-2


def test_decoder_backward_gradients():
    """loss-scaled 16-bit backward: every decoder parameter gradient against autograd through the fp32 oracle"""
    from wavenet_autoencoders_amd import backward as BW
    cfg, sd, ins, z, ocfg = golden_model("A")
    eng = _engine(cfg, sd)
    assert eng.grad_scale == 4096.0
    x, g = ins["x"], ins["g"]
    T = x.shape[1]
    c_up = torch.from_numpy(z["c_up"])
    lengths = torch.tensor([T, T - 137])
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("wavenet.") and "upsample_net" not in k}
    cl = c_up.clone().requires_grad_(True)
    y = O.wavenet_forward(psd, dict(ocfg, upsample_scales=None), ins["xin"], cl, g)
    loss = O.masked_ce_loss(y, x.unsqueeze(-1), lengths)
    loss.backward()
    out = eng.decoder_forward(x.cuda(), c_up.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), train=True,
                              c_is_upsampled=True, want_logits=False)
    BW.decoder_backward(eng, x.cuda(), x.cuda(), lengths, g.cuda())
    grads = BW.finish_grads(eng)
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(loss.detach())) < 5e-3 * float(loss.detach())
    bad = {}
    for k, v in psd.items():
        gref = v.grad if v.grad is not None else torch.zeros_like(v)
        got = grads[eng.lay.off(k):eng.lay.off(k) + eng.lay.numel(k)].view(eng.lay.shapes[k]).cpu()
        err, ref = float((got - gref).abs().max()), float(gref.abs().max())
        if err > 4e-2 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad


def test_full_train_step_tracks_fp32():
    """VQVAE train steps (encoder, VQ, upsampling, decoder, clip + Adam + EMA): the fp16 engine's losses follow the fp32
    engine's over several updates of the same batch, gradients norms agree, nothing overflows."""
    cfg, sd, ins, z, ocfg = golden_model("A")
    x, c, g = ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda()
    hist = {}
    for dt in ("fp32", "fp16"):
        eng = _engine(cfg, sd, dt)
        eng.init_optimizer()
        hist[dt] = [(float(r["ce"]), float(r["grad_norm"])) for r in (eng.train_step(x, c, g, lr=1e-3) for _ in range(5))]
        torch.cuda.synchronize()
        assert bool(torch.isfinite(eng.params).all())
    for (l32, n32), (l16, n16) in zip(hist["fp32"], hist["fp16"]):
        assert abs(l16 - l32) < 1e-2 * l32 and abs(n16 - n32) < 3e-2 * n32, (hist["fp32"], hist["fp16"])
    assert hist["fp16"][-1][0] < hist["fp16"][0][0]


def test_c5_shard_forward_backward_fp16():
    """BASELINE config C5 as named: 48 layers / 4 stacks, R = G = S = 512, fp16; one clip against the reference's logits, then a
    train step of a 4 x 5120 slice of the shard with finite, fp32-consistent loss."""
    z = load_npz("model_c5_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    eng = _engine(cfg, sd)
    T = int(z["T"])
    lat = O.hash_fill((1, cfg["Cc"], T // 640), int(z["lat_salt"]), 1.2)
    x = ((O.hash_fill((1, T), int(z["x_salt"])) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    out = eng.decoder_forward(x.cuda(), lat.cuda(), torch.from_numpy(z["g"]).cuda())
    torch.cuda.synchronize()
    pt = torch.from_numpy(z["probe_t"]).long()
    assert rel_err(out["logits"].cpu()[0][:, pt], z["y_probe"]) < 1e-2
    rng = np.random.default_rng(5)
    B = 4
    xb = torch.from_numpy(rng.integers(0, 256, size=(B, T))).cuda()
    lb = torch.from_numpy(rng.standard_normal((B, cfg["Cc"], T // 640)).astype(np.float32)).cuda()
    gb = torch.from_numpy(rng.integers(0, cfg["n_speakers"], size=(B,))).cuda()
    eng.init_optimizer()
    r16 = eng.train_step(xb, lb, gb, lr=1e-4)
    torch.cuda.synchronize()
    l16, n16 = float(r16["ce"]), float(r16["grad_norm"])
    del eng
    e32 = _engine(cfg, sd, "fp32")
    e32.init_optimizer()
    r32 = e32.train_step(xb, lb, gb, lr=1e-4)
    torch.cuda.synchronize()
    assert np.isfinite(l16) and abs(l16 - float(r32["ce"])) < 5e-3 * float(r32["ce"])
    assert abs(n16 - float(r32["grad_norm"])) < 5e-2 * float(r32["grad_norm"])
