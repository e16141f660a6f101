"""Pins the CPU oracle against vectors produced by the reference itself
(tests/golden/make_golden.py imported /root/reference to make them)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import wae_oracle as O
from helpers import GOLDEN, golden_model, load_npz, rel_err

TOL = 2e-5


@pytest.mark.parametrize("name", ["A", "B", "S"])
def test_model_forward_matches_reference(name):
    cfg, sd, ins, z, ocfg = golden_model(name)
    lat = O.encoder_forward(sd, ins["c"])
    assert rel_err(lat, z["latents"]) < TOL
    q, vq_loss, perp, idx = O.vq_forward(sd["vq.embedding.weight"], lat)
    assert np.array_equal(idx.numpy(), z["vq_idx"])                      # bit-exact
    assert rel_err(q, z["quant"]) < TOL
    assert abs(float(vq_loss) - float(z["vq_loss"])) < 1e-6 * max(1, abs(float(z["vq_loss"])))
    assert abs(float(perp) - float(z["perp"])) < 1e-4
    assert rel_err(O.upsample_forward(sd, q, cfg["upsample_scales"]), z["c_up"]) < TOL
    y, _, _, _ = O.vqvae_forward(sd, ocfg, ins["xin"], ins["c"], ins["g"])
    assert rel_err(y, z["y_hat"]) < TOL
    ys = O.wavenet_forward(sd, ocfg, ins["xin"], q, ins["g"], softmax=True)
    assert rel_err(ys[:, :, ::37], z["y_softmax_probe"]) < TOL


def test_wrong_cond_length_raises():
    cfg, sd, ins, z, ocfg = golden_model("A")
    q = torch.from_numpy(z["quant"])
    with pytest.raises(Exception):
        O.wavenet_forward(sd, ocfg, ins["xin"][:, :, :-3], q, ins["g"])


@pytest.mark.parametrize("name", ["A", "B"])
def test_glu_layer(name):
    cfg, sd, ins, zm, ocfg = golden_model(name)
    z = load_npz("glu_" + name)
    T, B = int(z["T"]), 2
    x = O.hash_fill((B, cfg["R"], T), int(z["x_salt"]), float(z["x_scale"]))
    c = O.hash_fill((B, cfg["Cc"], T), int(z["c_salt"]), float(z["c_scale"]))
    g = O.hash_fill((B, cfg["Cg"], 1), int(z["g_salt"]), float(z["g_scale"]))
    pt = torch.from_numpy(z["probe_t"])
    for d in (1, 2, 512):
        for tag, (cc, gg) in dict(cg=(c, g), none=(None, None)).items():
            xo, so = O.glu_layer_forward(sd, "wavenet.conv_layers.1.", x, cc, gg, d)
            assert rel_err(xo[:, :, pt], z[f"xo_d{d}_{tag}"]) < TOL
            assert rel_err(so[:, :, pt], z[f"so_d{d}_{tag}"]) < TOL


def test_glu_layer_non_causal():
    """ResidualConv1dGLU(causal=False) of the reference (modules.py:82-88): the oracle's forward and its autograd gradients against the
    reference's (tests/golden/glu_noncausal.npz)."""
    cfg, sd, ins, zm, ocfg = golden_model("A")
    z = load_npz("glu_noncausal")
    pre = "wavenet.conv_layers.1."
    B, T, sc = int(z["B"]), int(z["T"]), float(z["scale"])
    x, c = O.hash_fill((B, cfg["R"], T), int(z["x_salt"]), sc), O.hash_fill((B, cfg["Cc"], T), int(z["c_salt"]), sc)
    g = O.hash_fill((B, cfg["Cg"], 1), int(z["g_salt"]), sc)
    wx, wsk = O.hash_fill((B, cfg["R"], T), int(z["wx_salt"])), O.hash_fill((B, cfg["S"], T), int(z["ws_salt"]))
    for d in (1, 8):
        psd = {k: (v.clone().requires_grad_(True) if k.startswith(pre) else v) for k, v in sd.items()}
        xr, cr = x.clone().requires_grad_(True), c.clone().requires_grad_(True)
        xo, so = O.glu_layer_forward(psd, pre, xr, cr, g.expand(-1, -1, T), d, causal=False)
        assert rel_err(xo.detach(), z[f"xo_d{d}"]) < TOL and rel_err(so.detach(), z[f"so_d{d}"]) < TOL
        ((xo * wx).sum() + (so * wsk).sum()).backward()
        assert rel_err(xr.grad, z[f"dx_d{d}"]) < 1e-4 and rel_err(cr.grad, z[f"dc_d{d}"]) < 1e-4
        for k in [k for k in z if k.startswith(f"grad_d{d}:")]:
            assert rel_err(psd[pre + k.split(":", 1)[1]].grad, z[k]) < 1e-4, k


@pytest.mark.parametrize("name", ["A", "B"])
def test_masked_ce(name):
    cfg, sd, ins, zm, ocfg = golden_model(name)
    z = load_npz("ce_" + name)
    y_hat = torch.from_numpy(zm["y_hat"]).requires_grad_(True)
    loss = O.masked_ce_loss(y_hat, ins["x"].unsqueeze(-1), torch.from_numpy(z["lengths"]))
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 1e-6
    assert rel_err(y_hat.grad[:, :, ::29], z["dlogits_probe"]) < 1e-5


def test_dmol_loss_and_sampler():
    z = load_npz("dmol")
    y_hat, y = torch.from_numpy(z["y_hat"]), torch.from_numpy(z["y"])
    for lsm in (7, 9):
        yh = y_hat.clone().requires_grad_(True)
        el = O.dmol_loss(yh, y, 256, -float(lsm), reduce=False)
        assert rel_err(el, z[f"loss_el_{lsm}"]) < TOL
        s = O.dmol_loss(yh, y, 256, -float(lsm), reduce=True)
        s.backward()
        assert abs(float(s) - float(z[f"loss_sum_{lsm}"])) < 1e-3
        assert rel_err(yh.grad, z[f"grad_{lsm}"]) < 1e-4
    assert rel_err(O.dmol_loss(y_hat, y, 65536, -16.0, reduce=False), z["loss_el_65536"]) < TOL
    smp = O.dmol_sample(y_hat, torch.from_numpy(z["u_mix"]), torch.from_numpy(z["u_log"]), -7.0)
    assert rel_err(smp, z["sample"]) < 1e-6


@pytest.mark.parametrize("name", ["A", "B"])
def test_incremental_forward(name):
    cfg, sd, ins, zm, ocfg = golden_model(name)
    z = load_npz("ar_" + name)
    c_up = torch.from_numpy(z["c_up"])
    Tar = c_up.shape[-1]
    tf = O.incremental_forward(sd, ocfg, c_up, ins["g"], Tar, test_inputs=ins["xin"][:, :, :Tar], mode="logits")
    assert rel_err(tf, z["tf_logits"]) < TOL
    # known-answer property from SURVEY section 4: incremental == batch forward
    sd2 = {k: v for k, v in sd.items() if "upsample_net" not in k}
    fwd = O.wavenet_forward(sd2, dict(ocfg, upsample_scales=None), ins["xin"][:, :, :Tar], c_up, ins["g"])
    assert rel_err(tf, fwd) < TOL
    gr = O.incremental_forward(sd, ocfg, c_up[:, :, :24].contiguous(), ins["g"], 24,
                               initial_input=torch.from_numpy(z["init"]), mode="argmax")
    assert np.array_equal(gr.argmax(1).numpy(), z["greedy"])


def test_incremental_forward_scalar_input():
    """Scalar-input decoder (wavenet.py:284-285,325-333): mixture parameters of every step under teacher forcing and a
    free-running roll-out on explicit uniforms, against the reference's own incremental loop (tests/golden/make_golden.py)."""
    cfg, sd, ins, zm, ocfg = golden_model("S")
    z = load_npz("ar_S")
    c_up = torch.from_numpy(z["c_up"])
    Tar = c_up.shape[-1]
    tf = O.incremental_forward(sd, ocfg, c_up, ins["g"], Tar, test_inputs=ins["xin"][:, :, :Tar], mode="logits")
    assert rel_err(tf, z["params_tf"]) < TOL
    roll = O.incremental_forward(sd, ocfg, c_up[:, :, :24].contiguous(), ins["g"], 24, mode="sample",
                                 u_mix=torch.from_numpy(z["u_mix"]), u_log=torch.from_numpy(z["u_log"]), log_scale_min=-7.0)
    assert rel_err(roll, z["roll"]) < 1e-5


def test_train_step_grads_adam_ema():
    cfg, sd, ins, zm, ocfg = golden_model("A")
    z = load_npz("train_A")
    T = ins["x"].shape[-1]
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y, vq, perp, _ = O.vqvae_forward(psd, ocfg, ins["xin"], ins["c"], ins["g"])
    ce = O.masked_ce_loss(y, ins["x"].unsqueeze(-1), torch.tensor([T, T]))
    loss = ce + vq
    loss.backward()
    assert abs(float(loss) - float(z["loss"])) < 1e-5
    assert abs(float(perp) - float(z["perp"])) < 1e-4
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in psd.items()}
    gsq = json.loads(str(z["grad_sq_by_key"]))
    for k, v in gsq.items():
        assert abs(float((grads[k].double() ** 2).sum()) - v) <= 1e-3 * max(v, 1e-9) + 1e-12, k
    params = {k: v.detach().clone() for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    vv = {k: torch.zeros_like(v) for k, v in params.items()}
    sh = {k: v.clone() for k, v in params.items()}
    gn = O.clip_adam_ema_step(params, grads, m, vv, sh, 1, 4e-4)
    assert abs(float(gn) - float(z["grad_norm"])) < 1e-4 * float(z["grad_norm"])
    for key in [k[5:] for k in z if k.startswith("grad:")]:
        assert rel_err(grads[key], z["grad:" + key]) < 2e-4, key
        assert rel_err(params[key], z["new:" + key]) < 1e-5, key
        assert rel_err(sh[key], z["ema:" + key]) < 1e-6, key


def test_vqwae_fullsize_probe():
    z = load_npz("model_vqwae_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(cfg, int(z["salt"]))
    shapes = json.loads(str(z["keys"]))
    assert {k: list(v.shape) for k, v in sd.items()} == shapes and len(sd) == 302
    import sys
    sys.path.insert(0, os.path.join(GOLDEN))
    hop = 640
    c = O.hash_fill((1, 39, 16), 71, 1.7)
    T = 4 * hop
    x = ((O.hash_fill((1, T), 72) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    g = ((O.hash_fill((1,), 73) * 0.5 + 0.5) * 153).long().clamp(0, 152)
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    ocfg = dict(layers=20, stacks=2, upsample_scales=cfg["upsample_scales"], cin_pad=0)
    with torch.no_grad():
        y, vq, perp, aux = O.vqvae_forward(sd, ocfg, xin, c, g)
    assert np.array_equal(aux["idx"].numpy(), z["vq_idx"])
    assert rel_err(y[0][:, torch.from_numpy(z["probe_t"])], z["y_probe"]) < 5e-5
    assert abs(float(y.double().sum()) - float(z["y_sum"])) < 1e-3 * float(z["y_abs_sum"])


def test_misc_schedules_and_receptive_field():
    with open(os.path.join(GOLDEN, "misc.json")) as fh:
        m = json.load(fh)
    for s, lr in zip(m["lr_steps"], m["step_lr"]):
        assert O.step_learning_rate_decay(4e-4, s, 0.5, 400000) == lr
    for key, rf in m["receptive_field"].items():
        L, s, k = map(int, key.split("_"))
        assert O.receptive_field_size(L, s, k) == rf
    assert m["receptive_field"]["20_2_3"] == 4093


def test_mulaw_hand_values():
    # hand-computed from the mu-law formula (nnmnkwii absent: "unpinned by import")
    assert int(O.mulaw_quantize(0.0, 255)) == 127
    assert int(O.mulaw_quantize(1.0, 255)) == 255
    assert int(O.mulaw_quantize(-1.0, 255)) == 0
    x = np.array([-0.5, -0.01, 0.01, 0.5])
    q = O.mulaw_quantize(x, 255)
    assert q.tolist() == [15, 98, 156, 239]   # e.g. (1-ln(128.5)/ln(256))/2*255 = 15.84 -> 15
    back = O.inv_mulaw_quantize(q, 255)
    assert np.all(np.abs(back - x) < 0.03)


def test_sliced_and_ema_quantizers_against_reference_vectors():
    """vector_quantization.py:51-306 (reference classes run on CPU by make_golden.gen_quantizers)."""
    z = load_npz("quantizers")
    t = {k: torch.from_numpy(np.asarray(v)) for k, v in z.items()}
    q, loss, perp, (i1, i2) = O.sliced_vq_forward(t["e1"], t["e2"], t["lats"][0], 0.25)
    assert torch.equal(i1, t["s_idx1"]) and torch.equal(i2, t["s_idx2"])
    assert rel_err(q, t["s_quant"]) < 1e-6 and abs(float(loss) - float(t["s_loss"])) < 1e-6
    assert abs(float(perp) - float(t["s_perp"])) < 1e-4
    K, D = t["ef"].shape
    st = dict(embedding=t["ef"].clone(), ema_cluster_size=torch.zeros(K), ema_w=torch.zeros(K, D))
    for step in range(3):
        q, loss, perp, idx, st = O.vq_ema_forward(st, t["lats"][step], 0.25, 0.9, training=step < 2)
        assert torch.equal(idx, t[f"e{step}_idx"])
        assert rel_err(q, t[f"e{step}_quant"]) < 1e-5 and abs(float(loss) - float(t[f"e{step}_loss"])) < 1e-6
        assert rel_err(st["embedding"], t[f"e{step}_emb"]) < 1e-5 and rel_err(st["ema_cluster_size"], t[f"e{step}_n"]) < 1e-5
        assert rel_err(st["ema_w"], t[f"e{step}_w"]) < 1e-5
    st = dict(embedding1=t["e1"].clone(), embedding2=t["e2k"].clone(), ema_cluster_size1=torch.zeros(K),
              ema_cluster_size2=torch.zeros(K), ema_w1=torch.zeros(K, D // 2), ema_w2=torch.zeros(K, D // 2))
    for step in range(3):
        q, loss, perp, idxs, st = O.sliced_vq_ema_forward(st, t["lats"][step], 0.25, 0.9, training=step < 2)
        assert torch.equal(idxs[0], t[f"se{step}_idx1"]) and torch.equal(idxs[1], t[f"se{step}_idx2"])
        assert rel_err(q, t[f"se{step}_quant"]) < 1e-5 and abs(float(perp) - float(t[f"se{step}_perp"])) < 1e-4
        for s in "12":
            assert rel_err(st["embedding" + s], t[f"se{step}_emb{s}"]) < 1e-5
            assert rel_err(st["ema_cluster_size" + s], t[f"se{step}_n{s}"]) < 1e-5


def test_wide_probe():
    """BASELINE config C5's widths (R = G = S = 512) on a short stack: the oracle against the reference's own WaveNet
    (tests/golden/model_wide_probe.npz; inputs are closed-form fills regenerated here from their salts)."""
    z = load_npz("model_wide_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    B, T = 2, int(z["T"])
    x = ((O.hash_fill((B, T), int(z["x_salt"])) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    c = O.hash_fill((B, cfg["Cc"], T), int(z["c_salt"]), 1.3)
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    with torch.no_grad():
        y = O.wavenet_forward(sd, dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=None, cin_pad=0), xin, c,
                              torch.from_numpy(z["g"]))
    assert rel_err(y[:, :, torch.from_numpy(z["probe_t"])], z["y_probe"]) < TOL
    assert abs(float(y.double().sum()) - float(z["y_sum"])) < 1e-4 * float(z["y_abs_sum"])


# ---- full-geometry configurations (BASELINE.json C4, C1/C3, C5): the oracle against the reference's own vectors ------------
@pytest.mark.parametrize("name", ["A", "B"])
def test_dense_feedback_and_partial_forcing(name):
    cfg, sd, ins, zm, ocfg = golden_model(name)
    z = load_npz("ar_" + name)
    Ts = z["soft"].shape[-1]
    c_up = torch.from_numpy(z["c_up"])[:, :, :Ts].contiguous()
    init = torch.from_numpy(z["init"])
    nf = int(z["part_forced"])
    assert rel_err(O.incremental_forward(sd, ocfg, c_up, ins["g"], Ts, initial_input=init, mode="probs"), z["soft"]) < TOL
    assert rel_err(O.incremental_forward(sd, ocfg, c_up, ins["g"], Ts, initial_input=init, mode="logits"), z["raw"]) < TOL
    assert rel_err(O.incremental_forward(sd, ocfg, c_up, ins["g"], Ts, initial_input=init, test_inputs=ins["xin"][:, :, :nf],
                                         mode="probs"), z["part"]) < TOL


def test_c4_incremental_prefix():
    """hps/vqwae.json synthesis decoder: the oracle's incremental loop over the first 160 samples of ar_c4.npz (teacher-forced
    logits at the probe steps, then the reference's greedy decisions where its margin is clear)."""
    z = load_npz("ar_c4")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    Tq = 160
    c_up = O.upsample_forward(sd, torch.from_numpy(z["lat"]), cfg["upsample_scales"])[:, :, :Tq].contiguous()
    x = torch.from_numpy(z["x"].astype(np.int64))[:, :Tq]
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    g = torch.from_numpy(z["g"])
    with torch.no_grad():
        y = O.incremental_forward(sd, ocfg, c_up, g, Tq, test_inputs=xin, mode="logits")
        pt = z["probe_t"][z["probe_t"] < Tq]
        assert rel_err(y[0][:, pt], z["tf_probe"][:, :len(pt)]) < TOL
        assert rel_err(torch.logsumexp(y, 1), z["tf_lse"][:, :Tq]) < TOL
        gr = O.incremental_forward(sd, ocfg, c_up[:, :, :64].contiguous(), g, 64, mode="argmax").argmax(1).numpy()[0]
    clear = z["greedy_margin"][0][:64] > 1e-3
    first_unclear = int(np.argmin(clear)) if not clear.all() else 64
    assert np.array_equal(gr[:first_unclear], z["greedy"][0][:first_unclear])


def test_c5_full_depth_probe():
    z = load_npz("model_c5_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    T = 1920                                   # a prefix: causal, so the first 1920 steps do not depend on the rest
    lat = O.hash_fill((1, cfg["Cc"], int(z["T"]) // 640), int(z["lat_salt"]), 1.2)[:, :, :T // 640].contiguous()
    x = ((O.hash_fill((1, int(z["T"])), int(z["x_salt"])) * 0.5 + 0.5) * 256).long().clamp(0, 255)[:, :T]
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    with torch.no_grad():
        y = O.wavenet_forward(sd, dict(layers=48, stacks=4, upsample_scales=cfg["upsample_scales"], cin_pad=0), xin, lat,
                              torch.from_numpy(z["g"]))
    # the four smoothing FIRs together look 640 + 160 + 40 + 5 samples ahead: compare away from the cut
    pt = z["probe_t"][z["probe_t"] < T - 900]
    assert rel_err(y[0][:, pt], z["y_probe"][:, :len(pt)]) < TOL


def test_vqwae_full_geometry_train_step():
    """hps/vqwae.json in full, 2 x 5120 samples: loss terms and every parameter's gradient (probes + squared norm) of the
    oracle's autograd against the reference's own train step."""
    z = load_npz("train_vqwae")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]))
    s, B, T = int(z["in_salt"]), 2, 5120
    c = O.hash_fill((B, cfg["c_in"], 32), s + 1, 1.7)
    x = ((O.hash_fill((B, T), s + 2) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    g = ((O.hash_fill((B,), s + 3) * 0.5 + 0.5) * cfg["n_speakers"]).long().clamp(0, cfg["n_speakers"] - 1)
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    y, vq, perp, aux = O.vqvae_forward(psd, ocfg, xin, c, g)
    ce = O.masked_ce_loss(y, x.unsqueeze(-1), torch.from_numpy(z["lengths"]))
    (ce + vq).backward()
    assert np.array_equal(aux["idx"].numpy(), z["vq_idx"])
    assert abs(float(ce) - float(z["ce"])) < 1e-5 * float(z["ce"])
    assert abs(float(vq) - float(z["vq_loss"])) < 1e-6 * max(1.0, float(z["vq_loss"]))
    names = json.loads(str(z["names"]))
    off = 0
    for i, k in enumerate(names):
        gk = psd[k].grad.reshape(-1) if psd[k].grad is not None else torch.zeros(psd[k].numel())
        n = gk.numel()
        idx = np.unique(np.concatenate([np.arange(min(4, n)), (np.arange(24, dtype=np.int64) * 2654435761 + 12345) % n]))
        sl = slice(off, off + len(idx))
        off += len(idx)
        assert float((gk[idx] - torch.from_numpy(z["grad_probe"][sl])).abs().max()) < 1e-3 * float(z["grad_max"][i]) + 1e-8, k   # weight_g gradients are cancelling sums (largest ~1e-5 where weight_v's are ~1e-3): absolute floor
        assert abs(float((gk.double() ** 2).sum()) - z["grad_sq"][i]) < 1e-3 * z["grad_sq"][i] + 1e-14, k


def test_c2_probe_and_train_step():
    """BASELINE config C2 (the geometry bench.py times): the oracle's logits on the 8000-sample clip against the reference's probe,
    and its autograd of the one-clip train step against the reference's gradients."""
    z = load_npz("model_c2_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    T = int(z["T"])
    lat = O.hash_fill((1, cfg["Cc"], T // 320), int(z["lat_salt"]), 1.2)
    x = ((O.hash_fill((1, T), int(z["x_salt"])) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0)
    y = O.wavenet_forward(psd, ocfg, xin, lat, torch.from_numpy(z["g"]))
    assert rel_err(y[0][:, z["probe_t"]].detach(), z["y_probe"]) < TOL
    assert rel_err(torch.logsumexp(y.detach(), 1), z["y_lse"]) < TOL
    ce = O.masked_ce_loss(y, x.unsqueeze(-1), torch.tensor([T]))
    assert abs(float(ce) - float(z["loss"])) < 1e-5 * float(z["loss"])
    ce.backward()
    names = json.loads(str(z["names"]))
    off = 0
    for i, k in enumerate(names):
        gk = psd[k].grad.reshape(-1) if psd[k].grad is not None else torch.zeros(psd[k].numel())
        n = gk.numel()
        idx = np.unique(np.concatenate([np.arange(min(4, n)), (np.arange(24, dtype=np.int64) * 2654435761 + 12345) % n]))
        sl = slice(off, off + len(idx))
        off += len(idx)
        # fp32 autograd twice (reference and oracle sum in different orders): 2e-3 of the tensor's largest gradient
        assert float((gk[idx] - torch.from_numpy(z["grad_probe"][sl])).abs().max()) < 2e-3 * float(z["grad_max"][i]) + 1e-8, k


def test_cin_pad_model():
    """cin_pad = 1 (upsample.py:69-85): conv_in with three taps and no padding; features cin_pad frames wider than the audio."""
    z = load_npz("model_P")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(cfg, int(z["salt"]))
    c, g = torch.from_numpy(z["c"]), torch.from_numpy(z["g"])
    x = torch.from_numpy(z["x"]).long()
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=1)
    assert rel_err(O.upsample_forward(sd, torch.from_numpy(z["quant"]), cfg["upsample_scales"], cin_pad=1), z["c_up"]) < TOL
    y, vq, perp, aux = O.vqvae_forward(sd, ocfg, xin, c, g)
    assert np.array_equal(aux["idx"].numpy(), z["vq_idx"])
    assert rel_err(y, z["y_hat"]) < TOL
    feats = torch.from_numpy(z["feats"]).clone().requires_grad_(True)
    yd = O.wavenet_forward(sd, ocfg, xin, feats, g)
    assert rel_err(yd[:, :, ::7].detach(), z["y_dec_probe"]) < TOL
    (yd * O.hash_fill(tuple(yd.shape), int(z["w_salt"]), 1.0)).sum().backward()
    assert rel_err(feats.grad, z["dfeats"]) < 1e-4


def test_plain_upsample_network_model():
    """upsample_net = "UpsampleNetwork" (upsample.py:29-66): no conv_in, keys `upsample_net.up_layers.N`, the output trimmed by
    cin_pad * prod(scales) samples at either end -- c_up, logits and the feature gradient of the REFERENCE's WaveNet (model_U.npz) for
    cin_pad = 1 and 0."""
    z = load_npz("model_U")
    cfg0 = json.loads(str(z["cfg"]))
    g = torch.from_numpy(z["g"])
    for pad in (1, 0):
        cfg = dict(cfg0, cin_pad=pad)
        sd = O.make_state_dict(cfg, int(z["salt"]), with_encoder=False)
        assert "wavenet.upsample_net.conv_in.weight" not in sd
        x = torch.from_numpy(z[f"x{pad}"]).long()
        xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
        feats = torch.from_numpy(z[f"feats{pad}"]).clone().requires_grad_(True)
        ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=pad, conv_in=False)
        c_up = O.upsample_forward(sd, feats.detach(), cfg["upsample_scales"], cin_pad=pad, conv_in=False)
        assert c_up.shape[-1] == x.shape[1]
        assert rel_err(c_up[:, :, ::3], z[f"c_up_probe{pad}"]) < TOL
        y = O.wavenet_forward(sd, ocfg, xin, feats, g)
        assert rel_err(y[:, :, ::5].detach(), z[f"y_probe{pad}"]) < TOL
        (y * O.hash_fill(tuple(y.shape), int(z[f"w_salt{pad}"]), 1.0)).sum().backward()
        assert rel_err(feats.grad, z[f"dfeats{pad}"]) < 1e-4


def test_upsample_activation_models():
    """upsample_activation (upsample.py:44-46): LeakyReLU(0.2) and ReLU / Sigmoid behind the stages of ConvInUpsampleNetwork, Tanh behind
    those of the plain UpsampleNetwork (FIR keys at up_layers.{3 i + 1}) -- c_up, logits and the feature gradient of the REFERENCE's
    WaveNet (model_V.npz)."""
    z = load_npz("model_V")
    g = torch.from_numpy(z["g"])
    x = torch.from_numpy(z["x"]).long()
    for tag in ("leaky", "tanh", "relu", "sigm"):
        cfg = json.loads(str(z[f"cfg_{tag}"]))
        sd = O.make_state_dict(cfg, int(z["salt"]), with_encoder=False)
        key = "wavenet.upsample_net." + ("upsample." if cfg["conv_in"] else "") + "up_layers.4.weight_v"
        assert key in sd                                  # three modules per stage: the second stage's FIR is module 4
        xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
        feats = torch.from_numpy(z["feats"]).clone().requires_grad_(True)
        ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0, conv_in=cfg["conv_in"],
                    up_act=cfg["up_act"], up_act_slope=cfg["up_act_slope"])
        c_up = O.upsample_forward(sd, feats.detach(), cfg["upsample_scales"], conv_in=cfg["conv_in"], act=cfg["up_act"],
                                  act_slope=cfg["up_act_slope"])
        assert rel_err(c_up[:, :, ::3], z[f"c_up_probe_{tag}"]) < TOL
        y = O.wavenet_forward(sd, ocfg, xin, feats, g)
        assert rel_err(y[:, :, ::5].detach(), z[f"y_probe_{tag}"]) < TOL
        (y * O.hash_fill(tuple(y.shape), int(z["w_salt"]), 1.0)).sum().backward()
        assert rel_err(feats.grad, z[f"dfeats_{tag}"]) < 1e-4
