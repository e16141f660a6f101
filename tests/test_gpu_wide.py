"""GPU parity of the wide configuration (BASELINE config C5: residual / gate / skip widths of 512) against the CPU oracle.

Widths above 256 take different launches than the C1-C4 presets: wae_gemm_tm cuts outputs wider than 256 rows into
slices, and the head runs as separate GEMM launches with epilogues (wae_gemm_tm modes 3-6) instead of the register-chained
head kernels.  The same head path is also forced onto the narrow golden models (WAE_HEAD_WIDE=1), where the fused kernels
provide a second reference."""
import pytest
import torch

from helpers import golden_model, rel_err
from oracle import wae_oracle as O

pytestmark = pytest.mark.gpu

# C5 dims (R = G = S = 512, Cc = 64, Cg = 32) on a short stack / short clips so that the oracle finishes in seconds
WIDE = dict(layers=4, stacks=2, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=8, upsample_scales=None, cin_pad=0)


def _wide_inputs(B=2, T=777):
    sd = O.make_state_dict(dict(WIDE), salt=5, with_encoder=False)
    x = ((O.hash_fill((B, T), 21) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    c = O.hash_fill((B, WIDE["Cc"], T), 22, 1.3)
    g = torch.tensor([1, 6])[:B]
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    return sd, x, xin, c, g


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_wide_forward_logits_and_loss(dtype, tol):
    """decoder logits within 1e-3 relative (fp32; bf16: 5e-2 of the logit range) and the fused shifted CE of the wide head"""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    sd, x, xin, c, g = _wide_inputs()
    B, T = x.shape
    lengths = torch.tensor([T, T - 201])
    with torch.no_grad():
        y_ref = O.wavenet_forward(sd, dict(layers=4, stacks=2, upsample_scales=None, cin_pad=0), xin, c, g)
        loss_ref = O.masked_ce_loss(y_ref, x.unsqueeze(-1), lengths)
    eng = WaeEngine(Geometry.from_cfg(WIDE), dtype=dtype)
    assert eng.wide_head
    eng.load_state_dict(sd)
    out = eng.decoder_forward(x.cuda(), c.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), y_ref) < tol
    # and against the reference's own WaveNet on the same inputs (tests/golden/model_wide_probe.npz)
    from helpers import load_npz
    z = load_npz("model_wide_probe")
    assert int(z["T"]) == T and int(z["x_salt"]) == 21 and int(z["c_salt"]) == 22 and z["g"].tolist() == g.tolist()
    assert rel_err(out["logits"].cpu()[:, :, torch.from_numpy(z["probe_t"])], z["y_probe"]) < tol
    assert abs(float(out["loss"]) - float(loss_ref)) < (1e-4 if dtype == "fp32" else 3e-2)


def _grad_check(eng, cfg, sd, x, xin, c_up, g, lengths, ocfg, ref_dtype=torch.float32, engine_relu_masks=False):
    """Gradients of the masked CE: engine vs autograd through the oracle -> name -> (max abs error, max abs reference).

    engine_relu_masks (with ref_dtype float64): at 512 channels a head ReLU sees 655 k pre-activations per clip pair and about
    one of them lies within fp32 rounding of zero; on which side an fp32 evaluation lands (the oracle's own fp32 run included:
    it differs from its fp64 run by 1 % in db1) decides a whole gradient column.  The reference is therefore the fp64 oracle with
    the head's two ReLUs (wavenet.py:209,211) applied as the masks the engine saved (h0 > 0, h1 > 0) -- identical wherever the
    pre-activation is not a rounding error away from zero, and the same function of the weights on both sides."""
    import math
    import torch.nn.functional as F
    from wavenet_autoencoders_amd import backward as BW
    out = eng.decoder_forward(x.cuda(), c_up.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), train=True,
                              c_is_upsampled=True, want_logits=False)
    psd = {k: v.clone().to(ref_dtype).requires_grad_(True) for k, v in sd.items()
           if k.startswith("wavenet.") and "upsample_net" not in k}
    cl = c_up.clone().to(ref_dtype).requires_grad_(True)
    if engine_relu_masks:
        B, T = x.shape
        fw = eng._ws[(B, T, True)]
        m0 = (fw["h0"][:, :, :cfg["S"]] > 0).transpose(1, 2).cpu().to(ref_dtype)
        m1 = (fw["h1"][:, :, :cfg["S"]] > 0).transpose(1, 2).cpu().to(ref_dtype)
        _, _, inter = O.wavenet_forward(psd, dict(ocfg, upsample_scales=None), xin.to(ref_dtype), cl, g, return_intermediates=True)
        skips = sum(s_ for _, s_ in inter) * math.sqrt(1.0 / cfg["layers"])                                  # wavenet.py:208
        h1 = F.conv1d(skips * m0, O.eff_weight(psd, "wavenet.last_conv_layers.1"), psd["wavenet.last_conv_layers.1.bias"]) * m1
        y = F.conv1d(h1, O.eff_weight(psd, "wavenet.last_conv_layers.3"), psd["wavenet.last_conv_layers.3.bias"])
    else:
        y = O.wavenet_forward(psd, dict(ocfg, upsample_scales=None), xin.to(ref_dtype), cl, g)
    loss = O.masked_ce_loss(y, x.unsqueeze(-1), lengths)
    loss.backward()
    dc = BW.decoder_backward(eng, x.cuda(), x.cuda(), lengths, g.cuda())
    grads = BW.finish_grads(eng)
    torch.cuda.synchronize()
    res = {"loss": (abs(float(out["loss"].detach()) - float(loss.detach())), abs(float(loss.detach())))}
    for k, v in psd.items():
        gref = (v.grad if v.grad is not None else torch.zeros_like(v)).float()
        got = grads[eng.lay.off(k):eng.lay.off(k) + eng.lay.numel(k)].view(eng.lay.shapes[k]).cpu()
        res[k] = (float((got - gref).abs().max()), float(gref.abs().max()))
    dc_ref = cl.grad.transpose(1, 2).float()
    res["dc"] = (float((dc[:, :, :cfg["Cc"]].float().cpu() - dc_ref).abs().max()), float(dc_ref.abs().max()))
    return res


def test_wide_backward_fp32():
    """parameter gradients of the masked CE at R = G = S = 512 against autograd through the fp64 oracle (sliced wae_gemm_tm
    outputs, wide-head backward launches, 512-wide weight-gradient tiles): 1e-3 of each tensor's range (measured 2e-6)"""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    sd, x, xin, c, g = _wide_inputs(T=640)
    T = x.shape[1]
    eng = WaeEngine(Geometry.from_cfg(WIDE), dtype="fp32")
    eng.load_state_dict(sd)
    res = _grad_check(eng, WIDE, sd, x, xin, c, g, torch.tensor([T, T - 137]), dict(layers=4, stacks=2, cin_pad=0),
                      ref_dtype=torch.float64, engine_relu_masks=True)
    bad = {k: v for k, v in res.items() if v[0] > 1e-3 * max(v[1], 1e-6) + 1e-7}
    assert not bad, bad


def test_wide_train_step_bf16_matches_fp32_engine():
    """bf16 (stream-K weight gradients over 512-wide regions) against the fp32 engine on the same step: loss and the
    gradient norm agree to bf16 accuracy, and a repeated batch learns"""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    sd, x, xin, c, g = _wide_inputs(T=1280)
    got = {}
    for dtype in ("fp32", "bf16"):
        eng = WaeEngine(Geometry.from_cfg(WIDE), dtype=dtype)
        eng.load_state_dict(sd)
        losses = []
        for _ in range(4):
            r = eng.train_step(x.cuda(), c.cuda(), g.cuda(), lr=1e-3)
            losses.append(float(r["loss"]))
            if len(losses) == 1:
                gn = float(r["grad_norm"])
        torch.cuda.synchronize()
        got[dtype] = (losses, gn)
        assert bool(torch.isfinite(eng.params).all())
        assert losses[-1] < losses[0], losses
    assert abs(got["bf16"][0][0] - got["fp32"][0][0]) < 3e-2
    assert abs(got["bf16"][1] - got["fp32"][1]) < 0.1 * got["fp32"][1], got


@pytest.mark.parametrize("name", ["A", "B"])
def test_wide_head_path_on_golden_models_fp32(name, monkeypatch):
    """the separate-launch head (forced by WAE_HEAD_WIDE=1) on the golden models: logits against the reference's own
    vectors, gradients against autograd through the oracle"""
    monkeypatch.setenv("WAE_HEAD_WIDE", "1")
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model(name)
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype="fp32")
    assert eng.wide_head
    eng.load_state_dict(sd)
    x, g = ins["x"], ins["g"]
    T = x.shape[1]
    c_up = torch.from_numpy(z["c_up"])
    out = eng.decoder_forward(x.cuda(), c_up.cuda(), g.cuda(), c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), torch.from_numpy(z["y_hat"])) < 1e-3
    res = _grad_check(eng, cfg, sd, x, ins["xin"], c_up, g, torch.tensor([T, T - 137]), ocfg)
    bad = {k: v for k, v in res.items() if v[0] > 1e-3 * max(v[1], 1e-6) + 1e-7}
    assert not bad, bad


def test_wide_incremental_forward_matches_teacher_forced_fp32():
    """WaveNet.incremental_forward (wavenet.py:218-346) at R = S = 512 (1024 rows in the second matrix-vector product, more
    rows than threads): teacher-forced incremental logits against the oracle's parallel forward on the same inputs"""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    sd, x, xin, c, g = _wide_inputs(B=2, T=96)
    with torch.no_grad():
        y_ref = O.wavenet_forward(sd, dict(layers=4, stacks=2, upsample_scales=None, cin_pad=0), xin, c, g)
    eng = WaeEngine(Geometry.from_cfg(WIDE), dtype="fp32")
    eng.load_state_dict(sd)
    out = eng.incremental_forward(c.cuda(), g.cuda(), T=x.shape[1], mode="logits", test_inputs=x.cuda(), c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), y_ref) < 1e-3


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("geom", ["c5", "hp256", "r512s256"])
def test_static_weight_gradient_launch_on_cut_layers(geom, dtype, monkeypatch):
    """Round 5: layers wider than one region of wae_gemm_tn_static (dz > 384 columns, x / Ghat / dS > 256, u > 192) are cut into several
    jobs per kind, dealt into groups of at most six (backward._build_stream_table: C5's 512-wide layer = 12 tap + 2 conditioning + 4
    out/skip jobs in three groups).  Same 16-bit operands, three kinds of cut: against the per-layer 128 x 128 tile launches
    (WAE_TN_STREAM=0) to 2e-4 of each tensor's range; the any-shape stream-K launch (WAE_TN_STATIC=0) agrees to the same bound."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(WIDE)
    if geom == "hp256":       # only u (Hp = 256) and dz (512 columns) are cut; x, Ghat, dS are one region
        cfg.update(R=256, S=256, G=512)
    elif geom == "r512s256":      # x / Ghat cut in two, dS (256), dz (384 columns) and u (192) one region: out/skip jobs with and without a dS half
        cfg.update(R=512, S=256, G=384)
    sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
    B, T = 3, 777
    x = ((O.hash_fill((B, T), 21) * 0.5 + 0.5) * 256).long().clamp(0, 255).cuda()
    c = O.hash_fill((B, cfg["Cc"], T), 22, 1.3).cuda()
    g = (torch.arange(B) % cfg["n_speakers"]).cuda()
    lengths = torch.tensor([T, T - 137, T - 400])
    got = {}
    for tag, env in (("tiles", {"WAE_TN_STREAM": "0"}), ("static", {}), ("streamk", {"WAE_TN_STATIC": "0"})):
        for k in ("WAE_TN_STREAM", "WAE_TN_STATIC"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        eng.decoder_forward(x, c, g, targets=x, lengths=lengths.cuda(), train=True, c_is_upsampled=True, want_logits=False)
        BW.decoder_backward(eng, x, x, lengths, g)
        st = BW.bwd_workspace(eng, B, T)["stream"]
        if tag == "tiles":
            assert st is None
        elif tag == "static":
            assert isinstance(st, BW.StaticStreamTable) and BW.static_tn_split(eng) and not BW.static_head(eng, B, T)
            assert st.team_size <= 6 and len(st.groups) % cfg["layers"] == 0
        else:
            assert isinstance(st, BW.StreamTable)
        got[tag] = BW.finish_grads(eng).clone()
        torch.cuda.synchronize()
    lay = eng.lay
    for tag in ("static", "streamk"):
        bad = {}
        for kk in lay.offsets:
            a = got["tiles"][lay.off(kk):lay.off(kk) + lay.numel(kk)]
            b = got[tag][lay.off(kk):lay.off(kk) + lay.numel(kk)]
            err, ref = float((a - b).abs().max()), float(a.abs().max())
            if err > 2e-4 * max(ref, 1e-6) + 1e-7:
                bad[kk] = (err, ref)
        assert not bad, (tag, bad)
