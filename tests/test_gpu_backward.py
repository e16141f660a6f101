"""GPU parity of the decoder backward pass against torch autograd through the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import golden_model, rel_err
from oracle import wae_oracle as O

pytestmark = pytest.mark.gpu


def _run(name, dtype, lengths_list):
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model(name)
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    eng.load_state_dict(sd)
    x, g = ins["x"], ins["g"]
    B, T = x.shape
    c_up = torch.from_numpy(z["c_up"])
    lengths = torch.tensor(lengths_list)
    # oracle: autograd wrt decoder parameters and the upsampled conditioning
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("wavenet.") and "upsample_net" not in k}
    cl = c_up.clone().requires_grad_(True)
    y = O.wavenet_forward(psd, dict(ocfg, upsample_scales=None), ins["xin"], cl, g)
    loss = O.masked_ce_loss(y, x.unsqueeze(-1), lengths)
    loss.backward()
    out = eng.decoder_forward(x.cuda(), c_up.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), train=True,
                              c_is_upsampled=True, want_logits=False)
    dc = BW.decoder_backward(eng, x.cuda(), x.cuda(), lengths, g.cuda())
    grads = BW.finish_grads(eng)
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(loss)) < (1e-4 if dtype == "fp32" else 3e-2)
    res = {}
    for k, v in psd.items():
        gref = v.grad if v.grad is not None else torch.zeros_like(v)
        got = grads[eng.lay.off(k):eng.lay.off(k) + eng.lay.numel(k)].view(eng.lay.shapes[k]).cpu()
        res[k] = (float((got - gref).abs().max()), float(gref.abs().max()))
    dc_ref = cl.grad.transpose(1, 2)
    res["dc"] = (float((dc[:, :, :cfg["Cc"]].float().cpu() - dc_ref).abs().max()), float(dc_ref.abs().max()))
    return res


@pytest.mark.parametrize("name", ["A", "B"])
def test_decoder_backward_fp32(name):
    res = _run(name, "fp32", [1280 if name == "A" else 640, (1280 if name == "A" else 640) - 137])
    bad = {k: v for k, v in res.items() if v[0] > 1e-3 * max(v[1], 1e-6) + 1e-7}
    assert not bad, bad


@pytest.mark.parametrize("name", ["A", "B"])
def test_decoder_backward_fused_boundary_kernel_fp32(name, monkeypatch):
    """wae_glu_bwd_fused (K_X of layer l + K_U of layer l-1 in one launch; opt-in) against autograd through the oracle."""
    monkeypatch.setenv("WAE_BWD_FUSED", "1")
    res = _run(name, "fp32", [1280 if name == "A" else 640, (1280 if name == "A" else 640) - 137])
    bad = {k: v for k, v in res.items() if v[0] > 1e-3 * max(v[1], 1e-6) + 1e-7}
    assert not bad, bad


def test_scalar_input_dmol_backward_fp32():
    """input_type "raw" (hparams.py default): scalar input, discretized-mixture-of-logistics loss (mixture.py:26-106 through
    vqwae_train.py:382-401, shifted by one) -- decoder parameter gradients against autograd through the oracle."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model("S")
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype="fp32")
    eng.load_state_dict(sd)
    x, g = ins["x"][:, 0, :].contiguous(), ins["g"]   # x (B,T) fp32 in [-1,1]
    B, T = x.shape
    c_up = torch.from_numpy(z["c_up"])
    lengths = torch.tensor([T, T - 97])
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("wavenet.") and "upsample_net" not in k}
    y = O.wavenet_forward(psd, dict(ocfg, upsample_scales=None), ins["xin"], c_up, g)
    loss = O.masked_dmol_loss(y, x.unsqueeze(-1), lengths, 65536, -7.0)
    loss.backward()
    out = eng.decoder_forward(x.cuda(), c_up.cuda(), g.cuda(), train=True, c_is_upsampled=True, want_logits=True)
    got_loss, dyt = eng.dmol_loss_and_grad(out["logits"], x.cuda(), lengths.cuda(), 65536, -7.0)
    BW.decoder_backward(eng, x.cuda(), None, lengths, g.cuda(), ext_dy=dyt)
    grads = BW.finish_grads(eng)
    torch.cuda.synchronize()
    assert abs(float(got_loss) - float(loss)) < 1e-4 * max(1.0, abs(float(loss)))
    bad = {}
    for k, v in psd.items():
        gref = v.grad if v.grad is not None else torch.zeros_like(v)
        got = grads[eng.lay.off(k):eng.lay.off(k) + eng.lay.numel(k)].view(eng.lay.shapes[k]).cpu()
        err, ref = float((got - gref).abs().max()), float(gref.abs().max())
        if err > 1e-3 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_scalar_input_train_step_runs_and_learns(dtype):
    """VQVAE with a scalar-input decoder: full train step (encoder, VQ, DMoL loss, backward, Adam); the loss of a repeated
    batch goes down and every tensor stays finite."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model("S")
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    eng.load_state_dict(sd)
    x, c, g = ins["x"][:, 0, :].contiguous().cuda(), ins["c"].cuda(), ins["g"].cuda()
    losses = []
    for _ in range(6):
        r = eng.train_step(x, c, g, lengths=None, lr=2e-3, quantize_channels=65536, log_scale_min=-7.0)
        losses.append(float(r["ce"]))
    torch.cuda.synchronize()
    assert all(np.isfinite(l) for l in losses), losses
    assert bool(torch.isfinite(eng.params).all())
    assert losses[-1] < losses[0], losses


def test_decoder_backward_bf16_is_close():
    res = _run("A", "bf16", [1280, 1280])
    # bf16 storage of activations/gradients: compare at 8 % of each tensor's gradient range (dc is a heavily
    # cancelling sum over layers and gate channels of bf16-rounded dz: 30 %)
    bad = {k: v for k, v in res.items() if v[0] > (0.3 if k == "dc" else 8e-2) * max(v[1], 1e-6) + 1e-6}
    assert not bad, bad


@pytest.mark.parametrize("name,lengths", [("A", [1280, 1280 - 137]), ("B", [640, 500])])
def test_stream_weight_gradients_match_per_layer_tiles(name, lengths, monkeypatch):
    """bf16: wae_gemm_tn_stream (all layers, one launch, 384x256 regions) against wae_gemm_tn_tiles (per layer, 128x128 tiles):
    same bf16 operands, fp32 accumulation in a different order -> 2e-4 of each gradient tensor's range."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model(name)
    x, g = ins["x"].cuda(), ins["g"].cuda()
    c_up = torch.from_numpy(z["c_up"]).cuda()
    ln = torch.tensor(lengths)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("WAE_TN_STREAM", mode)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype="bf16")
        eng.load_state_dict(sd)
        eng.decoder_forward(x, c_up, g, targets=x, lengths=ln.cuda(), train=True, c_is_upsampled=True, want_logits=False)
        BW.decoder_backward(eng, x, x, ln, g)
        assert (BW.bwd_workspace(eng, *x.shape)["stream"] is not None) == (mode == "1")
        got[mode] = BW.finish_grads(eng).clone()
        torch.cuda.synchronize()
    lay = eng.lay
    bad = {}
    for k in lay.offsets:
        if not k.startswith("wavenet.") or "upsample_net" in k:     # (round 4: the head's and the first conv's gradients ride in the launch too)
            continue
        a = got["0"][lay.off(k):lay.off(k) + lay.numel(k)]
        b = got["1"][lay.off(k):lay.off(k) + lay.numel(k)]
        err, ref = float((a - b).abs().max()), float(a.abs().max())
        if err > 2e-4 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T", [(1, 2), (1, 15), (2, 16), (32, 17), (1, 33), (3, 130), (2, 257), (5, 1000), (32, 40), (7, 4099)])
def test_static_weight_gradient_launch_at_odd_shapes(B, T, dtype, monkeypatch):
    """wae_gemm_tn_static (hardware zero fill through per-clip buffer descriptors, teams walking segments that start and end anywhere)
    against the per-layer 128 x 128 tile launches on the same 16-bit operands: clips shorter than a half-slab, T % 16 != 0, the
    maximum batch of the static launch, clips longer than a team's share; ragged lengths.  2e-4 of each tensor's range (fp32 sums in
    a different order)."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model("B")
    x = ((O.hash_fill((B, T), 191) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1).cuda()
    c_up = O.hash_fill((B, cfg["Cc"], T), 192, 1.1).cuda()
    g = (torch.arange(B) % cfg["n_speakers"]).cuda()
    ln = torch.tensor([T] + [max(2, T - 1 - 7 * i) for i in range(1, B)])
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("WAE_TN_STREAM", mode)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        eng.decoder_forward(x, c_up, g, targets=x, lengths=ln.cuda(), train=True, c_is_upsampled=True, want_logits=False)
        BW.decoder_backward(eng, x, x, ln, g)
        st = BW.bwd_workspace(eng, B, T)["stream"]
        assert (st is None) if mode == "0" else isinstance(st, BW.StaticStreamTable)
        assert mode == "0" or BW.static_head(eng, B, T)
        got[mode] = BW.finish_grads(eng).clone()
        torch.cuda.synchronize()
    lay = eng.lay
    bad = {}
    for k in lay.offsets:
        if not k.startswith("wavenet.") or "upsample_net" in k:
            continue
        a = got["0"][lay.off(k):lay.off(k) + lay.numel(k)]
        b = got["1"][lay.off(k):lay.off(k) + lay.numel(k)]
        err, ref = float((a - b).abs().max()), float(a.abs().max())
        if err > 2e-4 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad


def test_full_train_step_against_golden():
    """(7) of SURVEY 8c: parameter gradients of one step, post-Adam weights and EMA shadow of the reference."""
    from helpers import load_npz
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, zm, ocfg = golden_model("A")
    z = load_npz("train_A")
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype="fp32")
    eng.load_state_dict(sd)
    x, c, g = ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda()
    T = x.shape[1]
    res = eng.train_step(x, c, g, lengths=torch.tensor([T, T]), lr=4e-4, clip_thresh=100.0, ema_decay=0.9999)
    torch.cuda.synchronize()
    assert abs(float(res["loss"]) - float(z["loss"])) < 1e-4
    assert abs(float(res["vq_loss"]) - float(z["vq_loss"])) < 1e-5
    assert abs(float(res["grad_norm"]) - float(z["grad_norm"])) < 1e-3 * float(z["grad_norm"])
    import json
    gsq = json.loads(str(z["grad_sq_by_key"]))
    lay = eng.lay
    bad = {}
    for k, want in gsq.items():
        got = float((eng.grads[lay.off(k):lay.off(k) + lay.numel(k)].double() ** 2).sum())
        if abs(got - want) > 5e-3 * max(want, 1e-12) + 1e-14:
            bad[k] = (got, want)
    assert not bad, bad
    for key in [k[5:] for k in z if k.startswith("grad:")]:
        view = lambda a: a[lay.off(key):lay.off(key) + lay.numel(key)].view(lay.shapes[key]).cpu()  # noqa: E731
        assert rel_err(view(eng.grads), z["grad:" + key]) < 2e-3, key
        assert rel_err(view(eng.params), z["new:" + key]) < 5e-5, key   # first Adam step ~ lr*sign(g): tiny-|g| elements are eps-sensitive
        assert rel_err(view(eng.shadow), z["ema:" + key]) < 1e-6, key


def test_c2_full_size_backward_properties(monkeypatch):
    """BASELINE config C2 at full size (24 layers, 8 x 8000 samples; too large for the oracle), backward: (1) the bf16 stream-K weight
    gradients (one launch, every layer, dilations up to 2048) agree with the per-layer tile kernel on the same operands to 2e-4 of
    each tensor's range; (2) the bf16 gradient arena agrees with the fp32 engine's to 5e-2 of its norm; (3) clip 0's contribution is
    independent of its batch neighbours: gradients of a batch whose other clips are masked out (lengths 0) equal those of clip 0
    alone in fp32 (atomics: to rounding)."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5],
               cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
    B, T = 8, 8000
    gen = torch.Generator().manual_seed(4321)
    x = torch.randint(0, 256, (B, T), generator=gen).cuda()
    lat = torch.randn(B, 64, T // 320, generator=gen).cuda()
    g = torch.randint(0, 153, (B,), generator=gen).cuda()

    def grads_of(dtype, stream, xs, lats, gs, lengths):
        monkeypatch.setenv("WAE_TN_STREAM", stream)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        eng.decoder_forward(xs, lats, gs, targets=xs, lengths=lengths.cuda() if lengths is not None else None, train=True,
                            want_logits=False)
        dc = BW.decoder_backward(eng, xs, xs, lengths, gs)
        BW.frontend_backward(eng, dc)
        out = BW.finish_grads(eng).clone()
        torch.cuda.synchronize()
        lay = eng.lay
        del eng
        torch.cuda.empty_cache()
        return out, lay

    g_stream, lay = grads_of("bf16", "1", x, lat, g, None)
    g_tiles, _ = grads_of("bf16", "0", x, lat, g, None)
    bad = {}
    for k in lay.offsets:
        if k.startswith("wavenet.conv_layers."):
            a, b_ = g_tiles[lay.off(k):lay.off(k) + lay.numel(k)], g_stream[lay.off(k):lay.off(k) + lay.numel(k)]
            err, ref = float((a - b_).abs().max()), float(a.abs().max())
            if err > 2e-4 * max(ref, 1e-6) + 1e-7:
                bad[k] = (err, ref)
    assert not bad, bad
    g32, _ = grads_of("fp32", "0", x, lat, g, None)
    rel = float((g_stream - g32).double().norm() / g32.double().norm())
    assert rel < 5e-2, rel
    # (3) masked-out neighbours contribute nothing
    lens = torch.tensor([T] + [0] * (B - 1))
    g_masked, _ = grads_of("fp32", "0", x, lat, g, lens)
    g_alone, _ = grads_of("fp32", "0", x[:1], lat[:1], g[:1], torch.tensor([T]))
    rel = float((g_masked - g_alone).double().norm() / g_alone.double().norm())
    assert rel < 1e-5, rel


@pytest.mark.parametrize("dtype,name,chains", [("fp32", "A", "auto"), ("bf16", "A", "auto"), ("bf16", "B", "auto"), ("bf16", "B", "2")])
def test_grad_sync_path_equals_plain_backward(dtype, name, chains):
    """Data-parallel step order: with a GradSync the arena is handed to the all-reduce in THREE pieces -- the upper half
    of the gated layers + the head in the middle of the backward sweep (the stream-K weight-gradient launch is cut into two layer
    halves for that), the lower half at its end, the rest in finish_grads; fp32 (per-layer tile launches) hands the layers + head
    over in one piece.  Single process (no collective runs): the hand-over order is checked, and the gradients and the updated
    weights must equal those of the plain order.  chains = "2": the sweep as two half-batch chains of launches (engine.chain_plan),
    joined for the hand-over in its middle and forked again behind it."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.distributed import GradSync
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model(name)
    x, c, g = ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda()
    lengths = torch.tensor([x.shape[1], x.shape[1] - 333])
    got = {}
    handed = []
    for tag in ("plain", "sync"):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.opt.chains = chains
        eng.load_state_dict(sd)
        eng.init_optimizer()
        assert (eng.chain_plan(*x.shape) is not None) == (chains == "2")
        seen = {}
        sync = GradSync(eng) if tag == "sync" else None
        if sync is not None:
            lo, hi = BW.layer_segment(eng)
            assert 0 < lo < hi < eng.lay.total and (hi - lo) > 0.5 * eng.lay.total
            assert any(b[0] == lo for b in sync.bounds) and any(b[1] == hi for b in sync.bounds)
            mid = BW.layer_segment_mid(eng)
            assert lo < mid < hi and any(b[0] == mid for b in sync.bounds)
            inner = sync.ready_range
            sync.ready_range = lambda a, b_, inner=inner: (handed.append((a, b_)), inner(a, b_))[1]
        r = eng.train_step(x, c, g, lengths=lengths, grad_sync=sync, grad_hook=lambda gr: seen.setdefault("g", gr.clone()))
        torch.cuda.synchronize()
        if sync is not None:
            assert sync.launched == [False] * len(sync.bounds)          # finish() re-armed the buckets
            assert handed == ([(mid, hi), (lo, mid)] if dtype == "bf16" else [(lo, hi)]), handed
        got[tag] = (seen["g"].cpu(), eng.params.cpu().clone(), float(r["loss"]), float(r["grad_norm"]))
    ga, gb = got["plain"][0], got["sync"][0]
    tol = 1e-6 if dtype == "fp32" else 1e-5            # fp32 atomics arrive in another order
    assert float((ga - gb).abs().max()) <= tol * float(ga.abs().max())
    assert float((got["plain"][1] - got["sync"][1]).abs().max()) < 1e-6
    assert abs(got["plain"][2] - got["sync"][2]) < 1e-6 and abs(got["plain"][3] - got["sync"][3]) < 1e-4 * got["plain"][3]


@pytest.mark.parametrize("p_drop", [0.05, 0.3])
def test_dropout_forward_and_backward_against_oracle(p_drop):
    """modules.py:127-128: F.dropout on the input of every dilated convolution in training mode (the residual path keeps the
    undropped input).  The engine's mask is a counter-based hash; the oracle applies the SAME mask (oracle.dropout_keep restates
    the hash) where the reference draws from torch's RNG: loss and every decoder parameter gradient against autograd, fp32."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model("A")
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype="fp32", dropout=p_drop, drop_seed=77)
    eng.load_state_dict(sd)
    x, g = ins["x"], ins["g"]
    B, T = x.shape
    c_up = torch.from_numpy(z["c_up"])
    lengths = torch.tensor([T, T - 137])
    out = eng.decoder_forward(x.cuda(), c_up.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), train=True,
                              c_is_upsampled=True, want_logits=True)
    seeds = list(eng._drop_seeds)
    assert seeds == [eng.layer_drop_seed(1, l) for l in range(cfg["layers"])]
    keep0 = O.dropout_keep(seeds[0], B, cfg["R"], T, p_drop)
    assert abs(float(keep0.float().mean()) - (1 - p_drop)) < 0.01                      # Bernoulli(1 - p)
    assert not torch.equal(keep0, O.dropout_keep(seeds[1], B, cfg["R"], T, p_drop))    # a fresh mask per layer
    BW.decoder_backward(eng, x.cuda(), x.cuda(), lengths, g.cuda())
    grads = BW.finish_grads(eng)
    torch.cuda.synchronize()
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("wavenet.") and "upsample_net" not in k}
    y = O.wavenet_forward(psd, dict(ocfg, upsample_scales=None), ins["xin"], c_up, g, dropout=(p_drop, seeds))
    loss = O.masked_ce_loss(y, x.unsqueeze(-1), lengths)
    loss.backward()
    assert rel_err(out["logits"].cpu(), y.detach()) < 1e-4
    assert abs(float(out["loss"]) - float(loss.detach())) < 1e-5 * float(loss.detach())
    bad = {}
    for k, v in psd.items():
        gref = v.grad if v.grad is not None else torch.zeros_like(v)
        got = grads[eng.lay.off(k):eng.lay.off(k) + eng.lay.numel(k)].view(eng.lay.shapes[k]).cpu()
        err, ref = float((got - gref).abs().max()), float(gref.abs().max())
        if err > 1e-3 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad
    # eval-mode forward of the same engine: no mask
    ev = eng.decoder_forward(x.cuda(), c_up.cuda(), g.cuda(), c_is_upsampled=True)["logits"].cpu()
    y0 = O.wavenet_forward(sd, dict(ocfg, upsample_scales=None), ins["xin"], c_up, g)
    assert rel_err(ev, y0) < 1e-4


def test_dropout_train_step_bf16_learns():
    """full VQVAE train steps with dropout 0.05 (the reference's constructor default) in bf16: finite, the loss of a repeated
    batch goes down, a second engine with the same seed reproduces the first step exactly (the mask is a pure function of seed,
    call number and layer)."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model("A")
    x, c, g = ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda()
    runs = []
    for _ in range(2):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype="bf16", dropout=0.05, drop_seed=5)
        eng.load_state_dict(sd)
        eng.init_optimizer()
        runs.append([float(eng.train_step(x, c, g, lr=2e-3)["ce"]) for _ in range(6)])
        torch.cuda.synchronize()
        assert bool(torch.isfinite(eng.params).all())
    assert runs[0][0] == runs[1][0]
    assert runs[0][-1] < runs[0][0]


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["B", "s44", "c2", "c3dec", "cc80", "nocond", "k2", "k4", "nofold"])
def test_fused_residual_gate_launch_16bit(case, dtype, monkeypatch):
    """Round 5: K_X of layer l + K_U of layer l-1 as ONE launch in 16-bit storage (csrc/glu_bwd.hip: glu_bwd_pair_kernel; two workgroups
    per CU, the residual launch's weight stream and chunk order; opt-in WAE_BWD_FUSED=1) against the two wae_gemm_tm launches on the
    same operands: the top layer's dx-hat BITWISE equal (same weights, same MFMA order), dz and the dx-hat below to 2e-2 of their range
    (the W_out^T contraction sums the same products in accumulator-row order; dz is rounded to 16 bits once in both paths, and every
    layer below inherits that rounding), dc and every parameter gradient to 2e-2 of each tensor's range.  Golden model B (Rp 128, Hp 64: instance <4, 2>), a 128 x 128 geometry (<4, 4>), C2 at full size (<8, 6>), the
    hps/vqwae.json decoder (<8, 4>); ragged lengths.
    The launch is the 16-bit DEFAULT (WAE_BWD_FUSED=auto) for every supported (Rp, Hp), also where the conditioning gradient cannot ride
    in it -- the FOLD = false instantiation (round-5 advisor finding: never exercised in 16 bits): Cc = 80 (the reference's default
    cin_channels: Ccp = 128), no local conditioning, 2 and 4 taps, and WAE_BWD_FOLD_DC=0 on a geometry that folds by default."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    if case == "nofold":
        monkeypatch.setenv("WAE_BWD_FOLD_DC", "0")
        case = "B"
    if case in ("s44", "cc80", "nocond", "k2", "k4"):
        cc = {"cc80": 80, "nocond": -1}.get(case, 64)
        cfg = dict(layers=4, stacks=2, R=96, G=256, S=96, O=64, Cc=cc, Cg=16, k={"k2": 2, "k4": 4}.get(case, 3), n_speakers=5,
                   upsample_scales=None)
        sd = O.make_state_dict(dict(cfg), salt=9, with_encoder=False)
        B, T = 3, 700
        x = ((O.hash_fill((B, T), 31) * 0.5 + 0.5) * 64).long().clamp(0, 63).cuda()
        c = O.hash_fill((B, cc, T), 32, 1.2).cuda() if cc > 0 else None
        g = (torch.arange(B) % 5).cuda()
        up = True
    elif case == "B":
        cfg, sd, ins, z, ocfg = golden_model(case)
        cfg = {k: v for k, v in cfg.items() if k not in ("encoder_hid", "c_in", "K")}
        sd = {k: v for k, v in sd.items() if k.startswith("wavenet.")}
        x, g = ins["x"].cuda(), ins["g"].cuda()
        c = torch.from_numpy(z["c_up"]).cuda()
        up = True
    else:
        if case == "c2":
            cfg = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5],
                       cin_pad=0)
            B, T, hop = 8, 8000, 320
        else:
            cfg = dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5],
                       cin_pad=0)
            B, T, hop = 3, 1920, 640
        sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
        gen = torch.Generator().manual_seed(77)
        x = torch.randint(0, 256, (B, T), generator=gen).cuda()
        c = torch.randn(B, 64, T // hop, generator=gen).cuda()
        g = torch.randint(0, cfg["n_speakers"], (B,), generator=gen).cuda()
        up = False
    B, T = x.shape
    lengths = torch.tensor([T - 53 * i for i in range(B)])
    got = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("WAE_BWD_FUSED", fused)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd, strict=False)
        eng.decoder_forward(x, c, g, targets=x, lengths=lengths.cuda(), train=True, c_is_upsampled=up, want_logits=False)
        dc = BW.decoder_backward(eng, x, x, lengths, g)
        assert eng.fused_bwd == (fused == "1")
        ws = eng._ws[("bwd", B, T)]
        if dc is None:
            dc = torch.zeros(1, device="cuda")
        got[fused] = (ws["dz"].clone(), [t_.clone() for t_ in ws["gx"]], dc.clone(), BW.finish_grads(eng).clone())
        torch.cuda.synchronize()
        lay = eng.lay
        del eng
        torch.cuda.empty_cache()
    a, b_ = got["0"][1][-1], got["1"][1][-1]          # dx-hat of the top layer: same dz, same weights, same chunk and MFMA order
    assert torch.equal(a.view(torch.int16), b_.view(torch.int16)), ("top dx-hat", float((a.float() - b_.float()).abs().max()))
    for a, b_ in zip(got["0"][1], got["1"][1]):
        err, ref = float((a.float() - b_.float()).abs().max()), float(a.float().abs().max())
        assert err < 2e-2 * ref + 1e-9, ("dx-hat", err, ref)
    for name, a, b_ in (("dz", got["0"][0], got["1"][0]), ("dc", got["0"][2], got["1"][2])):
        err, ref = float((a.float() - b_.float()).abs().max()), float(a.float().abs().max())
        assert err < 2e-2 * ref + 1e-9, (name, err, ref)
    bad = {}
    for k in lay.offsets:
        a = got["0"][3][lay.off(k):lay.off(k) + lay.numel(k)]
        b_ = got["1"][3][lay.off(k):lay.off(k) + lay.numel(k)]
        err, ref = float((a - b_).abs().max()), float(a.abs().max())
        if err > 2e-2 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("case", ["c2", "c2small", "c3small"])
def test_pair8_launch_is_bitwise_the_pair_launch(case, dtype):
    """Round 6: the backward pair launch on 8 waves x 256 columns with both operand streams through LDS (csrc/glu_bwd8.hip) against the
    round-5 kernel (4 waves x 128 columns, csrc/glu_bwd.hip; forced by dc_mode bit 2): same packed streams, same MFMA order per
    accumulator -- dz of every layer, every dx-hat and dc are compared BITWISE.  C2 at full size (Hp 192: 36 + 6 + 8 half-chunks), the
    same geometry on short ragged clips whose last tile is mostly past the end, and the hps/vqwae.json decoder (Hp 128)."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    if case == "c2":
        cfg = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5], cin_pad=0)
        B, T, hop = 8, 8000, 320
    elif case == "c2small":
        cfg = dict(layers=6, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5], cin_pad=0)
        B, T, hop = 3, 960, 320
    else:
        cfg = dict(layers=6, stacks=3, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5], cin_pad=0)
        B, T, hop = 2, 1280, 640
    sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
    gen = torch.Generator().manual_seed(177)
    x = torch.randint(0, 256, (B, T), generator=gen).cuda()
    c = torch.randn(B, 64, T // hop, generator=gen).cuda()
    g = torch.randint(0, cfg["n_speakers"], (B,), generator=gen).cuda()
    lengths = torch.tensor([T - 61 * i for i in range(B)])
    got = {}
    for four in (True, False):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.bwd_pair4 = four
        eng.load_state_dict(sd, strict=False)
        eng.decoder_forward(x, c, g, targets=x, lengths=lengths.cuda(), train=True, want_logits=False)
        dc = BW.decoder_backward(eng, x, x, lengths, g)
        assert eng.fused_bwd
        ws = eng._ws[("bwd", B, T)]
        got[four] = (ws["dz"].clone(), [t_.clone() for t_ in ws["gx"]], dc.clone(), BW.finish_grads(eng).clone())
        torch.cuda.synchronize()
        del eng
        torch.cuda.empty_cache()
    i16 = lambda t_: t_.view(torch.int16)  # noqa: E731
    assert float(got[True][0].float().abs().max()) > 0
    assert torch.equal(i16(got[True][0]), i16(got[False][0])), "dz"
    for i, (a, b_) in enumerate(zip(got[True][1], got[False][1])):
        assert torch.equal(i16(a), i16(b_)), ("dx-hat", i)
    assert torch.equal(i16(got[True][2]), i16(got[False][2])), "dc"


@pytest.mark.parametrize("case", ["c2", "c2small", "c3small", "wide"])
def test_two_chains_are_bitwise_one_chain(case):
    """Round 6: the gated stack and the backward sweep as two half-batch chains of launches on two streams, the second started half a
    launch late (engine.chain_plan, include/wae.h: wae_stream_delay) against one chain of full-batch launches.  Same kernels, same
    arithmetic per clip: every saved activation of the forward, the loss terms, dz of every layer, every dx-hat and dc are compared
    BITWISE; the weight gradients (one launch downstream of both chains, fp32 atomics) to the tolerance of two runs of that launch.
    C2 at full size (auto: 250 workgroups per launch -- the backward sweep; its forward stack stays on one chain), an odd batch of short ragged clips (forced: 2 + 1 clips), the hps/vqwae.json
    decoder, and a 512-wide fp16 model whose sweep runs the two launches per layer (forced)."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    dtype = "bf16"
    if case == "c2":
        cfg = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5], cin_pad=0)
        B, T, hop, force = 8, 8000, 320, "auto"
    elif case == "c2small":
        cfg = dict(layers=6, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5], cin_pad=0)
        B, T, hop, force = 3, 960, 320, "2"
    elif case == "c3small":
        cfg = dict(layers=6, stacks=3, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5], cin_pad=0)
        B, T, hop, force = 2, 1280, 640, "2"
    else:
        cfg = dict(layers=4, stacks=2, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=7, upsample_scales=[4, 4, 4, 5], cin_pad=0)
        B, T, hop, force, dtype = 4, 1600, 320, "2", "fp16"
    sd = O.make_state_dict(dict(cfg), salt=6, with_encoder=False)
    gen = torch.Generator().manual_seed(277)
    x = torch.randint(0, 256, (B, T), generator=gen).cuda()
    c = torch.randn(B, 64, T // hop, generator=gen).cuda()
    g = torch.randint(0, cfg["n_speakers"], (B,), generator=gen).cuda()
    lengths = torch.tensor([T - 61 * i for i in range(B)])
    got = {}
    for chains in ("1", force):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.opt.chains = chains
        eng.load_state_dict(sd, strict=False)
        assert (eng.chain_plan(B, T, backward=True) is not None) == (chains != "1")
        out = eng.decoder_forward(x, c, g, targets=x, lengths=lengths.cuda(), train=True, want_logits=False)
        fw = eng._ws[(B, T, True)]
        dc = BW.decoder_backward(eng, x, x, lengths, g)
        ws = eng._ws[("bwd", B, T)]
        torch.cuda.synchronize()
        got[chains] = dict(x=[t_.clone() for t_ in fw["x"]], z=[t_.clone() for t_ in fw["z"]], u=fw["u"].clone(), nll=fw["nll"].clone(),
                           loss=out["loss"].detach().clone(), dz=ws["dz"].clone(), gx=[t_.clone() for t_ in ws["gx"]], dc=dc.clone(),
                           grads=BW.finish_grads(eng).clone())
        torch.cuda.synchronize()
        # ... and the inference forward (two ping-pong buffers instead of one per layer, no z)
        got[chains]["logits"] = eng.decoder_forward(x, c, g, train=False, want_logits=True)["logits"].clone()
        torch.cuda.synchronize()
        lay = eng.lay
        del eng
        torch.cuda.empty_cache()
    one, two = got["1"], got[force]
    bits = lambda t_: t_.contiguous().view(torch.int16 if t_.element_size() == 2 else torch.int32)  # noqa: E731
    assert float(one["dz"].float().abs().max()) > 0 and float(one["u"].float().abs().max()) > 0
    for k in ("u", "nll", "loss", "dz", "dc", "logits"):
        assert torch.equal(bits(one[k]), bits(two[k])), k
    for k in ("x", "z", "gx"):
        for i, (a, b_) in enumerate(zip(one[k], two[k])):
            if k == "x" and i == len(one[k]) - 1:
                continue                                   # (the last layer's x' is dead: never written)
            assert torch.equal(bits(a), bits(b_)), (k, i)
    bad = {}
    for k in lay.offsets:
        a = one["grads"][lay.off(k):lay.off(k) + lay.numel(k)]
        b_ = two["grads"][lay.off(k):lay.off(k) + lay.numel(k)]
        err, ref = float((a - b_).abs().max()), float(a.abs().max())
        if err > 1e-4 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad


@pytest.mark.parametrize("dtype,name", [("fp32", "A"), ("bf16", "A"), ("bf16", "B")])
def test_side_streams_change_nothing(dtype, name):
    """Round 6 (options.py: WAE_SIDE): the step's independent side work on side streams -- the upsampling network + hoisted global
    conditioning and the backward's weight packing (with the clearing of the gradient accumulators) beside the forward weight packing,
    the front end's backward beside the scatter of the layers' weight gradients -- against every launch on one stream.  Three train
    steps each (a hazard between a step's early packing and the previous step's optimizer would show in the second): losses,
    gradients and parameters to the tolerance of two runs of the atomically accumulated weight gradients."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model(name)
    x, c, g = ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda()
    lengths = torch.tensor([x.shape[1], x.shape[1] - 217])
    got = {}
    for side in (False, True):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.opt.side = side
        eng.load_state_dict(sd)
        eng.init_optimizer()
        losses, seen = [], {}
        for _ in range(3):
            r = eng.train_step(x, c, g, lengths=lengths, lr=1e-3, grad_hook=lambda gr: seen.__setitem__("g", gr.clone()))
            losses.append(r["loss"].detach().clone())
        torch.cuda.synchronize()
        assert (eng._ev_wn is not None) == side
        got[side] = (torch.stack(losses).cpu(), seen["g"].cpu(), eng.params.cpu().clone())
        del eng
        torch.cuda.empty_cache()
    (l0, g0, p0), (l1, g1, p1) = got[False], got[True]
    # the first step sees the same weights: the same loss (bit for bit but for the VQ term's atomically summed statistics)
    assert float((l0[:1] - l1[:1]).abs().max()) <= 1e-6 * float(l0[:1].abs().max()), (l0, l1)
    assert float((l0 - l1).abs().max()) < (1e-5 if dtype == "fp32" else 2e-3), (l0, l1)
    assert float((g0 - g1).abs().max()) <= (2e-5 if dtype == "fp32" else 2e-3) * float(g0.abs().max())
    assert float((p0 - p1).abs().max()) < (1e-5 if dtype == "fp32" else 2e-3)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_weight_gradients_beside_an_underfilled_sweep(dtype):
    """Round 6: when the sweep's launches leave a quarter of the CUs or more idle (hps/vqwae.json's shard: 160 workgroups on 256 CUs), the
    weight gradients of the upper layers and the head run BESIDE the lower part of the sweep -- a launch of the same static kernel
    sized for the idle CUs, on a side stream -- and the lower layers' follow the sweep (backward.decoder_backward: `beside`).  Same jobs,
    same regions of the dense gradient tiles: every parameter gradient against the one-launch order (WAE_SIDE=0) to the tolerance of
    two runs of that launch (fp32 atomics arrive in another order), over two consecutive steps."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=10, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=11, upsample_scales=[4, 4, 8, 5], cin_pad=0)
    B, T, hop = 3, 1920, 640
    sd = O.make_state_dict(dict(cfg), salt=9, with_encoder=False)
    gen = torch.Generator().manual_seed(377)
    x = torch.randint(0, 256, (B, T), generator=gen).cuda()
    c = torch.randn(B, 64, T // hop, generator=gen).cuda()
    g = torch.randint(0, cfg["n_speakers"], (B,), generator=gen).cuda()
    lengths = torch.tensor([T - 97 * i for i in range(B)])
    got = {}
    for side in (False, True):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.opt.side = side
        eng.load_state_dict(sd, strict=False)
        eng.init_optimizer()
        seen = []
        for _ in range(2):
            eng.train_step(x, c, g, lengths=lengths, lr=1e-3, grad_hook=lambda gr: seen.append(gr.clone()))
        torch.cuda.synchronize()
        assert ("stream_beside" in eng._ws[("bwd", B, T)]) == side
        got[side] = ([s_.cpu() for s_ in seen], eng.params.cpu().clone())
        lay = eng.lay
        del eng
        torch.cuda.empty_cache()
    for step in range(2):
        a, b_ = got[False][0][step], got[True][0][step]
        bad = {}
        for k in lay.offsets:
            ga, gb = a[lay.off(k):lay.off(k) + lay.numel(k)], b_[lay.off(k):lay.off(k) + lay.numel(k)]
            err, ref = float((ga - gb).abs().max()), float(ga.abs().max())
            if step == 0 and err > 1e-4 * max(ref, 1e-6) + 1e-7:
                bad[k] = (err, ref)
        assert not bad, (step, bad)
        # (the second step starts from weights that differ in their last bits, and 16-bit activations amplify that: norm-wise only)
        assert float((a - b_).norm()) < 5e-2 * float(a.norm()), step
    assert float((got[False][1] - got[True][1]).abs().max()) < 2e-3


def test_weight_gradients_beside_the_sweep_with_the_encoder_in_front():
    """The same on the reference's own preset (hps/vqwae.json in full: encoder, VQ, upsampling network, 20-layer decoder; a short
    batch): there the lower layers' launch also leaves 32 CUs to the front end's backward, which runs on its own side stream from the
    end of the sweep.  Losses, VQ statistics and every parameter gradient of the first step against the one-stream order."""
    import bench
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    conf = dict(bench.CONFIGS["c3"], B=2, T=2560)
    sd = O.make_state_dict(dict(conf["cfg"]), salt=conf["salt"], with_encoder=True)
    x, lat, g = bench.synth_inputs(0, torch.device("cuda:0"), conf)
    got = {}
    for side in (False, True):
        eng = WaeEngine(Geometry.from_cfg(conf["cfg"]), dtype="bf16")
        eng.opt.side = side
        eng.load_state_dict(sd)
        eng.init_optimizer()
        seen = []
        r = eng.train_step(x, lat, g, lr=1e-3, grad_hook=lambda gr: seen.append(gr.clone()))
        torch.cuda.synchronize()
        ws = eng._ws[("bwd", 2, 2560)]
        assert ("stream_beside" in ws) == side
        if side:
            assert ws["stream_below"].nwg < ws["stream"].nwg            # 32 CUs left to the front end's backward
        got[side] = (seen[0].cpu(), float(r["loss"]), float(r["vq_loss"]), float(r["perp"]))
        lay = eng.lay
        del eng
        torch.cuda.empty_cache()
    # the forward is the same launches in the same order per stream: the CE bit for bit; the VQ statistics are atomic sums
    assert got[False][1] == got[True][1]
    assert abs(got[False][2] - got[True][2]) < 1e-5 * abs(got[False][2]) + 1e-9 and abs(got[False][3] - got[True][3]) < 1e-4 * got[False][3]
    a, b_ = got[False][0], got[True][0]
    bad = {}
    for k in lay.offsets:
        ga, gb = a[lay.off(k):lay.off(k) + lay.numel(k)], b_[lay.off(k):lay.off(k) + lay.numel(k)]
        err, ref = float((ga - gb).abs().max()), float(ga.abs().max())
        if err > 1e-4 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad
