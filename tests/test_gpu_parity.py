"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden fixtures.

Tolerances (north star): fp32 mode <= 1e-3 relative on encoder latents and decoder logits, VQ indices
bit-exact.  bf16 mode (bf16 storage of activations/weights, fp32 accumulate) is checked against the same
oracle with a looser, explicitly stated tolerance: 5e-2 of the logit range.
"""
import math

import numpy as np
import pytest
import torch

from helpers import golden_model, load_npz, rel_err
from oracle import wae_oracle as O

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3
BF16_TOL = 5e-2


def _engine(cfg, sd, dtype):
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    eng.load_state_dict(sd)
    return eng


@pytest.mark.parametrize("dtype,tol", [("fp32", FP32_TOL), ("bf16", BF16_TOL)])
@pytest.mark.parametrize("name", ["A", "B"])
def test_full_model_against_golden(name, dtype, tol):
    cfg, sd, ins, z, ocfg = golden_model(name)
    eng = _engine(cfg, sd, dtype)
    x, c, g = ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda()
    T = x.shape[1]
    lengths = torch.tensor([T, T - 137])
    out = eng.forward(x, c, g, targets=x, lengths=lengths.cuda())
    torch.cuda.synchronize()
    assert rel_err(out["latents"].cpu(), z["latents"]) < FP32_TOL          # encoder + VQ are always fp32
    assert np.array_equal(out["idx"].cpu().numpy(), z["vq_idx"])            # bit-exact
    assert rel_err(out["quant"].cpu(), z["quant"]) < 1e-6
    assert abs(float(out["vq_loss"]) - float(z["vq_loss"])) < 1e-4 * max(1.0, float(z["vq_loss"]))
    assert abs(float(out["perp"]) - float(z["perp"])) < 1e-3
    assert rel_err(out["logits"].cpu(), z["y_hat"]) < tol
    ce = O.masked_ce_loss(torch.from_numpy(z["y_hat"]), ins["x"].unsqueeze(-1), lengths)
    assert abs(float(out["loss"]) - float(ce)) < (1e-4 if dtype == "fp32" else 2e-2)


@pytest.mark.parametrize("dtype,tol", [("fp32", FP32_TOL), ("bf16", BF16_TOL)])
def test_upsample_against_golden(dtype, tol):
    cfg, sd, ins, z, ocfg = golden_model("A")
    eng = _engine(cfg, sd, dtype)
    eng.prepare_weights()
    q = torch.from_numpy(z["quant"]).cuda()
    B, Cc, Tq = q.shape
    T = Tq * 640
    out = torch.zeros(B, T, eng.g.Ccp, dtype=eng.tdtype, device="cuda")
    eng.upsample_forward(q, out)
    got = out[:, :, :Cc].float().transpose(1, 2).cpu()
    assert rel_err(got, z["c_up"]) < (1e-5 if dtype == "fp32" else 1e-2)
    assert float(out[:, :, Cc:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
@pytest.mark.parametrize("name", ["A", "B"])
@pytest.mark.parametrize("d", [1, 2, 512])
def test_single_glu_layer(name, d, dtype, tol):
    """wae_glu_layer_fwd alone against the golden single-layer vectors (modules.py:115-163)."""
    import ctypes
    from wavenet_autoencoders_amd import _lib as L
    cfg, sd, ins, zm, ocfg = golden_model(name)
    z = load_npz("glu_" + name)
    eng = _engine(cfg, sd, dtype)
    eng.prepare_weights()
    g = eng.g
    T, B = int(z["T"]), 2
    x = O.hash_fill((B, cfg["R"], T), int(z["x_salt"]), float(z["x_scale"]))
    c = O.hash_fill((B, cfg["Cc"], T), int(z["c_salt"]), float(z["c_scale"]))
    gv = O.hash_fill((B, cfg["Cg"], 1), int(z["g_salt"]), float(z["g_scale"]))
    st = eng.stream()
    xin = torch.zeros(B, T, g.Rp, dtype=eng.tdtype, device="cuda")
    cin = torch.zeros(B, T, g.Ccp, dtype=eng.tdtype, device="cuda")
    L.check(eng.lib.wae_to_btc(L.ptr(x.cuda()), L.ptr(xin), B, g.R, T, g.Rp, eng.dt, st))
    L.check(eng.lib.wae_to_btc(L.ptr(c.cuda()), L.ptr(cin), B, g.Cc, T, g.Ccp, eng.dt, st))
    zb = torch.zeros(B, g.layers, 2 * g.Hp, device="cuda")
    gvec = gv.view(B, -1).contiguous().cuda()
    L.check(eng.lib.wae_gproj_fwd(L.ptr(eng.eff), eng.lay.off("wavenet.conv_layers.0.conv1x1g.weight_v"),
                                  eng.lay.off("wavenet.conv_layers.0.conv.bias"), eng.lay.layer_stride, None, 0,
                                  L.ptr(gvec), L.ptr(zb), B, g.layers, g.G, g.Hp, g.Cg, 0, None, st))
    xout = torch.zeros_like(xin)
    Ku = 2 * g.Hp + 64
    ubuf = torch.full((B, T, Ku), 7.0, dtype=eng.tdtype, device="cuda")
    zsave = torch.zeros(B, T, 2 * g.Hp, dtype=eng.tdtype, device="cuda")
    desc = L.GluDesc(eng.dt, B, T, g.Rp, g.Ccp, g.Hp, g.k, d, L.GLU_SAVE_Z)
    i = 1  # golden uses the weights of conv_layers.1
    es = eng.w_glu.element_size()
    ucol = g.Hp  # store this layer's u at column offset Hp of the wider buffer
    L.check(eng.lib.wae_glu_layer_fwd(ctypes.byref(desc), L.ptr(xin), L.ptr(xout), L.ptr(cin),
                                      ctypes.c_void_p(ubuf.data_ptr() + ucol * es), Ku,
                                      ctypes.c_void_p(zb.data_ptr() + i * 2 * g.Hp * 4), g.layers * 2 * g.Hp, L.ptr(zsave),
                                      ctypes.c_void_p(eng.w_glu.data_ptr() + i * eng.glu_elems * es),
                                      ctypes.c_void_p(eng.b_glu.data_ptr() + i * g.Rp * 4), st))
    torch.cuda.synchronize()
    pt = torch.from_numpy(z["probe_t"])
    xo = xout[:, :, :g.R].float().transpose(1, 2).cpu()[:, :, pt]
    assert rel_err(xo, z[f"xo_d{d}_cg"]) < tol
    # gated activation against the oracle, and the skip 1x1 of it against the golden skip output
    u_ref = O.glu_layer_gate(sd, "wavenet.conv_layers.1.", x, c, gv, d)
    u = ubuf[:, :, ucol:ucol + g.H].float().transpose(1, 2).cpu()
    assert rel_err(u, u_ref) < tol
    w_skip = O.eff_weight(sd, "wavenet.conv_layers.1.conv1x1_skip")
    so = torch.nn.functional.conv1d(u, w_skip, sd["wavenet.conv_layers.1.conv1x1_skip.bias"])[:, :, pt]
    assert rel_err(so, z[f"so_d{d}_cg"]) < tol
    # pad channels stay exactly zero, neighbours of the u slice are untouched, z = pre-activation is consistent
    assert float(xout[:, :, g.R:].abs().max()) == 0.0
    assert float(ubuf[:, :, ucol + g.H:ucol + g.Hp].float().abs().max()) == 0.0
    assert float((ubuf[:, :, :ucol].float() - 7.0).abs().max()) == 0.0
    assert float((ubuf[:, :, ucol + g.Hp:].float() - 7.0).abs().max()) == 0.0
    za = zsave[:, :, :g.H].float()
    zg = zsave[:, :, g.Hp:g.Hp + g.H].float()
    u_from_z = (torch.tanh(za) * torch.sigmoid(zg)).transpose(1, 2).cpu()
    assert rel_err(u_from_z, u_ref) < (tol if dtype == "fp32" else 5e-2)


@pytest.mark.parametrize("dtype,tol", [("fp32", FP32_TOL), ("bf16", BF16_TOL)])
def test_ragged_T_and_wrong_cond_length(dtype, tol):
    """T not a multiple of the 128-step tile; seeded inputs against the oracle; wavenet.py:198-200 error."""
    cfg, sd, ins, z, ocfg = golden_model("B")
    eng = _engine(cfg, sd, dtype)
    B, T = 3, 1000
    x = ((O.hash_fill((B, T), 5) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
    c = O.hash_fill((B, cfg["Cc"], T), 6, 1.0)
    g = torch.tensor([0, 3, 1])
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    sd2 = {k: v for k, v in sd.items() if "upsample_net" not in k}
    with torch.no_grad():
        ref = O.wavenet_forward(sd2, dict(ocfg, upsample_scales=None), xin, c, g)
    out = eng.decoder_forward(x.cuda(), c.cuda(), g.cuda(), c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), ref) < tol
    with pytest.raises(Exception):
        eng.decoder_forward(x.cuda(), c[:, :, :-5].cuda(), g.cuda(), c_is_upsampled=True)


@pytest.mark.parametrize("dtype,tol", [("fp32", FP32_TOL), ("bf16", BF16_TOL)])
@pytest.mark.parametrize("G,R", [(192, 128), (256, 128), (368, 256), (512, 128)])
def test_gate_channel_widths_against_oracle(G, R, dtype, tol):
    """Every gate-channel tiling of csrc/glu_fwd.hip (Hp/32 = 3, 4, 6, 8; one and two GEMM-1 passes) on seeded inputs
    against the oracle's full decoder forward (wavenet.py:164-216)."""
    cfg, sd, ins, z, ocfg = golden_model("B")
    cfg = dict(cfg, G=G, R=R, layers=4, stacks=2)
    sd = O.make_state_dict(cfg, 31)
    ocfg = dict(ocfg, layers=4, stacks=2)
    eng = _engine(cfg, sd, dtype)
    B, T = 2, 700
    x = ((O.hash_fill((B, T), 15) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
    c = O.hash_fill((B, cfg["Cc"], T), 16, 1.0)
    g = torch.tensor([2, 3])
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    sd2 = {k: v for k, v in sd.items() if "upsample_net" not in k}
    with torch.no_grad():
        ref = O.wavenet_forward(sd2, dict(ocfg, upsample_scales=None), xin, c, g)
    out = eng.decoder_forward(x.cuda(), c.cuda(), g.cuda(), c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), ref) < tol


def test_vqwae_fullsize_probe_fp32():
    """hps/vqwae.json geometry (R=G=S=256, L=20), sparse golden probe from the reference."""
    import json
    z = load_npz("model_vqwae_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(cfg, int(z["salt"]))
    eng = _engine(cfg, sd, "fp32")
    c = O.hash_fill((1, 39, 16), 71, 1.7)
    T = 4 * 640
    x = ((O.hash_fill((1, T), 72) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    g = ((O.hash_fill((1,), 73) * 0.5 + 0.5) * 153).long().clamp(0, 152)
    out = eng.forward(x.cuda(), c.cuda(), g.cuda())
    torch.cuda.synchronize()
    assert np.array_equal(out["idx"].cpu().numpy(), z["vq_idx"])
    y = out["logits"].cpu()
    assert rel_err(y[0][:, torch.from_numpy(z["probe_t"])], z["y_probe"]) < FP32_TOL
    assert abs(float(y.double().sum()) - float(z["y_sum"])) < 1e-3 * float(z["y_abs_sum"])


def test_c2_full_size_properties():
    """BASELINE config C2 at full size (24 layers, R256 G368 S256, 8 x 8000 samples; too large for the oracle): properties that
    do not depend on size -- (1) causality: logits at t < t* do not change when the inputs from t* on change (the receptive
    field looks backwards only, wavenet.py:42-60); (2) clips are independent: clip 0 alone gives the logits it gives inside the
    batch, bit for bit (the forward has no cross-clip arithmetic and no atomics); (3) the bf16 engine's teacher-forced loss agrees
    with the fp32 engine's within 2e-2; (4) a repeated forward is bitwise reproducible."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5],
               cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
    B, T = 8, 8000
    gen = torch.Generator().manual_seed(1234)
    x = torch.randint(0, 256, (B, T), generator=gen)
    lat = torch.randn(B, 64, T // 320, generator=gen)
    g = torch.randint(0, 153, (B,), generator=gen)
    losses = {}
    for dtype in ("fp32", "bf16"):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        out = eng.decoder_forward(x.cuda(), lat.cuda(), g.cuda(), targets=x.cuda())
        y = out["logits"].clone()
        losses[dtype] = float(out["loss"])
        out2 = eng.decoder_forward(x.cuda(), lat.cuda(), g.cuda(), targets=x.cuda())
        assert torch.equal(out2["logits"], y), "forward is not reproducible"
        if dtype == "fp32":
            ts = 5000
            x2 = x.clone()
            x2[:, ts:] = (x2[:, ts:] + 37) % 256
            y2 = eng.decoder_forward(x2.cuda(), lat.cuda(), g.cuda())["logits"]
            assert torch.equal(y2[:, :, :ts], y[:, :, :ts]), "a future input changed a past output"
            assert not torch.equal(y2[:, :, ts:], y[:, :, ts:])
            y1 = eng.decoder_forward(x[:1].cuda(), lat[:1].cuda(), g[:1].cuda())["logits"]
            assert torch.equal(y1[0], y[0]), "clip 0 depends on its batch neighbours"
        del eng
        torch.cuda.empty_cache()
    assert abs(losses["bf16"] - losses["fp32"]) < 2e-2, losses


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_glu_workgroup_shapes_agree(dtype):
    """the fused layer kernel's two workgroup shapes (8 waves x 256 steps, the default; 4 waves x 128 steps) contract in the same
    order: identical logits on a model with ragged T"""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import _lib as L
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model("B")
    c_up = torch.from_numpy(z["c_up"])
    T = 515
    outs = []
    for nw in (8, 4):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.glu_flags = L.GLU_WAVES4 if nw == 4 else 0              # wae_glu_desc.flags of every layer launch
        eng.load_state_dict(sd)
        out = eng.decoder_forward(ins["x"][:, :T].cuda(), c_up[:, :, :T].cuda(), ins["g"].cuda(), c_is_upsampled=True)
        torch.cuda.synchronize()
        outs.append(out["logits"].cpu())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("B,T", [(1, 2), (1, 33), (3, 130), (2, 257), (1, 1025), (40, 48)])   # 40 clips: gproj_bwd stages 32 clips per launch
def test_odd_shapes_forward_and_backward(B, T):
    """clips shorter than a tile / a wave / the receptive field, odd batch sizes: logits and loss against the oracle (fp32), and a
    backward pass whose gradients match autograd through the oracle"""
    from wavenet_autoencoders_amd import Geometry, backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model("B")
    x = ((O.hash_fill((B, T), 91) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
    c = O.hash_fill((B, cfg["Cc"], T), 92, 1.1)
    g = torch.arange(B) % cfg["n_speakers"]
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    lengths = torch.tensor([T] + [max(2, T - 1 - 7 * i) for i in range(1, B)])
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("wavenet.") and "upsample_net" not in k}
    y_ref = O.wavenet_forward(psd, dict(ocfg, upsample_scales=None), xin, c, g)
    loss_ref = O.masked_ce_loss(y_ref, x.unsqueeze(-1), lengths)
    loss_ref.backward()
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype="fp32")
    eng.load_state_dict(sd)
    out = eng.decoder_forward(x.cuda(), c.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), train=True, c_is_upsampled=True)
    BW.decoder_backward(eng, x.cuda(), x.cuda(), lengths, g.cuda())
    grads = BW.finish_grads(eng)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), y_ref.detach()) < FP32_TOL
    assert abs(float(out["loss"]) - float(loss_ref.detach())) < 1e-4 * max(1.0, abs(float(loss_ref.detach())))
    bad = {}
    for k, v in psd.items():
        gref = v.grad if v.grad is not None else torch.zeros_like(v)
        got = grads[eng.lay.off(k):eng.lay.off(k) + eng.lay.numel(k)].view(eng.lay.shapes[k]).cpu()
        err, ref = float((got - gref).abs().max()), float(gref.abs().max())
        if err > 1e-3 * max(ref, 1e-6) + 1e-7:
            bad[k] = (err, ref)
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_glu_64_column_shape_is_bitwise_identical(dtype):
    """WAE_GLU_CG2 (4 waves x 64 columns, every weight fragment feeds two MFMAs; an A/B shape, measured slower than the default):
    the same contraction order -> identical logits, loss and saved pre-activations on a model with ragged T."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import _lib as L
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, z, ocfg = golden_model("B")
    c_up = torch.from_numpy(z["c_up"])
    T = 515
    outs = []
    for flags in (0, L.GLU_CG2):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.glu_flags = flags
        eng.load_state_dict(sd)
        out = eng.decoder_forward(ins["x"][:, :T].cuda(), c_up[:, :, :T].cuda(), ins["g"].cuda(), targets=ins["x"][:, :T].cuda(),
                                  c_is_upsampled=True, train=True)
        torch.cuda.synchronize()
        outs.append((out["logits"].cpu(), float(out["loss"]), [zz.clone() for zz in eng._ws[(2, T, True)]["z"]]))
    assert torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
    assert all(torch.equal(a, b) for a, b in zip(outs[0][2], outs[1][2]))


@pytest.mark.parametrize("R,G", [(256, 368), (256, 256), (512, 512)])
def test_glu_barrier_on_every_second_chunk_is_bitwise_identical(R, G, monkeypatch):
    """WAE_GLU_PAIR (the default of inference launches): GEMM 1 meets at a workgroup barrier on even weight chunks only.  A timing
    device -- the logits of a few layers at C2 / hps/vqwae.json / C5 widths must not differ by a bit from the per-chunk barriers,
    on ragged shapes and on repeats (a race would show as run-to-run differences)."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=6, stacks=2, R=R, G=G, S=R, O=256, Cc=64, Cg=32, k=3, n_speakers=10, upsample_scales=[4, 4, 4, 5], cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
    ref = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("WAE_GLU_PAIR", mode)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype="bf16")
        eng.load_state_dict(sd)
        for B, F in ((1, 3), (3, 13), (8, 9)):
            T = F * 320
            x = torch.randint(0, 256, (B, T), generator=torch.Generator().manual_seed(B * 100 + F)).to(torch.int32).cuda()
            lat = torch.randn(B, 64, F, generator=torch.Generator().manual_seed(7)).cuda()
            g = torch.randint(0, 10, (B,), generator=torch.Generator().manual_seed(9)).cuda()
            for rep in range(1 if mode == "0" else 4):
                y = eng.decoder_forward(x, lat, g)["logits"]
                torch.cuda.synchronize()
                if mode == "0":
                    ref[(B, F)] = y.clone()
                else:
                    assert torch.equal(y, ref[(B, F)]), (B, F, rep)


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("name", ["A", "B"])
def test_split_head_agrees_with_the_one_kernel_head(name, dtype, monkeypatch):
    """16-bit engines run the head's skip contraction as a wae_gemm_tm launch (mode 3) and the rest through wae_head_fwd_from_h0; the
    one-kernel wae_head_fwd (WAE_HEAD_SPLIT=0) is the same arithmetic with the skip-bias sum added before instead of after the
    contraction: logits within 1e-2 of the logit range, the loss within 2e-3, and both within tolerance of the oracle."""
    cfg, sd, ins, z, ocfg = golden_model(name)
    x, c, g = ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda()
    outs = []
    for split in ("1", "0"):
        monkeypatch.setenv("WAE_HEAD_SPLIT", split)
        eng = _engine(cfg, sd, dtype)
        assert eng.split_head == (split == "1")
        out = eng.forward(x, c, g, targets=x, lengths=None)
        torch.cuda.synchronize()
        outs.append((out["logits"].float().cpu(), float(out["loss"])))
        assert rel_err(outs[-1][0], z["y_hat"]) < BF16_TOL
    assert rel_err(outs[0][0], outs[1][0]) < 1e-2
    assert abs(outs[0][1] - outs[1][1]) < 2e-3


@pytest.mark.parametrize("k", [1, 2, 4])
def test_kernel_sizes_against_oracle(k):
    """kernel_size 1, 2 and 4 (modules.py:71-107 accepts any; every preset uses 3): teacher-forced logits, loss, every parameter
    gradient (fp32: 1e-3 of each tensor's range) and teacher-forced incremental decoding against the oracle; 16-bit logits at the
    bf16 / fp16 bounds.  (T = 640: no ReLU pre-activation of this closed-form model sits within rounding of zero there -- at other
    lengths one flipped mask moves a head-bias gradient by ~1/sqrt(T), in the oracle's favour or ours.)"""
    from wavenet_autoencoders_amd import Geometry, backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=6, stacks=2, R=128, G=192, S=128, O=256, Cc=64, Cg=32, k=k, n_speakers=7, upsample_scales=None)
    sd = O.make_state_dict(dict(cfg), 11, with_encoder=False)
    B, T = 2, 640
    x = ((O.hash_fill((B, T), 91) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    c = O.hash_fill((B, 64, T), 92, 1.1)
    g = torch.arange(B) % 7
    xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
    lengths = torch.tensor([T, T - 50])
    psd = {kk: v.clone().requires_grad_(True) for kk, v in sd.items() if kk.startswith("wavenet.")}
    y_ref = O.wavenet_forward(psd, dict(cfg), xin, c, g)
    loss_ref = O.masked_ce_loss(y_ref, x.unsqueeze(-1), lengths)
    loss_ref.backward()
    y_ref = y_ref.detach()
    for dtype, tol in (("fp32", FP32_TOL), ("bf16", BF16_TOL), ("fp16", 1e-2)):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        out = eng.decoder_forward(x.cuda(), c.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), train=True, c_is_upsampled=True)
        BW.decoder_backward(eng, x.cuda(), x.cuda(), lengths, g.cuda())
        grads = BW.finish_grads(eng)
        ar = eng.incremental_forward(c[:, :, :96].contiguous().cuda(), g.cuda(), 96, mode="logits", test_inputs=x[:, :96].cuda(),
                                     c_is_upsampled=True, want_logits=True)
        torch.cuda.synchronize()
        assert rel_err(out["logits"].cpu(), y_ref) < tol
        assert rel_err(ar["logits"].cpu(), y_ref[:, :, :96]) < tol
        assert abs(float(out["loss"]) - float(loss_ref.detach())) < (1e-4 if dtype == "fp32" else 2e-2)
        if dtype == "fp32":      # (the 16-bit weight-gradient launches at these kernel sizes: test_static_weight_gradient_launch_at_other_kernel_sizes)
            bad = {}
            for kk, v in psd.items():
                gref = v.grad if v.grad is not None else torch.zeros_like(v)
                got = grads[eng.lay.off(kk):eng.lay.off(kk) + eng.lay.numel(kk)].view(eng.lay.shapes[kk]).cpu()
                err, ref = float((got - gref).abs().max()), float(gref.abs().max())
                if err > 1e-3 * max(ref, 1e-6) + 1e-7:
                    bad[kk] = (err, ref)
            assert not bad, bad


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("k", [1, 2, 4])
def test_static_weight_gradient_launch_at_other_kernel_sizes(k, dtype, monkeypatch):
    """Round-4 advisor finding: with k = 1, 2, 4 the 16-bit backward runs wae_gemm_tn_static with teams of 3, 4 and 6 members (null-job
    padding of the head group, four tap jobs, the head group riding at k = 4 and staying on the tile launches at k < 3) and nothing
    compared its output.  Same operands, same 16-bit activations: the static launch against the per-layer 128 x 128 tile launches
    (WAE_TN_STREAM=0) to 2e-4 of each tensor's range, the bound of test_static_weight_gradient_launch_at_odd_shapes."""
    from wavenet_autoencoders_amd import Geometry, backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=6, stacks=2, R=128, G=192, S=128, O=256, Cc=64, Cg=32, k=k, n_speakers=7, upsample_scales=None)
    sd = O.make_state_dict(dict(cfg), 11, with_encoder=False)
    B, T = 3, 640
    x = ((O.hash_fill((B, T), 91) * 0.5 + 0.5) * 256).long().clamp(0, 255).cuda()
    c = O.hash_fill((B, 64, T), 92, 1.1).cuda()
    g = (torch.arange(B) % 7).cuda()
    lengths = torch.tensor([T, T - 50, T - 333])
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("WAE_TN_STREAM", mode)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        eng.decoder_forward(x, c, g, targets=x, lengths=lengths.cuda(), train=True, c_is_upsampled=True, want_logits=False)
        BW.decoder_backward(eng, x, x, lengths, g)
        st = BW.bwd_workspace(eng, B, T)["stream"]
        assert (st is None) if mode == "0" else (isinstance(st, BW.StaticStreamTable) and st.team_size == k + 2)
        assert mode == "0" or BW.static_head(eng, B, T) == (k >= 3)
        got[mode] = BW.finish_grads(eng).clone()
        torch.cuda.synchronize()
    lay = eng.lay
    bad = {}
    for kk in lay.offsets:
        a = got["0"][lay.off(kk):lay.off(kk) + lay.numel(kk)]
        b = got["1"][lay.off(kk):lay.off(kk) + lay.numel(kk)]
        err, ref = float((a - b).abs().max()), float(a.abs().max())
        if err > 2e-4 * max(ref, 1e-6) + 1e-7:
            bad[kk] = (err, ref)
    assert not bad, bad
