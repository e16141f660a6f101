"""GPU parity of the configurations bench.py times (BASELINE.json configs C1/C3, C4, C5) at their FULL geometry, against
vectors the reference itself produced (tests/golden/make_golden.py: ar_c4, train_vqwae, model_c5_probe).

C4  hps/vqwae.json synthesis decoder (20 layers, R = G = S = 256, dilations to 512), T = 2560: the per-layer history rings
    (2 * 512 + 1 rows) wrap twice; 32-CU cooperative kernel and one-CU kernel, fp32 and bf16.
C1/C3  hps/vqwae.json in full (302 tensors), 2 x 5120 samples: one train step, every parameter gradient.
C5  48 layers / 4 stacks (dilations to 2048), R = G = S = 512: logits of one 5120-sample clip; causality and clip independence
    of the 16 x 5120 shard.
"""
import json

import numpy as np
import pytest
import torch

from helpers import load_npz, rel_err
from oracle import wae_oracle as O

pytestmark = pytest.mark.gpu


def _engine(cfg, sd, dtype):
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    eng.load_state_dict(sd)
    return eng


# ------------------------------------------------------------------------------------------------------------------ C4
@pytest.fixture(scope="module")
def c4():
    z = load_npz("ar_c4")
    cfg = {k: v for k, v in json.loads(str(z["cfg"])).items() if k not in ("encoder_hid", "c_in", "K")}   # decoder only
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    return cfg, sd, z


def _set_ar_path(monkeypatch, coop):
    """'1': the cooperative kernel (this geometry: the one with the sizes as constants), 'fused': the same with one hand-over per layer
    (wae_ar_generate_coop_fused with the host-formed W1_cur . W_out products), 'generic': the any-shape cooperative kernel on the same
    geometry, '0': one CU per utterance"""
    monkeypatch.setenv("WAE_AR_COOP", "0" if coop == "0" else "1")
    return dict(generic=coop == "generic", one_handover=coop == "fused")      # WaeEngine.ar_path(**...)


@pytest.mark.parametrize("coop", ["1", "fused", "generic", "0"])
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_c4_teacher_forced_logits(c4, dtype, tol, coop, monkeypatch):
    """Teacher-forced incremental decode over 2560 samples == the reference's incremental_forward(test_inputs) at the probe
    steps (every 13th + the steps around both wraps of the d = 512 rings), and the log-sum-exp of EVERY step."""
    cfg, sd, z = c4
    path = _set_ar_path(monkeypatch, coop)          # (WAE_AR_COOP is read when the engine is built)
    eng = _engine(cfg, sd, dtype).ar_path(**path)
    T = z["x"].shape[1]
    x = torch.from_numpy(z["x"].astype(np.int64)).cuda()
    out = eng.incremental_forward(torch.from_numpy(z["lat"]).cuda(), torch.from_numpy(z["g"]).cuda(), T, mode="logits",
                                  test_inputs=x)
    torch.cuda.synchronize()
    y = out["logits"].cpu()[0]                                     # (O, T)
    pt = torch.from_numpy(z["probe_t"]).long()
    assert rel_err(y[:, pt], z["tf_probe"]) < tol                  # norm-wise: max|a-b| / max|b|
    lse = torch.logsumexp(y, 0)
    assert float((lse - torch.from_numpy(z["tf_lse"])[0]).abs().max()) < tol * float(np.abs(z["tf_probe"]).max())
    if dtype == "fp32":
        # the most likely class of every step, except where the reference's own two best logits are closer than 1e-3
        am = y.argmax(0).numpy()
        top2 = torch.from_numpy(z["tf_probe"]).topk(2, dim=0)[0]
        clear = ((top2[0] - top2[1]) > 1e-3).numpy()
        assert (am[pt.numpy()] == z["tf_argmax"][0][pt.numpy()])[clear].all()


def _check_rollout(got, want, margin, thresh, what):
    """free-running sequences can only part where the reference's own decision was within rounding (margin < thresh);
    everything before the first such step must be identical"""
    diff = np.nonzero(got != want)[0]
    if diff.size == 0:
        return len(want)
    first = int(diff[0])
    assert margin[first] < thresh, f"{what}: first difference at step {first} where the reference's margin is {margin[first]:.3e}"
    return first


@pytest.mark.parametrize("coop", ["1", "fused", "generic"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_c4_cooperative_decode_is_bitwise_reproducible(c4, dtype, coop, monkeypatch):
    """The members' shares are added in a fixed order (csrc/ar_coop.hip: arc_allsum / arc_allsum2), not by atomics in arrival order:
    two runs give the same bits, logits and drawn samples alike."""
    cfg, sd, z = c4
    path = _set_ar_path(monkeypatch, coop)          # (WAE_AR_COOP is read when the engine is built)
    eng = _engine(cfg, sd, dtype).ar_path(**path)
    T = z["x"].shape[1]
    lat, g = torch.from_numpy(z["lat"]).cuda(), torch.from_numpy(z["g"]).cuda()
    uni = torch.rand(1, T, generator=torch.Generator().manual_seed(5)).cuda()
    runs = []
    for _ in range(2):
        out = eng.incremental_forward(lat, g, T, mode="sample", init_idx=127, uniforms=uni, want_logits=True)
        torch.cuda.synchronize()
        runs.append((out["idx"].cpu().clone(), out["logits"].cpu().clone()))
    assert torch.equal(runs[0][0], runs[1][0])
    assert torch.equal(runs[0][1], runs[1][1])


@pytest.mark.parametrize("coop", ["1", "fused", "generic", "0"])
def test_c4_greedy_rollout_fp32(c4, coop, monkeypatch):
    cfg, sd, z = c4
    path = _set_ar_path(monkeypatch, coop)          # (WAE_AR_COOP is read when the engine is built)
    eng = _engine(cfg, sd, "fp32").ar_path(**path)
    T = z["x"].shape[1]
    lat, g = torch.from_numpy(z["lat"]).cuda(), torch.from_numpy(z["g"]).cuda()
    out = eng.incremental_forward(lat, g, T, mode="argmax", init_idx=127)
    torch.cuda.synchronize()
    n = _check_rollout(out["idx"].cpu().numpy()[0], z["greedy"][0], z["greedy_margin"][0], 1e-3, "greedy")
    assert n >= 300, f"greedy roll-out left the reference's after {n} steps"
    # whatever happened after a near-tie: teacher-forced on the reference's own sequence, the engine picks the reference's next
    # sample at every step whose margin is clear
    seq = torch.from_numpy(np.concatenate([[127], z["greedy"][0][:-1]]).astype(np.int64))[None].cuda()
    tf = eng.incremental_forward(lat, g, T, mode="logits", test_inputs=seq)
    torch.cuda.synchronize()
    am = tf["logits"].cpu()[0].argmax(0).numpy()
    clear = z["greedy_margin"][0] > 1e-3
    assert (am == z["greedy"][0])[clear].all()
    assert clear.mean() > 0.95


def test_c4_sampled_rollout_and_partial_forcing_fp32(c4):
    cfg, sd, z = c4
    eng = _engine(cfg, sd, "fp32")
    T = z["x"].shape[1]
    lat, g = torch.from_numpy(z["lat"]).cuda(), torch.from_numpy(z["g"]).cuda()
    u = (O.hash_fill((1, T), int(z["u_salt"])) * 0.5 + 0.5).cuda()
    out = eng.incremental_forward(lat, g, T, mode="sample", uniforms=u, init_idx=127)
    torch.cuda.synchronize()
    n = _check_rollout(out["idx"].cpu().numpy()[0], z["sampled"][0], z["cdf_gap"][0], 1e-5, "inverse-CDF draw")
    assert n >= 400, f"sampled roll-out left the reference's after {n} steps"
    # test_inputs shorter than T (wavenet.py:300-305): 700 forced steps, then greedy feedback
    nf, Tp = int(z["partial_forced"]), z["partial"].shape[1]
    x = torch.from_numpy(z["x"].astype(np.int64))[:, :nf].cuda()
    out = eng.incremental_forward(lat[:, :, :Tp // 640].contiguous(), g, Tp, mode="argmax", test_inputs=x, n_forced=nf)
    torch.cuda.synchronize()
    got, want, mg = out["idx"].cpu().numpy()[0], z["partial"][0], z["partial_margin"][0]
    assert (got[:nf] == want[:nf])[mg[:nf] > 1e-3].all()             # forced steps: independent decisions
    n = _check_rollout(got[nf:], want[nf:], mg[nf:], 1e-3, "free-running tail")
    assert n >= 100


def test_c4_bf16_follows_the_reference_sequence(c4):
    """bf16 storage cannot reproduce a 2560-step greedy path bit for bit; teacher-forced on the reference's path it must make
    the reference's decision wherever the two best logits are further apart than twice the stated bf16 bound (5 % of the
    logit range on each)."""
    cfg, sd, z = c4
    eng = _engine(cfg, sd, "bf16")
    T = z["x"].shape[1]
    lat, g = torch.from_numpy(z["lat"]).cuda(), torch.from_numpy(z["g"]).cuda()
    seq = torch.from_numpy(np.concatenate([[127], z["greedy"][0][:-1]]).astype(np.int64))[None].cuda()
    tf = eng.incremental_forward(lat, g, T, mode="logits", test_inputs=seq)
    torch.cuda.synchronize()
    am = tf["logits"].cpu()[0].argmax(0).numpy()
    bound = 5e-2 * float(np.abs(z["tf_probe"]).max())              # the stated bf16 tolerance on one logit
    clear = z["greedy_margin"][0] > 2 * bound
    assert clear.sum() > 300 and (am == z["greedy"][0])[clear].all()


# --------------------------------------------------------------------------------------------------------------- C1 / C3
def test_vqwae_full_geometry_train_step_fp32():
    """hps/vqwae.json, 2 clips x 5120 samples, ragged lengths: loss terms, VQ indices (bit-exact), every one of the 302
    parameter gradients at its probes and by its norm, the clip norm, post-Adam parameters and the EMA shadow -- against
    the reference's own step (fp32) and the fp64 oracle of the same step."""
    z = load_npz("train_vqwae")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]))
    eng = _engine(cfg, sd, "fp32")
    B, F = 2, 32
    s = int(z["in_salt"])
    c = O.hash_fill((B, cfg["c_in"], F), s + 1, 1.7)
    T = 5120
    x = ((O.hash_fill((B, T), s + 2) * 0.5 + 0.5) * cfg["O"]).long().clamp(0, cfg["O"] - 1)
    g = ((O.hash_fill((B,), s + 3) * 0.5 + 0.5) * cfg["n_speakers"]).long().clamp(0, cfg["n_speakers"] - 1)
    lengths = torch.from_numpy(z["lengths"])
    eng.init_optimizer()
    grads_seen = {}

    def hook(grads):
        grads_seen["g"] = grads.clone()

    res = eng.train_step(x.cuda(), c.cuda(), g.cuda(), lengths=lengths.cuda(), lr=4e-4, clip_thresh=100.0, ema_decay=0.9999,
                         grad_hook=hook)
    torch.cuda.synchronize()
    assert np.array_equal(eng._fe["idx"].cpu().numpy(), z["vq_idx"])
    assert abs(float(res["ce"]) - float(z["ce"])) < 1e-4 * float(z["ce"])
    assert abs(float(res["vq_loss"]) - float(z["vq_loss"])) < 1e-5 * max(1.0, float(z["vq_loss"]))
    assert abs(float(res["perp"]) - float(z["perp"])) < 1e-3
    assert abs(float(res["grad_norm"]) - float(z["grad_norm"])) < 1e-3 * float(z["grad_norm"])
    names = json.loads(str(z["names"]))
    counts = z["probe_counts"]
    grads = grads_seen["g"].cpu()
    params, shadow = eng.params.cpu(), eng.shadow.cpu()
    off, bad = 0, {}
    for i, k in enumerate(names):
        n = eng.lay.numel(k)
        idx = np.unique(np.concatenate([np.arange(min(4, n)), (np.arange(24, dtype=np.int64) * 2654435761 + 12345) % n]))
        assert len(idx) == counts[i]
        sl = slice(off, off + len(idx))
        off += len(idx)
        o = eng.lay.off(k)
        gk = grads[o:o + n]
        gmax = float(z["grad_max"][i])
        e64 = float((gk[idx].double() - torch.from_numpy(z["grad64_probe"][sl])).abs().max())
        e32 = float((gk[idx] - torch.from_numpy(z["grad_probe"][sl])).abs().max())
        sq = float((gk.double() ** 2).sum())
        # 2e-3 of the tensor's largest gradient against fp64: bias / FIR / weight_g gradients are sums of ~1e4..1e5 signed fp32
        # terms formed by atomics in arrival order (observed 2e-4..1e-3, varying from run to run; the reference's own fp32 step
        # is within 2.2e-4 of fp64), norms to 4e-3
        # (weight_g gradients are cancelling sums over a row of dW . v -- largest ~1e-5..1e-4 where the weight_v gradients they are
        # formed from are ~1e-3 -- hence the absolute floor: 1e-7 is 1e-4 of the summands)
        floor = 1e-7 if k.endswith("weight_g") else 2e-8
        if e64 > 2e-3 * gmax + floor or e32 > 3e-3 * gmax + floor or abs(sq - z["grad64_sq"][i]) > 4e-3 * z["grad64_sq"][i] + 1e-12:
            bad[k] = (e64, e32, gmax, sq, float(z["grad64_sq"][i]))
        # the first Adam step moves every weight by lr * g / (|g| + eps): where |g| is not far above eps = 1e-8 (weight_g
        # rows whose gradient cancels to ~0) the step inherits the gradient's relative error -- at most 2 lr, when the sign of a
        # ~1e-9 gradient differs
        pk, sk = params[o:o + n], shadow[o:o + n]
        gr = torch.from_numpy(z["grad_probe"][sl])
        tol = 8e-4 * torch.clamp(4 * (gk[idx] - gr).abs() / (gr.abs() + 1e-8), max=1.0) + 2e-6
        assert bool(((pk[idx] - torch.from_numpy(z["new_probe"][sl])).abs() <= tol).all()), k
        assert bool(((sk[idx] - torch.from_numpy(z["ema_probe"][sl])).abs() <= 1e-4 * tol + 1e-6).all()), k
    assert not bad, bad


# ------------------------------------------------------------------------------------------------------------------ C5
@pytest.fixture(scope="module")
def c5():
    z = load_npz("model_c5_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    return cfg, sd, z


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_c5_full_depth_logits(c5, dtype, tol):
    cfg, sd, z = c5
    eng = _engine(cfg, sd, dtype)
    T = int(z["T"])
    lat = O.hash_fill((1, cfg["Cc"], T // 640), int(z["lat_salt"]), 1.2)
    x = ((O.hash_fill((1, T), int(z["x_salt"])) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    out = eng.decoder_forward(x.cuda(), lat.cuda(), torch.from_numpy(z["g"]).cuda())
    torch.cuda.synchronize()
    y = out["logits"].cpu()[0]
    pt = torch.from_numpy(z["probe_t"]).long()
    assert rel_err(y[:, pt], z["y_probe"]) < tol
    scale = float(np.abs(z["y_probe"]).max())
    assert float((torch.logsumexp(y, 0) - torch.from_numpy(z["y_lse"])[0]).abs().max()) < tol * scale
    if dtype == "fp32":
        assert abs(float(y.double().sum()) - float(z["y_sum"])) < 1e-4 * float(z["y_abs_sum"])


def test_c5_shard_properties_bf16(c5):
    """The per-GPU shard of C5 (16 clips x 5120 samples, bf16): causality through all 48 layers (a change at t0 leaves every
    earlier step bit-identical), independence of the clips, and the bf16 loss against the fp32 engine."""
    cfg, sd, z = c5
    B, T = 16, 5120
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.integers(0, 256, size=(B, T))).cuda()
    lat = torch.from_numpy(rng.standard_normal((B, cfg["Cc"], T // 640)).astype(np.float32)).cuda()
    g = torch.from_numpy(rng.integers(0, cfg["n_speakers"], size=(B,))).cuda()
    eng = _engine(cfg, sd, "bf16")
    base = eng.decoder_forward(x, lat, g, targets=x)
    y0, l0 = base["logits"].clone(), float(base["loss"])
    t0 = 4097
    x2 = x.clone()
    x2[3, t0:] = (x2[3, t0:] + 101) % 256                          # clip 3 changes from t0 on
    y1 = eng.decoder_forward(x2, lat, g)["logits"]
    torch.cuda.synchronize()
    assert torch.equal(y1[3, :, :t0], y0[3, :, :t0]), "a later input changed an earlier output"
    assert not torch.equal(y1[3, :, t0 + 1:], y0[3, :, t0 + 1:])
    others = [b for b in range(B) if b != 3]
    assert torch.equal(y1[others], y0[others]), "clips are not independent"
    # the same clips alone (B = 2) give the same rows as inside the shard
    y2 = eng.decoder_forward(x[5:7].contiguous(), lat[5:7].contiguous(), g[5:7].contiguous())["logits"]
    torch.cuda.synchronize()
    assert torch.equal(y2, y0[5:7])
    del eng
    e32 = _engine(cfg, sd, "fp32")
    l32 = float(e32.decoder_forward(x[:4].contiguous(), lat[:4].contiguous(), g[:4].contiguous(), targets=x[:4].contiguous(),
                                    want_logits=False)["loss"])
    eb = _engine(cfg, sd, "bf16")
    lb = float(eb.decoder_forward(x[:4].contiguous(), lat[:4].contiguous(), g[:4].contiguous(), targets=x[:4].contiguous(),
                                  want_logits=False)["loss"])
    assert abs(lb - l32) < 2e-2 * l32, (lb, l32)
    assert np.isfinite(l0)


# ------------------------------------------------------------------------------------------------------------------ C2
@pytest.fixture(scope="module")
def c2():
    z = load_npz("model_c2_probe")
    cfg = json.loads(str(z["cfg"]))
    sd = O.make_state_dict(dict(cfg), int(z["salt"]), with_encoder=False)
    T = int(z["T"])
    lat = O.hash_fill((1, cfg["Cc"], T // 320), int(z["lat_salt"]), 1.2)
    x = ((O.hash_fill((1, T), int(z["x_salt"])) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    return cfg, sd, z, x, lat


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 5e-2), ("fp16", 1e-2)])
def test_c2_logits_against_the_reference(c2, dtype, tol):
    """BASELINE config C2 -- the geometry the bench line is measured on (24 layers / 2 stacks, R 256, G 368, S 256, Cc 64, Cg 64,
    scales 4,4,4,5; Hp = 192 runs the one-pass static-schedule layer kernel in the 16-bit modes) -- one 8000-sample clip against
    the logits of the reference's own WaveNet (tests/golden/model_c2_probe.npz): 1e-3 fp32 (north star), 5e-2 bf16, 1e-2 fp16."""
    cfg, sd, z, x, lat = c2
    eng = _engine(cfg, sd, dtype)
    out = eng.decoder_forward(x.cuda(), lat.cuda(), torch.from_numpy(z["g"]).cuda(), targets=x.cuda())
    torch.cuda.synchronize()
    y = out["logits"].cpu()[0]
    pt = torch.from_numpy(z["probe_t"]).long()
    assert rel_err(y[:, pt], z["y_probe"]) < tol
    scale = float(np.abs(z["y_probe"]).max())
    assert float((torch.logsumexp(y, 0) - torch.from_numpy(z["y_lse"])[0]).abs().max()) < tol * scale
    assert abs(float(out["loss"]) - float(z["loss"])) < (1e-4 if dtype == "fp32" else 2e-2) * float(z["loss"])
    if dtype == "fp32":
        assert abs(float(y.double().sum()) - float(z["y_sum"])) < 1e-4 * float(z["y_abs_sum"])


def test_c2_train_step_gradients_fp32(c2):
    """One C2 clip, one train step: every decoder parameter gradient of the fp32 engine at the fixture's probes against the
    fp64 oracle's autograd (2e-3 of the tensor's largest gradient, the bound the reference's own fp32 autograd meets with 7e-3
    on this model) and by its norm."""
    from wavenet_autoencoders_amd import backward as BW
    cfg, sd, z, x, lat = c2
    eng = _engine(cfg, sd, "fp32")
    xs, g = x.cuda(), torch.from_numpy(z["g"]).cuda()
    out = eng.decoder_forward(xs, lat.cuda(), g, targets=xs, train=True, want_logits=False)
    assert abs(float(out["loss"]) - float(z["loss"])) < 1e-4 * float(z["loss"])
    dc = BW.decoder_backward(eng, xs, xs, None, g)
    BW.frontend_backward(eng, dc)
    grads = BW.finish_grads(eng).cpu()
    names = json.loads(str(z["names"]))
    off, bad = 0, {}
    for i, k in enumerate(names):
        n = eng.lay.numel(k)
        idx = np.unique(np.concatenate([np.arange(min(4, n)), (np.arange(24, dtype=np.int64) * 2654435761 + 12345) % n]))
        assert len(idx) == z["probe_counts"][i]
        sl = slice(off, off + len(idx))
        off += len(idx)
        o = eng.lay.off(k)
        gk = grads[o:o + n]
        gmax = float(z["grad_max"][i])
        e64 = float((gk[idx].double() - torch.from_numpy(z["grad64_probe"][sl])).abs().max())
        sq = float((gk.double() ** 2).sum())
        floor = 1e-7 if k.endswith("weight_g") else 2e-8
        if e64 > 2e-3 * gmax + floor or abs(sq - z["grad64_sq"][i]) > 6e-3 * z["grad64_sq"][i] + 1e-12:
            bad[k] = (e64, gmax, sq, float(z["grad64_sq"][i]))
    assert not bad, bad


@pytest.mark.parametrize("dtype,tol", [("bf16", 8e-2), ("fp16", 4e-2)])
def test_c2_train_step_gradients_16bit(c2, dtype, tol):
    """The same step in the throughput modes (the stream-K weight-gradient launch, the static layer kernel with z saved): every
    tensor's gradient at the probes within tol of its largest entry of the fp64 oracle's."""
    from wavenet_autoencoders_amd import backward as BW
    cfg, sd, z, x, lat = c2
    eng = _engine(cfg, sd, dtype)
    xs, g = x.cuda(), torch.from_numpy(z["g"]).cuda()
    eng.decoder_forward(xs, lat.cuda(), g, targets=xs, train=True, want_logits=False)
    dc = BW.decoder_backward(eng, xs, xs, None, g)
    BW.frontend_backward(eng, dc)
    grads = BW.finish_grads(eng).cpu()
    names = json.loads(str(z["names"]))
    off, bad = 0, {}
    for i, k in enumerate(names):
        n = eng.lay.numel(k)
        idx = np.unique(np.concatenate([np.arange(min(4, n)), (np.arange(24, dtype=np.int64) * 2654435761 + 12345) % n]))
        sl = slice(off, off + len(idx))
        off += len(idx)
        gk = grads[eng.lay.off(k):eng.lay.off(k) + n]
        e = float((gk[idx].double() - torch.from_numpy(z["grad64_probe"][sl])).abs().max())
        # weight_g gradients are cancelling row sums of dW * v / |v| (largest ~1e-4 here where the weight_v gradients they are formed
        # from reach ~1e-2): their error follows the 16-bit rounding of the summands, hence the absolute floor
        floor = 2e-5 if k.endswith("weight_g") else 1e-6
        if e > tol * float(z["grad_max"][i]) + floor:
            bad[k] = (e, float(z["grad_max"][i]))
    assert not bad, bad


# ------------------------------------------------------------------------------------------------------------ cin_pad = 1
def test_cin_pad_one_forward_and_backward():
    """cin_pad = 1 (upsample.py:69-85: conv_in with 2*cin_pad+1 taps and no padding; features cin_pad frames wider than the audio,
    vqwae_train.py:455-478): VQ indices, c_up and logits of the whole model against the reference's own (model_P.npz), then one
    train step's gradients -- conv_in's three-tap weight among them -- against autograd through the oracle."""
    from helpers import golden_model
    cfg, sd, ins, z, _ = golden_model("P")
    assert cfg["cin_pad"] == 1 and sd["wavenet.upsample_net.conv_in.weight"].shape[-1] == 3
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=1)
    eng = _engine(cfg, sd, "fp32")
    x, c, g = ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda()
    out = eng.forward(x, c, g, targets=x)
    torch.cuda.synchronize()
    assert np.array_equal(out["idx"].cpu().numpy(), z["vq_idx"])
    assert rel_err(out["logits"].cpu(), z["y_hat"]) < 1e-3
    T = x.shape[1]
    assert T == (z["latents"].shape[-1] - 2) * int(np.prod(cfg["upsample_scales"]))
    # gradients of one train step
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y_o, vq_o, _, _ = O.vqvae_forward(psd, ocfg, ins["xin"], ins["c"], ins["g"])
    (O.masked_ce_loss(y_o, ins["x"].unsqueeze(-1), torch.full((2,), T)) + vq_o).backward()
    eng.init_optimizer()
    seen = {}
    eng.train_step(x, c, g, lr=0.0, clip_thresh=-1.0, grad_hook=lambda gr: seen.update(g=gr.clone()))
    torch.cuda.synchronize()
    grads = seen["g"].cpu()
    for k in ("wavenet.upsample_net.conv_in.weight", "wavenet.upsample_net.upsample.up_layers.1.weight_v", "encoder.lin.weight",
              "vq.embedding.weight", "wavenet.conv_layers.0.conv1x1c.weight_v", "wavenet.first_conv.weight_v"):
        gk = grads[eng.lay.off(k):eng.lay.off(k) + eng.lay.numel(k)].view(eng.lay.shapes[k])
        assert rel_err(gk, psd[k].grad) < 2e-3, k


def test_cin_pad_one_module_decoder_and_feature_gradient():
    """The drop-in WaveNet with upsample_params cin_pad=1 on (B, Cc, Tc) features: logits and the gradient w.r.t. the features
    (through conv_in's three taps) against the reference's own (model_P.npz: y_dec_probe, dfeats)."""
    from helpers import golden_model
    from wavenet_autoencoders_amd.wavenet_vocoder import WaveNet
    cfg, sd, ins, z, _ = golden_model("P")
    wn = WaveNet(out_channels=cfg["O"], layers=cfg["layers"], stacks=cfg["stacks"], residual_channels=cfg["R"],
                 gate_channels=cfg["G"], skip_out_channels=cfg["S"], kernel_size=cfg["k"], dropout=0.0, cin_channels=cfg["Cc"],
                 gin_channels=cfg["Cg"], n_speakers=cfg["n_speakers"], upsample_conditional_features=True,
                 upsample_params=dict(upsample_scales=cfg["upsample_scales"], cin_channels=cfg["Cc"], cin_pad=1),
                 use_speaker_embedding=True, cin_pad=1)
    wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items() if k.startswith("wavenet.")})
    wn = wn.cuda().train()
    feats = torch.from_numpy(z["feats"]).cuda().requires_grad_(True)
    y = wn(ins["xin"].cuda(), feats, ins["g"].cuda())
    assert rel_err(y[:, :, ::7].detach().cpu(), z["y_dec_probe"]) < 1e-3
    (y * O.hash_fill(tuple(y.shape), int(z["w_salt"]), 1.0).cuda()).sum().backward()
    assert rel_err(feats.grad.cpu(), z["dfeats"]) < 2e-3


@pytest.mark.parametrize("pad", [1, 0])
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
def test_plain_upsample_network_module(pad, dtype, tol):
    """The drop-in WaveNet with upsample_net="UpsampleNetwork" (upsample.py:29-66: the stages without conv_in, `indent` samples trimmed
    at either end; state_dict keys `upsample_net.up_layers.N`) on (B, Cc, Tc) features: logits, and in fp32 the gradient w.r.t. the
    features and every parameter's, against the reference's own (model_U.npz) / autograd through the oracle."""
    import json
    from helpers import load_npz
    from wavenet_autoencoders_amd.wavenet_vocoder import WaveNet
    z = load_npz("model_U")
    cfg = dict(json.loads(str(z["cfg"])), cin_pad=pad)
    sd = O.make_state_dict(cfg, int(z["salt"]), with_encoder=False)
    wn = WaveNet(out_channels=cfg["O"], layers=cfg["layers"], stacks=cfg["stacks"], residual_channels=cfg["R"],
                 gate_channels=cfg["G"], skip_out_channels=cfg["S"], kernel_size=cfg["k"], dropout=0.0, cin_channels=cfg["Cc"],
                 gin_channels=cfg["Cg"], n_speakers=cfg["n_speakers"], upsample_conditional_features=True, upsample_net="UpsampleNetwork",
                 upsample_params=dict(upsample_scales=cfg["upsample_scales"], cin_channels=cfg["Cc"], cin_pad=pad),
                 use_speaker_embedding=True, cin_pad=pad)
    assert set(wn.state_dict()) == {k[len("wavenet."):] for k in sd} and "upsample_net.up_layers.1.weight_v" in wn.state_dict()
    wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items()})
    wn = wn.cuda().train().set_compute_dtype(dtype)
    x = torch.from_numpy(z[f"x{pad}"]).long()
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    g = torch.from_numpy(z["g"])
    feats = torch.from_numpy(z[f"feats{pad}"]).cuda().requires_grad_(True)
    y = wn(xin.cuda(), feats, g.cuda())
    assert y.shape[-1] == x.shape[1]
    assert rel_err(y[:, :, ::5].detach().cpu(), z[f"y_probe{pad}"]) < tol
    if dtype != "fp32":
        return
    wsum = O.hash_fill(tuple(y.shape), int(z[f"w_salt{pad}"]), 1.0)
    (y * wsum.cuda()).sum().backward()
    assert rel_err(feats.grad.cpu(), z[f"dfeats{pad}"]) < 2e-3
    # every parameter gradient (incl. the four smoothing FIRs under their plain-network names) against autograd through the oracle
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=pad, conv_in=False)
    yo = O.wavenet_forward(psd, ocfg, xin, torch.from_numpy(z[f"feats{pad}"]), g)
    (yo * wsum).sum().backward()
    for k, p_ in wn.named_parameters():
        gref = psd["wavenet." + k].grad
        gref = gref if gref is not None else torch.zeros_like(psd["wavenet." + k])
        assert p_.grad is not None and rel_err(p_.grad.cpu(), gref) < 2e-3, k


@pytest.mark.parametrize("tag", ["leaky", "tanh", "relu", "sigm"])
def test_upsample_activation_module(tag):
    """The drop-in WaveNet with upsample_params upsample_activation = LeakyReLU(0.2) / ReLU / Sigmoid (ConvInUpsampleNetwork) and Tanh
    (plain UpsampleNetwork): state_dict keys as the reference's (three modules per stage), logits in fp32 and bf16, the gradient w.r.t.
    the features against the reference's own (model_V.npz), every parameter gradient against autograd through the oracle."""
    import json
    from helpers import load_npz
    from wavenet_autoencoders_amd.wavenet_vocoder import WaveNet
    z = load_npz("model_V")
    cfg = json.loads(str(z[f"cfg_{tag}"]))
    sd = O.make_state_dict(cfg, int(z["salt"]), with_encoder=False)
    params = {"negative_slope": cfg["up_act_slope"]} if cfg["up_act"] == "LeakyReLU" else {}
    x = torch.from_numpy(z["x"]).long()
    xin = torch.nn.functional.one_hot(x, cfg["O"]).float().transpose(1, 2).contiguous()
    g = torch.from_numpy(z["g"])
    for dtype, tol in (("bf16", 5e-2), ("fp32", 1e-3)):
        wn = WaveNet(out_channels=cfg["O"], layers=cfg["layers"], stacks=cfg["stacks"], residual_channels=cfg["R"],
                     gate_channels=cfg["G"], skip_out_channels=cfg["S"], kernel_size=cfg["k"], dropout=0.0, cin_channels=cfg["Cc"],
                     gin_channels=cfg["Cg"], n_speakers=cfg["n_speakers"], upsample_conditional_features=True,
                     upsample_net="ConvInUpsampleNetwork" if cfg["conv_in"] else "UpsampleNetwork",
                     upsample_params=dict(upsample_scales=cfg["upsample_scales"], cin_channels=cfg["Cc"], cin_pad=0,
                                          upsample_activation=cfg["up_act"], upsample_activation_params=params),
                     use_speaker_embedding=True, cin_pad=0)
        assert set(wn.state_dict()) == {k[len("wavenet."):] for k in sd}
        wn.load_state_dict({k[len("wavenet."):]: v for k, v in sd.items()})
        wn = wn.cuda().train().set_compute_dtype(dtype)
        feats = torch.from_numpy(z["feats"]).cuda().requires_grad_(True)
        y = wn(xin.cuda(), feats, g.cuda())
        assert rel_err(y[:, :, ::5].detach().cpu(), z[f"y_probe_{tag}"]) < tol
    wsum = O.hash_fill(tuple(y.shape), int(z["w_salt"]), 1.0)
    (y * wsum.cuda()).sum().backward()
    assert rel_err(feats.grad.cpu(), z[f"dfeats_{tag}"]) < 2e-3
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ocfg = dict(layers=cfg["layers"], stacks=cfg["stacks"], upsample_scales=cfg["upsample_scales"], cin_pad=0, conv_in=cfg["conv_in"],
                up_act=cfg["up_act"], up_act_slope=cfg["up_act_slope"])
    yo = O.wavenet_forward(psd, ocfg, xin, torch.from_numpy(z["feats"]), g)
    (yo * wsum).sum().backward()
    for k, p_ in wn.named_parameters():
        gref = psd["wavenet." + k].grad
        gref = gref if gref is not None else torch.zeros_like(psd["wavenet." + k])
        assert p_.grad is not None and rel_err(p_.grad.cpu(), gref) < 2e-3, k
