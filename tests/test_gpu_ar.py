"""GPU parity: autoregressive decoding (a12) against the golden vectors of the reference's incremental_forward."""
import numpy as np
import pytest
import torch

from helpers import golden_model, load_npz, rel_err
from oracle import wae_oracle as O

pytestmark = pytest.mark.gpu


def _engine(cfg, sd, dtype):
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    eng.load_state_dict(sd)
    return eng


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 5e-2)])
@pytest.mark.parametrize("name", ["A", "B"])
def test_teacher_forced_equals_reference(name, dtype, tol):
    cfg, sd, ins, zm, ocfg = golden_model(name)
    z = load_npz("ar_" + name)
    eng = _engine(cfg, sd, dtype)
    c_up = torch.from_numpy(z["c_up"]).cuda()
    Tar = c_up.shape[-1]
    out = eng.incremental_forward(c_up, ins["g"].cuda(), Tar, mode="logits", test_inputs=ins["x"][:, :Tar].cuda(),
                                  c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), z["tf_logits"]) < tol
    # known-answer property: incremental == batch forward of the same engine
    fwd = eng.decoder_forward(ins["x"][:, :Tar].cuda(), c_up, ins["g"].cuda(), c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), fwd["logits"].cpu()) < (1e-4 if dtype == "fp32" else 5e-2)


@pytest.mark.parametrize("name", ["A", "B"])
def test_greedy_rollout_bit_exact(name):
    cfg, sd, ins, zm, ocfg = golden_model(name)
    z = load_npz("ar_" + name)
    eng = _engine(cfg, sd, "fp32")
    c_up = torch.from_numpy(z["c_up"])[:, :, :24].contiguous().cuda()
    out = eng.incremental_forward(c_up, ins["g"].cuda(), 24, mode="argmax", init_idx=cfg["O"] // 2 - 1, c_is_upsampled=True)
    torch.cuda.synchronize()
    assert np.array_equal(out["idx"].cpu().numpy(), z["greedy"])


def test_sampling_matches_oracle_inverse_cdf():
    cfg, sd, ins, zm, ocfg = golden_model("A")
    z = load_npz("ar_A")
    eng = _engine(cfg, sd, "fp32")
    T = 40
    c_up = torch.from_numpy(z["c_up"])[:, :, :T].contiguous()
    u = O.hash_fill((2, T), 77) * 0.5 + 0.5
    init = torch.zeros(2, cfg["O"], 1)
    init[:, cfg["O"] // 2 - 1, 0] = 1
    ref = O.incremental_forward(sd, ocfg, c_up, ins["g"], T, initial_input=init, mode="sample", uniforms=u)
    out = eng.incremental_forward(c_up.cuda(), ins["g"].cuda(), T, mode="sample", uniforms=u.cuda(),
                                  init_idx=cfg["O"] // 2 - 1, c_is_upsampled=True)
    torch.cuda.synchronize()
    got = out["idx"].cpu().numpy()
    want = ref.argmax(1).numpy()
    # a draw can differ only where u lands within float rounding of a CDF step; require >= 95 % identical prefix-free
    first_diff = np.argmax(got != want, axis=1) if (got != want).any() else None
    assert (got == want).mean() > 0.95 or first_diff is None, (got, want)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 5e-2)])
def test_scalar_input_decode_against_reference_vectors(dtype, tol):
    """Scalar-input decoder (wavenet.py:284-285,325-333): mixture parameters of every step under teacher forcing and a
    free-running roll-out on explicit uniforms, against the vectors the reference's own incremental loop produced."""
    from helpers import golden_model, load_npz, rel_err
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, zm, ocfg = golden_model("S")
    z = load_npz("ar_S")
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    eng.load_state_dict(sd)
    c_up = torch.from_numpy(z["c_up"]).cuda()
    Tar = c_up.shape[-1]
    x = ins["x"][:, 0, :Tar].contiguous().cuda()
    g = ins["g"].cuda()
    out = eng.incremental_forward(c_up, g, Tar, mode="logits", test_inputs=x, c_is_upsampled=True)
    torch.cuda.synchronize()
    assert rel_err(out["logits"].cpu(), z["params_tf"]) < tol
    if dtype == "fp32":
        roll = eng.incremental_forward(c_up[:, :, :24].contiguous(), g, 24, mode="sample", c_is_upsampled=True,
                                       u_mix=torch.from_numpy(z["u_mix"])[:, :24].contiguous().cuda(),
                                       u_log=torch.from_numpy(z["u_log"])[:, :24].contiguous().cuda(), log_scale_min=-7.0)
        torch.cuda.synchronize()
        assert float((roll["x"].cpu() - torch.from_numpy(z["roll"])[:, 0]).abs().max()) < 1e-3


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-4), ("bf16", 5e-2)])
@pytest.mark.parametrize("name", ["A", "B"])
def test_dense_feedback_and_partial_teacher_forcing(name, dtype, tol):
    """quantize=False free-running (wavenet.py:303-305 with :335-338 skipped): the softmax probabilities, or the raw logits, of
    step t are the dense input of step t+1; and test_inputs shorter than T (forced, then free) -- against the rows the
    reference's own incremental_forward returned."""
    cfg, sd, ins, zm, ocfg = golden_model(name)
    z = load_npz("ar_" + name)
    eng = _engine(cfg, sd, dtype)
    Ts = z["soft"].shape[-1]
    c_up = torch.from_numpy(z["c_up"])[:, :, :Ts].contiguous().cuda()
    g = ins["g"].cuda()
    init = cfg["O"] // 2 - 1
    soft = eng.incremental_forward(c_up, g, Ts, mode="probs", init_idx=init, c_is_upsampled=True)["logits"]
    raw = eng.incremental_forward(c_up, g, Ts, mode="raw", init_idx=init, c_is_upsampled=True)["logits"]
    nf = int(z["part_forced"])
    part = eng.incremental_forward(c_up, g, Ts, mode="probs", test_inputs=ins["x"][:, :nf].cuda(), init_idx=init,
                                   c_is_upsampled=True)["logits"]
    torch.cuda.synchronize()
    assert rel_err(soft.cpu(), z["soft"]) < tol
    assert rel_err(raw.cpu(), z["raw"]) < tol
    assert rel_err(part.cpu(), z["part"]) < tol
    assert abs(float(soft[:, :, -1].sum()) - soft.shape[0]) < 1e-3        # rows are probabilities


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("bf16", 1e-5), ("fp16", 1e-5)])
@pytest.mark.parametrize("cc", [0, 64, 128, 256])
def test_constant_size_cooperative_kernel_every_packet_count(cc, dtype, tol, monkeypatch):
    """csrc/ar_coop.hip: ar_coop_fast_kernel<E, NU> is instantiated per number of gate-row packets a thread holds (3 / 4 for 16-bit
    elements, 6 / 7 / 8 for fp32 -- a function of the conditioning width).  Every instantiation, and its one-hand-over-per-layer variant
    (wae_ar_generate_coop_fused), against the one-CU kernel on the same teacher-forced inputs: the cooperative sums are exact fp32 in a
    fixed order, so only the summation order differs (1e-7); the fused variant multiplies by host-formed, rounded products."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=6, stacks=2, R=256, G=256, S=256, O=256, Cc=cc, Cg=32, k=3, n_speakers=7, upsample_scales=None, cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=3, with_encoder=False)
    T = 700
    gen = torch.Generator().manual_seed(1)
    x = torch.randint(0, 256, (2, T), generator=gen).cuda()
    c = torch.randn(2, cc, T, generator=gen).cuda() if cc > 0 else None
    gid = torch.tensor([1, 4]).cuda()

    def run(coop, fused):
        monkeypatch.setenv("WAE_AR_COOP", coop)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.ar_path(one_handover=fused == "1")
        eng.load_state_dict(sd)
        out = eng.incremental_forward(c, gid, T, mode="logits", test_inputs=x, c_is_upsampled=True)
        torch.cuda.synchronize()
        return out["logits"].float().cpu()

    one_cu = run("0", "0")
    assert rel_err(run("1", "0"), one_cu) < tol
    assert rel_err(run("1", "1"), one_cu) < (1e-5 if dtype == "fp32" else (2e-3 if dtype == "bf16" else 3e-4))


def test_decode_weights_follow_every_parameter_update():
    """The decode weights are a derivative of the parameters (make_generation_fast_, wavenet.py:358-364): a decode after a train step
    -- or after an averaged-weights swap and its restore, as vqwae_train.py's eval_model does -- must run on the weights of THAT
    moment.  (Round 3 repacked them only at the first decode; every later in-training evaluation used the first one's weights.)"""
    cfg, sd, ins, zm, ocfg = golden_model("A")
    z = load_npz("ar_A")
    eng = _engine(cfg, sd, "fp32")
    eng.init_optimizer()
    c_up = torch.from_numpy(z["c_up"]).cuda()
    Tar = c_up.shape[-1]
    x, g = ins["x"][:, :Tar].cuda(), ins["g"].cuda()

    def decode(e):
        out = e.incremental_forward(c_up, g, Tar, mode="logits", test_inputs=x, c_is_upsampled=True)
        torch.cuda.synchronize()
        return out["logits"].cpu()

    first = decode(eng)
    eng.train_step(ins["x"].cuda(), ins["c"].cuda(), g, lengths=None, lr=1e-2)
    second = decode(eng)
    fresh = _engine(cfg, eng.state_dict(), "fp32")
    assert rel_err(second, decode(fresh)) < 1e-6                 # the decode after the step = a fresh pack of the updated weights
    assert rel_err(second, first) > 1e-4                         # ... and the step did move them
    # the averaged-weights swap of eval_model and its restore
    saved = eng.params.clone()
    eng.params.copy_(eng.shadow)
    eng.weights_dirty = True
    ema = decode(eng)
    eng.params.copy_(saved)
    eng.weights_dirty = True
    assert rel_err(decode(eng), second) < 1e-6                   # restored: back to the trained weights, not the shadow's pack
    shadow_eng = _engine(cfg, {k: v for k, v in zip(eng.lay.offsets, [eng.shadow[eng.lay.off(k):eng.lay.off(k) + eng.lay.numel(k)].view(eng.lay.shapes[k]) for k in eng.lay.offsets])}, "fp32")
    assert rel_err(ema, decode(shadow_eng)) < 1e-6


def test_long_decode_40960_steps_against_the_oracle(monkeypatch):
    """The benchmarked autoregressive clip is 160 000 steps; parity elsewhere stops at 2 560.  Here: 40 960 teacher-forced steps of the
    synthesis geometry (C4: 20 layers, 256 channels, dilations to 512 -- every history ring wraps 40 times or more, the cooperative
    exchange's sequence numbers and banks 800 000 times) on the cooperative kernel in fp32 and bf16 and on the one-CU kernel, against
    the oracle on the same inputs at EVERY step.  The oracle side is its batch forward: teacher-forced incremental decoding and the
    batch forward are the same function (wavenet.py:218-346 vs :164-216; pinned step by step on the reference's own roll-outs in
    tests/test_gpu_configs.py and, for the first 1 024 steps of this very input, against the oracle's incremental loop below).  A
    free-running sampled decode of the same length is bitwise reproducible."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5], cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=11, with_encoder=False)
    ocfg = dict(layers=20, stacks=2, upsample_scales=cfg["upsample_scales"], cin_pad=0)
    T = 64 * 640
    x = ((O.hash_fill((1, T), 901) * 0.5 + 0.5) * 256).long().clamp(0, 255)         # hash-filled class ids
    lat = O.hash_fill((1, 64, 64), 902)
    gid = torch.tensor([5])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    with torch.no_grad():
        # teacher forcing feeds test_inputs[:, t] at step t (wavenet.py:300-301): the batch forward on the same sequence
        xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
        want = O.wavenet_forward(sd, ocfg, xin, lat, gid)[0]                        # (O, T)
        c_up = O.upsample_forward(sd, lat, cfg["upsample_scales"])[:, :, :1024].contiguous()
        init = torch.zeros(1, 256, 1)
        init[:, 127, 0] = 1
        inc = O.incremental_forward(sd, ocfg, c_up, gid, 1024, initial_input=init, mode="logits",
                                    test_inputs=torch.nn.functional.one_hot(x[:, :1024], 256).float().transpose(1, 2).contiguous())
        inc = inc["logits"] if isinstance(inc, dict) else inc
    assert rel_err(torch.as_tensor(inc)[0][:, :1024], want[:, :1024]) < 1e-4         # the oracle's loop == the oracle's batch forward

    def decode(dtype, coop, mode="logits", **kw):
        monkeypatch.setenv("WAE_AR_COOP", coop)
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        out = eng.incremental_forward(lat.cuda(), gid.cuda(), T, mode=mode, init_idx=127, **kw)
        torch.cuda.synchronize()
        return out

    scale = float(want.abs().max())
    lse_w = torch.logsumexp(want, 0)
    coop32 = decode("fp32", "1", test_inputs=x.cuda())["logits"].cpu()[0]
    assert float((coop32 - want).abs().max()) < 1e-3 * scale                         # every logit of every one of the 40 960 steps
    assert float((torch.logsumexp(coop32, 0) - lse_w).abs().max()) < 1e-3 * scale
    # the last 4 096 steps on their own: an error that grows with the step count would show here first
    assert float((coop32[:, -4096:] - want[:, -4096:]).abs().max()) < 1e-3 * scale
    one_cu = decode("fp32", "0", test_inputs=x.cuda())["logits"].cpu()[0]
    assert float((one_cu - want).abs().max()) < 1e-3 * scale
    assert float((one_cu - coop32).abs().max()) < 1e-5 * scale                       # same fp32 arithmetic, another summation order
    coop16 = decode("bf16", "1", test_inputs=x.cuda())["logits"].float().cpu()[0]
    assert float((coop16 - want).abs().max()) < 5e-2 * scale
    assert float((coop16[:, -4096:] - want[:, -4096:]).abs().max()) < 5e-2 * scale
    # free-running, sampled, 40 960 steps, twice: the same bits
    uni = torch.rand(1, T, generator=torch.Generator().manual_seed(17)).cuda()
    a = decode("bf16", "1", mode="sample", uniforms=uni)["idx"].cpu()
    b = decode("bf16", "1", mode="sample", uniforms=uni)["idx"].cpu()
    assert torch.equal(a, b) and int(torch.unique(a).numel()) > 200


def test_full_clip_160000_steps_windows_against_the_oracle():
    """The benchmarked clip in full: 160 000 teacher-forced steps (250 latent frames x 640) of the C4 geometry on the default kernels
    (fp32; bf16 = every layer's weights resident on chip), logits compared with the oracle inside three windows of 4 096 steps at
    60 000, 120 000 and the clip's end.  A logit depends on the 4 093 inputs before it only (receptive field of 2 stacks of dilations
    1 .. 512 at 3 taps), so the oracle's batch forward on [a - 4093, a + 4096) is exact for [a, a + 4096): the windows cost seconds
    where the whole clip would cost minutes, and an error that grows with the step count -- ring wrap-around, exchange sequence
    numbers (3.2 million hand-overs per member), the one-hot feedback path -- would show in the late ones."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5], cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=13, with_encoder=False)
    T, RF, N = 250 * 640, 4093, 4096
    x = ((O.hash_fill((1, T), 911) * 0.5 + 0.5) * 256).long().clamp(0, 255)
    lat = O.hash_fill((1, 64, 250), 912)
    gid = torch.tensor([9])
    torch.set_num_threads(min(16, torch.get_num_threads()))
    wcfg = dict(layers=20, stacks=2, upsample_scales=None, cin_pad=0)
    want = {}
    with torch.no_grad():
        c_up = O.upsample_forward(sd, lat, cfg["upsample_scales"])                   # (1, 64, T)
        for a in (60000, 120000, T - N):
            xin = torch.nn.functional.one_hot(x[:, a - RF:a + N], 256).float().transpose(1, 2).contiguous()
            want[a] = O.wavenet_forward(sd, wcfg, xin, c_up[:, :, a - RF:a + N].contiguous(), gid)[0][:, RF:]
    for dtype, tol in (("fp32", 1e-3), ("bf16", 5e-2)):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        got = eng.incremental_forward(lat.cuda(), gid.cuda(), T, mode="logits", init_idx=127, test_inputs=x.cuda())["logits"]
        torch.cuda.synchronize()
        for a, w in want.items():
            g_ = got[0][:, a:a + N].float().cpu()
            scale = float(w.abs().max())
            assert float((g_ - w).abs().max()) < tol * scale, (dtype, a)
            assert float((torch.logsumexp(g_, 0) - torch.logsumexp(w, 0)).abs().max()) < tol * scale, (dtype, a)
        del got, eng
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_resident_weights_and_shared_ring_do_not_change_a_bit(dtype, monkeypatch):
    """Round 5: the 16-bit fast cooperative kernel keeps the packets of 6 layers in LDS, of 11 more in the accumulation registers and
    of the last 3 in hand-allocated arch VGPRs (ar_coop_fast_vb_kernel: bank slots 11-13), and its 32 members share one history ring
    (csrc/ar_coop.hip: LDSW).  Where a layer's packets live is not arithmetic: a sampled
    decode at the C4 geometry (20 layers, dilations to 512, T = 1500: the rings of the wide layers wrap) must be BITWISE the decode of the
    streaming form (ar_path(lds_layers=0): no resident layer, private rings), for every split of the layers between LDS, registers and
    memory; teacher-forced logits likewise.  (Round 6: the split is a field of wae_ar_desc; rounds 4-5 read it from the environment
    inside the library.)"""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5], cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=7, with_encoder=False)
    T = 1280 + 640
    gen = torch.Generator().manual_seed(5)
    lat = torch.randn(2, 64, T // 640, generator=gen).cuda()
    gid = torch.tensor([3, 77]).cuda()
    uni = torch.rand(2, T, generator=gen).cuda()
    forced = torch.randint(0, 256, (2, T), generator=gen).cuda()
    got = {}
    for tag, split in (("stream", dict(lds_layers=0)), ("default", {}), ("lds3", dict(lds_layers=3, reg_layers=0)),
                       ("lds2+bank5", dict(lds_layers=2, reg_layers=5)),
                       ("lds6+bank11", dict(reg_layers=11)),            # the instantiation without the arch-VGPR bank
                       ("lds1+bank13", dict(lds_layers=1, reg_layers=13))):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.ar_path(**split)
        eng.load_state_dict(sd)
        a = eng.incremental_forward(lat, gid, T, mode="sample", uniforms=uni)["idx"].clone()
        b_ = eng.incremental_forward(lat, gid, T, mode="logits", test_inputs=forced, want_logits=True)["logits"].clone()
        torch.cuda.synchronize()
        got[tag] = (a, b_)
    assert len(torch.unique(got["stream"][0])) > 50
    for tag in ("default", "lds3", "lds2+bank5", "lds6+bank11", "lds1+bank13"):
        assert torch.equal(got["stream"][0], got[tag][0]), tag
        assert torch.equal(got["stream"][1], got[tag][1]), tag


@pytest.mark.parametrize("layers,stacks", [(30, 3), (24, 2), (6, 2), (2, 1)])
def test_resident_weights_at_other_depths(layers, stacks, monkeypatch):
    """The residency of the fast cooperative kernel is sized for the reference's 20 layers (6 in LDS + 11 + 3 in registers); deeper stacks
    stream the layers that do not fit, shallower ones leave bank slots empty: a sampled decode stays bitwise the streaming form's."""
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg = dict(layers=layers, stacks=stacks, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5],
               cin_pad=0)
    sd = O.make_state_dict(dict(cfg), salt=7, with_encoder=False)
    T = 1280
    gen = torch.Generator().manual_seed(5)
    lat = torch.randn(2, 64, T // 640, generator=gen).cuda()
    gid = torch.tensor([3, 77]).cuda()
    uni = torch.rand(2, T, generator=gen).cuda()
    got = {}
    for tag, split in (("stream", dict(lds_layers=0)), ("default", {})):
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype="bf16")
        eng.ar_path(**split)
        eng.load_state_dict(sd)
        got[tag] = eng.incremental_forward(lat, gid, T, mode="sample", uniforms=uni)["idx"].clone()
        torch.cuda.synchronize()
    assert torch.equal(got["stream"], got["default"]) and int(torch.unique(got["default"]).numel()) > 50



@pytest.mark.parametrize("kernel", ["coop", "one_cu"])
def test_start_class_per_utterance_against_oracle(kernel, monkeypatch):
    """wavenet.py:283-297: every batch item starts from ITS row of initial_input.  Two utterances with different start classes,
    greedy roll-out (fp32: bit-exact against the oracle, which restates the reference loop with the same initial_input), on the
    cooperative and on the one-CU kernel; a single start class still broadcasts (an int, or one row)."""
    monkeypatch.setenv("WAE_AR_COOP", "1" if kernel == "coop" else "0")
    cfg, sd, ins, zm, ocfg = golden_model("A")
    z = load_npz("ar_A")
    eng = _engine(cfg, sd, "fp32")
    T = 24
    c_up = torch.from_numpy(z["c_up"])[:, :, :T].contiguous()
    starts = [3, cfg["O"] - 2]
    init = torch.zeros(2, cfg["O"], 1)
    for b, s in enumerate(starts):
        init[b, s, 0] = 1
    ref = O.incremental_forward(sd, ocfg, c_up, ins["g"], T, initial_input=init, mode="argmax").argmax(1).numpy()
    out = eng.incremental_forward(c_up.cuda(), ins["g"].cuda(), T, mode="argmax", init_idx=torch.tensor(starts), c_is_upsampled=True)
    torch.cuda.synchronize()
    got = out["idx"].cpu().numpy()
    assert np.array_equal(got, ref)
    # the two utterances really started differently: the same start class for both gives another roll-out for at least one of them
    same = eng.incremental_forward(c_up.cuda(), ins["g"].cuda(), T, mode="argmax", init_idx=starts[0], c_is_upsampled=True)["idx"].cpu().numpy()
    assert np.array_equal(same[0], got[0])
    with pytest.raises(IndexError):
        eng.incremental_forward(c_up.cuda(), ins["g"].cuda(), T, mode="argmax", init_idx=torch.tensor([0, cfg["O"]]), c_is_upsampled=True)
    with pytest.raises(ValueError):
        eng.incremental_forward(c_up.cuda(), ins["g"].cuda(), T, mode="argmax", init_idx=torch.tensor([0, 1, 2]), c_is_upsampled=True)
