"""CPU-only checks of the host logic: C-ABI exports, parameter arena, fragment-order index maps (emulated MFMA
contraction in numpy against the oracle layer), config surface."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from oracle import wae_oracle as O
from wavenet_autoencoders_amd import packing as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(layers=4, stacks=2, R=32, G=48, S=32, O=64, Cc=16, Cg=8, k=3, n_speakers=5,
           upsample_scales=[4, 4, 8, 5], encoder_hid=32, c_in=39, K=32, cin_pad=0)


def test_library_exports_every_declared_symbol():
    from wavenet_autoencoders_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "wae.h")).read()
    declared = set(re.findall(r"\b(wae_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert declared <= set(_lib.SIGNATURES), declared - set(_lib.SIGNATURES)
    assert b"gfx950" in _lib.lib().wae_version()


def test_static_weight_gradient_launch_refuses_clips_its_descriptors_cannot_zero_fill():
    """Round-4 advisor finding: the C entry itself checks the 2^30-byte bound of the per-clip buffer descriptors (argument check only:
    it returns before any HIP call, so it runs without a GPU)."""
    from wavenet_autoencoders_amd import _lib
    lib = _lib.lib()
    dummy = ctypes.c_void_p(0x1000)
    for bad in (0, 1 << 30, 1 << 40):
        rc = lib.wae_gemm_tn_static(1, dummy, dummy, dummy, 1, 5, 5, 1, 64, None, None, 0, 0, 3, bad, None)
        assert rc == -1 and b"max_clip_bytes" in lib.wae_last_error()


def test_param_layout_matches_reference_state_dict():
    g = P.Geometry.from_cfg(CFG)
    lay = P.ParamLayout(g)
    sd = O.make_state_dict(CFG, 1)
    assert set(lay.offsets) == set(sd)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(lay.shapes[k]), k
    full = P.Geometry(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, n_speakers=153,
                      upsample_scales=[4, 4, 8, 5], c_in=39, encoder_hid=256, K=256)
    fl = P.ParamLayout(full)
    assert len(fl.offsets) == 302 and sum(fl.numel(k) for k in fl.offsets) == 7555218
    assert full.receptive_field == 4093
    assert len(fl.wn_cols) == sum(s[0] for n, s, v in P.param_specs(full) if v)
    # the arena ORDER is the reference model's named_parameters() order (recorded by make_golden.py from the reference's
    # own VQVAE): optimizer-state index i of a reference checkpoint is arena tensor i (checkpoint.py)
    import json
    from helpers import load_npz
    assert json.loads(str(load_npz("train_vqwae")["names"])) == list(fl.offsets)


def _arena(lay, sd):
    a = np.zeros(lay.total, dtype=np.float64)
    for k in lay.offsets:
        w = O.eff_weight(sd, k[:-9]) if k.endswith(".weight_v") else sd[k]
        a[lay.off(k):lay.off(k) + lay.numel(k)] = w.double().reshape(-1).numpy()
    return a


@pytest.mark.parametrize("dtype", [P.F32, P.BF16])
def test_fragment_maps_reproduce_the_layer(dtype):
    """Contract the packed streams exactly as the kernels do (A fragment element j of lane (i,h) meets B element j
    of lane (n,h)) and compare with the oracle's layer."""
    g = P.Geometry.from_cfg(CFG)
    lay = P.ParamLayout(g)
    sd = O.make_state_dict(CFG, 1)
    eff = _arena(lay, sd)
    t = P._traits(dtype)
    EPL, CK, KBU = t["EPL"], t["CK"], t["KBU"]
    li = 2
    base = li * lay.layer_stride

    def gather(m):
        return np.where(m >= 0, eff[np.maximum(m, 0) + base], 0.0)

    T, d = 40, 2
    x = O.hash_fill((1, g.R, T), 5, 1.0)
    c = O.hash_fill((1, g.Cc, T), 6, 1.0)
    p = f"wavenet.conv_layers.{li}."
    sdz = dict(sd)
    u_ref = O.glu_layer_gate(sdz, p, x, c, None, d)[0].double().numpy()           # (H, T)
    xo_ref, so_ref = O.glu_layer_forward(sdz, p, x, c, None, d)
    # operand vectors per time step in padded channel space
    xp = np.zeros((T, g.Rp)); xp[:, :g.R] = x[0].double().numpy().T
    cp = np.zeros((T, g.Ccp)); cp[:, :g.Cc] = c[0].double().numpy().T
    NM = 2 * g.NP
    w1 = gather(P.glu_w1_map(g, lay, dtype)).reshape(-1, 4, NM, 2, 32, EPL)        # [q][blk][m][h][i][j]
    cpr = g.Rp // CK
    z = np.zeros((NM * 32, T))
    for q in range(w1.shape[0]):
        for n in range(T):
            if q < g.k * cpr:
                cblk, tap = divmod(q, g.k)          # taps of one column block back to back (csrc/glu_fwd.hip: b_src)
                ts = n - (g.k - 1 - tap) * d
                row = xp[ts] if ts >= 0 else np.zeros(g.Rp)
            else:
                cblk = q - g.k * cpr
                row = cp[n]
            v = row[cblk * CK:(cblk + 1) * CK].reshape(4, 2, EPL)                  # [blk][h][j]
            z[:, n] += np.einsum("bmhij,bhj->mi", w1[q], v).reshape(-1)
    bias = sd[p + "conv.bias"].double().numpy()
    a = z[:g.H] + bias[:g.H, None]
    b = z[g.Hp:g.Hp + g.H] + bias[g.H:, None]
    u = np.tanh(a) / (1 + np.exp(-b))
    assert np.abs(u - u_ref).max() < 1e-5                                    # oracle is fp32
    assert np.abs(z[g.H:g.Hp]).max() == 0 and np.abs(z[g.Hp + g.H:]).max() == 0     # padded rows stay zero
    # second GEMM: k index of (kb, h, j) is the accumulator-tile row order
    NKB = g.NP * KBU
    w2 = gather(P.glu_w2_map(g, lay, dtype)).reshape(g.Rp // 32, NKB, 2, 32, EPL)   # [gm][kb][h][i][j]
    up = np.zeros((g.Hp, T)); up[:g.H] = u
    kb, h, j = np.meshgrid(np.arange(NKB), np.arange(2), np.arange(EPL), indexing="ij")
    ur = P.u_row_index(dtype, kb, h, j)
    assert sorted(ur.reshape(-1).tolist()) == list(range(g.Hp))                     # bijection onto the rows of u
    y = np.einsum("gkhij,khjn->gin", w2, up[ur]).reshape(g.Rp, T)
    bo = gather(P.glu_bias2_map(g, lay))
    xo = (y + bo[:, None] + xp.T) * np.sqrt(0.5)
    assert np.abs(xo[:g.R] - xo_ref[0].double().numpy()).max() < 1e-5
    assert np.abs(xo[g.R:]).max() == 0


@pytest.mark.parametrize("dtype", [P.F32, P.BF16])
def test_head_and_ar_maps(dtype):
    g = P.Geometry.from_cfg(CFG)
    lay = P.ParamLayout(g)
    sd = O.make_state_dict(CFG, 1)
    eff = _arena(lay, sd)
    t = P._traits(dtype)
    EPL, CK, KBU = t["EPL"], t["CK"], t["KBU"]
    hw = P.head_w_map(g, lay, dtype)
    assert hw.size == P.head_packed_elems(g, dtype)
    NT = g.Sp // 32
    n0 = (g.Ku // CK) * 4 * NT * 64 * EPL
    w0 = np.where(hw[:n0] >= 0, eff[np.maximum(hw[:n0], 0)], 0.0).reshape(-1, 4, NT, 2, 32, EPL)
    # skip contraction: sum_l W_skip_l u_l for random u
    u = np.random.default_rng(0).standard_normal((g.layers, g.H))
    uvec = np.zeros(g.Ku)
    for l in range(g.layers):
        uvec[l * g.Hp:l * g.Hp + g.H] = u[l]
    got = np.einsum("qbmhij,qbhj->mi", w0, uvec.reshape(-1, 4, 2, EPL)).reshape(-1)[:g.S]
    want = sum(O.eff_weight(sd, f"wavenet.conv_layers.{l}.conv1x1_skip")[:, :, 0].double().numpy() @ u[l] for l in range(g.layers))
    assert np.abs(got - want).max() < 1e-9
    # AR matrix-vector layout
    lm, w2_off = P.ar_layer_map(g, lay, dtype)
    K1 = g.k * g.R + g.Cc
    gp = (g.G + 63) // 64 * 64
    nkb = (K1 + EPL - 1) // EPL
    W1 = np.where(lm[:w2_off] >= 0, eff[np.maximum(lm[:w2_off], 0)], 0.0).reshape(nkb, gp, EPL)
    v = np.random.default_rng(1).standard_normal(nkb * EPL); v[K1:] = 0
    got = np.einsum("krj,kj->r", W1, v.reshape(nkb, EPL))[:g.G]
    conv = O.eff_weight(sd, "wavenet.conv_layers.0.conv").double().numpy()          # (G, R, k)
    cw = O.eff_weight(sd, "wavenet.conv_layers.0.conv1x1c").double().numpy()[:, :, 0]
    want = sum(conv[:, :, tap] @ v[tap * g.R:(tap + 1) * g.R] for tap in range(g.k)) + cw @ v[g.k * g.R:K1]
    assert np.abs(got - want).max() < 1e-9
    roff = P.ar_ring_offsets(g)
    assert roff[-1] == sum(((g.k - 1) * d + 1) * g.R for d in g.dilations)


def test_engine_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from wavenet_autoencoders_amd.engine import WaeEngine
    from wavenet_autoencoders_amd._lib import WaeError
    with pytest.raises(WaeError):
        WaeEngine(P.Geometry.from_cfg(CFG))


def test_feature_export_output_path_follows_reference_layout():
    """inference_2019.py:226-230: <dst>2019/<lan>/test/<utterance>.txt from a six-part base_dir."""
    import inference_2019 as I
    assert I.output_path("db/x/english/test/utt9/", "out/") == "out/2019/english/test/utt9.txt"
    with pytest.raises(AssertionError):
        I.output_path("/abs/db/x/english/test/utt9/", "out/")


def test_wavenet_constructor_refuses_options_it_does_not_implement():
    """wavenet.py:151 builds ConvInUpsampleNetwork(**upsample_params) (upsample.py:69-85).  Every option of that signature is
    either implemented or refused: a silently ignored key would give silently different numbers."""
    import pytest
    from wavenet_autoencoders_amd.wavenet_vocoder import WaveNet
    base = dict(out_channels=64, layers=4, stacks=2, residual_channels=32, gate_channels=48, skip_out_channels=32, cin_channels=16,
                gin_channels=8, n_speakers=5, upsample_conditional_features=True, use_speaker_embedding=True)
    up = dict(upsample_scales=[4, 4], cin_channels=16)
    w = WaveNet(**base, upsample_params=dict(up, cin_pad=1, upsample_activation="none", mode="nearest", freq_axis_kernel_size=1))
    assert w.geom.cin_pad == 1 and tuple(w.state_dict()["upsample_net.conv_in.weight"].shape) == (16, 16, 3)
    # no key in upsample_params: ConvInUpsampleNetwork's own default 0 (upsample.py:72) -- the reference never hands WaveNet's
    # cin_pad argument to the network (wavenet.py:151)
    w0 = WaveNet(**base, upsample_params=up, cin_pad=2)
    assert w0.geom.cin_pad == 0 and tuple(w0.state_dict()["upsample_net.conv_in.weight"].shape) == (16, 16, 1)
    # upsample_activation: ReLU / LeakyReLU / Tanh / Sigmoid are implemented since round 5 (three modules per stage: FIRs at 3 i + 1)
    wa = WaveNet(**base, upsample_params=dict(up, upsample_activation="LeakyReLU", upsample_activation_params={"negative_slope": 0.3}))
    assert wa.geom.up_act == "LeakyReLU" and wa.geom.up_act_slope == 0.3 and "upsample_net.upsample.up_layers.4.weight_v" in wa.state_dict()
    for bad, exc in ((dict(upsample_activation="PReLU"), NotImplementedError),
                     (dict(upsample_activation="ReLU", upsample_activation_params={"foo": 1}), NotImplementedError),
                     (dict(mode="linear"), NotImplementedError),
                     (dict(freq_axis_kernel_size=3), NotImplementedError), (dict(cin_channels=80), ValueError),
                     (dict(typo_scales=[4]), TypeError)):
        with pytest.raises(exc):
            WaveNet(**base, upsample_params=dict(up, **bad))
    with pytest.raises(TypeError):
        WaveNet(**base, upsample_params=dict(cin_channels=16))
    # the plain UpsampleNetwork (round 5): no conv_in, the stages under `upsample_net.up_layers.N`; any other name is what the reference's
    # getattr(upsample, upsample_net) raises for (wavenet.py:150)
    wu = WaveNet(**base, upsample_params=dict(up, cin_pad=1), upsample_net="UpsampleNetwork")
    assert not wu.geom.conv_in and wu.geom.cin_pad == 1 and "upsample_net.conv_in.weight" not in wu.state_dict()
    assert tuple(wu.state_dict()["upsample_net.up_layers.3.weight_v"].shape) == (1, 1, 1, 9)
    with pytest.raises(AttributeError):
        WaveNet(**base, upsample_params=up, upsample_net="FancyUpsampleNetwork")
    with pytest.raises(NotImplementedError):
        WaveNet(**base, upsample_params=up, kernel_size=5)     # 1..4 taps are implemented (tests/test_gpu_parity.py), wider ones refused
    from wavenet_autoencoders_amd.wavenet_vocoder.wavenet import receptive_field_size
    assert WaveNet(**base, upsample_params=up, kernel_size=2).receptive_field == receptive_field_size(base["layers"], base["stacks"], 2)


def test_synthesis_postprocessing_closed_form():
    """synthesis.py:382-394 after the decode: inv_mulaw_quantize(idx, 256), inv_preemphasis (audio.py:64-65: the IIR
    y[n] = x[n] + 0.85 y[n-1]), division by global_gain_scale -- against hand-derived values."""
    import synthesis as S
    mu = 256.0       # wavegen passes hparams.quantize_channels = 256 as nnmnkwii's `mu` (synthesis.py:384; SURVEY 8c notes 255 elsewhere)
    inv = lambda i: (lambda v: np.sign(v) * ((1.0 + mu) ** np.abs(v) - 1.0) / mu)(2.0 * i / mu - 1.0)   # noqa: E731  (companding formula)
    idx = np.array([255, 0, 128, 127, 200, 31])
    x = inv(idx.astype(np.float64))
    assert abs(x[1] + 1.0) < 1e-12 and x[2] == 0.0 and -2e-4 < x[3] < 0 and 0.9 < x[0] < 1.0     # class 128 is exact silence at mu = 256
    got = S.postprocess_indices(idx, 256, "inv_preemphasis", 0.55)
    want = np.empty(6)
    acc = 0.0
    for n in range(6):                        # the recurrence, term by term
        acc = x[n] + 0.85 * acc
        want[n] = acc / 0.55
    assert np.allclose(got, want, rtol=0, atol=1e-6)
    # impulse response of the de-emphasis filter: 0.85^n; no post-processing / no gain leave the companded values alone
    imp = S.inv_preemphasis(np.array([1.0, 0, 0, 0, 0]), 0.85)
    assert np.allclose(imp, 0.85 ** np.arange(5), atol=1e-12)
    assert np.allclose(S.postprocess_indices(idx, 256, "none", 0.0), x, atol=1e-6)
    assert np.allclose(S.postprocess_indices(idx, 256, None, 1.0), x, atol=1e-6)


def test_engine_options_are_read_once_and_checked(monkeypatch):
    """wavenet_autoencoders_amd/options.py: the launch-path switches -- defaults, every documented variable reaches its field, and a
    value outside a switch's vocabulary raises instead of silently selecting something."""
    import re
    from wavenet_autoencoders_amd.options import EngineOptions
    import wavenet_autoencoders_amd.options as opts
    for k in list(os.environ):
        if k.startswith("WAE_"):
            monkeypatch.delenv(k, raising=False)
    d = EngineOptions.from_env()
    assert d == EngineOptions() and d.bwd_fused == "auto" and d.bwd_fold_dc and d.dp_wire == "fp32" and d.glu_pair == "inference" and d.chains == "auto"
    flips = dict(WAE_TN_STREAM=("0", "tn_stream", False), WAE_TN_STATIC=("0", "tn_static", False), WAE_HEAD_SPLIT=("0", "head_split", False),
                 WAE_HEAD_WIDE=("1", "head_wide", True), WAE_GLU_PAIR=("1", "glu_pair", "1"), WAE_DP_SPLIT=("0", "dp_split", False),
                 WAE_DP_WIRE=("bf16", "dp_wire", "bf16"), WAE_AR_COOP=("0", "ar_coop", False), WAE_AR_COOP_C=("16", "ar_coop_c", 16),
                 WAE_BWD_FUSED=("0", "bwd_fused", "0"), WAE_BWD_FOLD_DC=("0", "bwd_fold_dc", False),
                 WAE_TN_STATIC_HEAD=("0", "tn_static_head", False), WAE_TN_SWAP=("0", "tn_swap", False), WAE_CHAINS=("2", "chains", "2"), WAE_SIDE=("0", "side", False))
    documented = set(re.findall(r"^    (WAE_[A-Z0-9_]+) ", opts.__doc__, re.M))
    assert documented == set(flips), documented ^ set(flips)
    for var, (val, field, want) in flips.items():
        monkeypatch.setenv(var, val)
        assert getattr(EngineOptions.from_env(), field) == want, var
        monkeypatch.delenv(var)
    for var, bad in (("WAE_BWD_FUSED", "yes"), ("WAE_DP_WIRE", "fp16"), ("WAE_GLU_PAIR", "2"), ("WAE_CHAINS", "3")):
        monkeypatch.setenv(var, bad)
        with pytest.raises(ValueError):
            EngineOptions.from_env()
        monkeypatch.delenv(var)



def test_negative_leaky_relu_slope_is_refused():
    """round-5 advisor finding: wae_act_bwd derives the derivative from the sign of the activation's OUTPUT, which a negative
    negative_slope flips -- Geometry refuses it (torch accepts it), for the module constructor and the training script alike."""
    import pytest
    from wavenet_autoencoders_amd import Geometry
    cfg = dict(layers=4, stacks=2, R=64, G=128, S=64, O=64, Cc=16, Cg=8, upsample_scales=[4], up_act="LeakyReLU")
    assert Geometry.from_cfg(dict(cfg, up_act_slope=0.2)).up_act_slope == 0.2
    assert Geometry.from_cfg(dict(cfg, up_act_slope=0.0)).up_act_slope == 0.0
    with pytest.raises(NotImplementedError, match="negative"):
        Geometry.from_cfg(dict(cfg, up_act_slope=-0.1))


def test_two_chains_rule():
    """options.two_chains (WaeEngine.chain_plan): which sweeps of layer launches run as two half-batch chains."""
    from wavenet_autoencoders_amd.options import two_chains
    c2, c3, c5 = (8, 8000), (8, 5120), (16, 5120)            # BASELINE configs[1], [2], [4]: 250 / 160 / 320 workgroups per layer launch
    assert two_chains("auto", True, *c2, backward=True) and not two_chains("auto", True, *c2, backward=False)
    assert not two_chains("auto", True, *c3, backward=True) and not two_chains("auto", True, *c3, backward=False)
    assert two_chains("auto", True, *c5, backward=True) and two_chains("auto", True, *c5, backward=False)
    assert not two_chains("auto", False, *c5, backward=True)                       # fp32 engines: one chain
    assert not two_chains("auto", True, 1, 64000, backward=True)                   # one clip cannot be cut
    assert not two_chains("1", True, *c5, backward=True) and two_chains("2", True, 2, 256, backward=False)
    assert not two_chains("2", True, 1, 256, backward=False) and not two_chains("2", False, 2, 256, backward=True)
