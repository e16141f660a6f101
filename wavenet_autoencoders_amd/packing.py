"""Host logic: parameter arena layout and the index maps that put weights into MFMA A-fragment order.

Everything here is numpy on the host and runs once per model geometry; the maps are uploaded and applied on
the device by ``wae_pack_gather`` (csrc/misc.hip) every time the weights change.

Fragment order (must match csrc/glu_fwd.hip and csrc/head_fwd.hip)
------------------------------------------------------------------
A *fragment block* is 64 lanes x 16 bytes = one A operand of a 32x32 MFMA tile: lane l = (i = l & 31, h = l >> 5)
holds, for output row ``32*m + i`` of M-tile m, EPL consecutive-in-k elements (EPL = 8 bf16 / 4 f32).

GEMM 1 stream ``[pass][q][blk][m][lane][j]`` (chunk q = one 128-byte slice of an activation row; pass p covers the
gate channels [p*NPH*32, (p+1)*NPH*32), NPH = glu_pass_tiles(NP); m < NPH: tanh rows, m >= NPH: sigmoid rows):
    q <  k*(Rp/CK): cblk = q // k, tap = q % k               -> conv weight (G, R, k)  (taps of a column block back to back)
    q >=          : c chunk                                   -> conv1x1c weight (G, Cc)
    channel = cblk*CK + blk*2*EPL + h*EPL + j          (CK = 64 bf16 / 32 f32 channels per chunk)
    gate-a row = 32*(p*NPH + m) + i, gate-b row = H + 32*(p*NPH + m - NPH) + i (reference row = half*H + i)
GEMM 2 stream ``[q2][mt][kb][lane][j]``: M-tile gm = q2*MT2 + mt over [out rows (Rp) | skip rows (Sp)],
    k index = row of u inside accumulator tile ut = kb // KBU, sub-block s = kb % KBU:
      bf16: u_row = 32*ut + 16*s + 8*(j >> 2) + 4*h + (j & 3)
      f32 : u_row = 32*ut +  8*s + 4*h + j
    (the register order in which a 32x32 accumulator tile is reused as the next MFMA's B operand).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

F32, BF16, F16 = 0, 1, 2    # include/wae.h WAE_F32 / WAE_BF16 / WAE_F16 (fp16 packs exactly like bf16)


def _ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class Geometry:
    """Decoder/encoder sizes (reference ctor args: wavenet.py:98-111, vqvae_model.py:54) + padded sizes."""
    layers: int
    stacks: int
    R: int
    G: int
    S: int
    O: int
    Cc: int = -1
    Cg: int = -1
    k: int = 3
    n_speakers: Optional[int] = None
    upsample_scales: Optional[List[int]] = None
    cin_pad: int = 0
    scalar_input: bool = False
    use_speaker_embedding: bool = True
    # autoencoder front end (None = decoder only)
    c_in: Optional[int] = None
    encoder_hid: Optional[int] = None
    K: int = 256
    dilations_override: Optional[List[int]] = None   # standalone layers (wavenet_vocoder.modules.ResidualConv1dGLU)
    # upsample_net "ConvInUpsampleNetwork" (upsample.py:69-85: a Conv1d of 2 cin_pad + 1 taps without padding in front of the stages;
    # every preset) or, conv_in = False, the plain "UpsampleNetwork" (upsample.py:29-66: the stages alone, the output trimmed by
    # cin_pad * prod(scales) samples at either end) -- the same output length, other state_dict keys
    conv_in: bool = True
    # upsample_activation (upsample.py:44-46): "none" (every preset) or an element-wise torch.nn module behind every stage's FIR --
    # ReLU, LeakyReLU (up_act_slope = negative_slope), Tanh, Sigmoid are implemented (csrc/misc.hip: wae_act_fwd / _bwd).  With an
    # activation the ModuleList holds three modules per stage, so the FIRs sit at up_layers.{3 i + 1}.
    up_act: str = "none"
    up_act_slope: float = 0.01
    # Global features that VARY over time (modules.py:148-152 convolves any (B, Cg, T) tensor): conv1x1g then is one more 1x1 over a
    # time series, exactly what conv1x1c is -- its Cg columns ride behind the Cc columns of the local conditioning in the layer
    # kernel's operand ([c ; g] in c_up, [Wc | Wg] in the packed GEMM-1 stream), forward and backward; the hoisted per-clip projection
    # (gproj) is not used.  Set by the stand-alone layer module when it is handed such a tensor.
    g_local: bool = False

    def __post_init__(self):
        assert self.layers % self.stacks == 0                       # wavenet.py:117
        assert self.G % 2 == 0
        self.H = self.G // 2
        self.Rp = _ru(self.R, 128)
        self.Sp = _ru(self.S, 128)
        self.Op = _ru(self.O, 128)
        self.Ccp = _ru(self.Cc, 64) if self.Cc > 0 else 0
        self.Cx = max(self.Cc, 0)                     # conditioning columns of the layer kernel's operand
        if self.g_local:
            assert self.Cg > 0
            self.Cx += self.Cg
            self.Ccp = _ru(self.Cx, 64)
        self.Hp = _ru(self.H, 32)
        self.NP = self.Hp // 32
        self.Ku = _ru(self.layers * self.Hp, 64)       # columns of the (B,T,Ku) buffer of all layers' gated activations
        per = self.layers // self.stacks
        self.dilations = [2 ** (i % per) for i in range(self.layers)]   # wavenet.py:126
        if self.dilations_override is not None:
            assert len(self.dilations_override) == self.layers
            self.dilations = [int(d) for d in self.dilations_override]
        self.receptive_field = (self.k - 1) * sum(self.dilations) + 1   # wavenet.py:42-60
        self.has_encoder = self.c_in is not None
        if self.up_act == "LeakyReLU" and not self.up_act_slope >= 0.0:
            # wae_act_bwd forms act'(x) from the sign of the stored OUTPUT (csrc/misc.hip: act_bwd_kernel); with a negative slope the
            # output's sign no longer is the input's (torch accepts such slopes): refused, not differentiated wrongly
            raise NotImplementedError(f"upsample_activation LeakyReLU(negative_slope={self.up_act_slope}): negative slopes are not implemented")

    @staticmethod
    def from_cfg(cfg: dict) -> "Geometry":
        return Geometry(layers=cfg["layers"], stacks=cfg["stacks"], R=cfg["R"], G=cfg["G"], S=cfg["S"], O=cfg["O"],
                        Cc=cfg.get("Cc", -1), Cg=cfg.get("Cg", -1), k=cfg.get("k", 3), n_speakers=cfg.get("n_speakers"),
                        upsample_scales=cfg.get("upsample_scales"), cin_pad=cfg.get("cin_pad", 0),
                        scalar_input=bool(cfg.get("scalar_input")), c_in=cfg.get("c_in"),
                        encoder_hid=cfg.get("encoder_hid"), K=cfg.get("K", 256), conv_in=bool(cfg.get("conv_in", True)),
                        up_act=cfg.get("up_act", "none"), up_act_slope=float(cfg.get("up_act_slope", 0.01)))


ENCODER_BLOCKS = [(3, 1), (3, 1), (5, 2), (5, 2), (3, 1), (3, 1), (1, 1), (1, 1), (1, 1), (1, 1)]  # vqvae_model.py:32-40


UP_ACT_KINDS = {"ReLU": 1, "LeakyReLU": 2, "Tanh": 3, "Sigmoid": 4}      # include/wae.h: wae_act_fwd


def up_stage_name(g: Geometry, i: int) -> str:
    """state_dict prefix of upsampling stage i's smoothing FIR: the ModuleList holds [Stretch2d, Conv2d] per stage (upsample.py:38-44),
    under `.upsample` when ConvInUpsampleNetwork wraps it (upsample.py:80-82)"""
    per = 2 if g.up_act == "none" else 3
    return f"wavenet.upsample_net.{'upsample.' if g.conv_in else ''}up_layers.{per * i + 1}"


def param_specs(g: Geometry) -> List[Tuple[str, Tuple[int, ...], bool]]:
    """(name, shape, is_weight_norm_v) in the reference's registration order (state_dict keys)."""
    out: List[Tuple[str, Tuple[int, ...], bool]] = []

    def wn(prefix, cout, cin, k, bias=True, conv2d=False):
        if bias:
            out.append((prefix + ".bias", (cout,), False))
        gs = (cout, 1, 1, 1) if conv2d else (cout, 1, 1)
        vs = (cout, cin, 1, k) if conv2d else (cout, cin, k)
        out.append((prefix + ".weight_g", gs, False))
        out.append((prefix + ".weight_v", vs, True))

    wn("wavenet.first_conv", g.R, 1 if g.scalar_input else g.O, 1)
    for i in range(g.layers):
        p = f"wavenet.conv_layers.{i}."
        wn(p + "conv", g.G, g.R, g.k)
        if g.Cc > 0:
            wn(p + "conv1x1c", g.G, g.Cc, 1, bias=False)
        if g.Cg > 0:
            wn(p + "conv1x1g", g.G, g.Cg, 1, bias=False)
        wn(p + "conv1x1_out", g.R, g.H, 1)
        wn(p + "conv1x1_skip", g.S, g.H, 1)
    wn("wavenet.last_conv_layers.1", g.S, g.S, 1)
    wn("wavenet.last_conv_layers.3", g.O, g.S, 1)
    if g.Cg > 0 and g.use_speaker_embedding and g.n_speakers:
        out.append(("wavenet.embed_speakers.weight", (g.n_speakers, g.Cg), False))
    if g.upsample_scales:
        if g.conv_in:
            out.append(("wavenet.upsample_net.conv_in.weight", (g.Cc, g.Cc, 2 * g.cin_pad + 1), False))
        for i, s in enumerate(g.upsample_scales):
            wn(up_stage_name(g, i), 1, 1, 2 * s + 1, bias=False, conv2d=True)
    if g.has_encoder:
        dims = [(g.c_in, g.encoder_hid)] + [(g.encoder_hid, g.encoder_hid)] * 9
        for i, ((ci, co), (kk, _)) in enumerate(zip(dims, ENCODER_BLOCKS)):
            out.append((f"encoder.net.{i}.conv.weight", (co, ci, kk), False))
            out.append((f"encoder.net.{i}.conv.bias", (co,), False))
        out.append(("encoder.lin.weight", (g.Cc, g.encoder_hid), False))
        out.append(("encoder.lin.bias", (g.Cc,), False))
        out.append(("vq.embedding.weight", (g.K, g.Cc), False))
    return out


class ParamLayout:
    """Flat fp32 arena: name -> (offset, shape).  One contiguous buffer for parameters, one for gradients,
    one for the effective (weight-normed) weights: a single all-reduce / optimizer launch covers everything."""

    def __init__(self, g: Geometry):
        self.geom = g
        self.offsets: "OrderedDict[str, int]" = OrderedDict()
        self.shapes: Dict[str, Tuple[int, ...]] = {}
        off = 0
        v_off, g_off, cols = [], [], []
        for name, shape, is_v in param_specs(g):
            n = int(np.prod(shape))
            self.offsets[name] = off
            self.shapes[name] = shape
            off += _ru(n, 4)            # keep every tensor 16-byte aligned
        self.total = off
        for name, shape, is_v in param_specs(g):
            if is_v:
                rows = shape[0]
                c = int(np.prod(shape[1:]))
                gname = name[:-1] + "g"
                for r in range(rows):
                    v_off.append(self.offsets[name] + r * c)
                    g_off.append(self.offsets[gname] + r)
                    cols.append(c)
        self.wn_v_off = np.asarray(v_off, dtype=np.int64)
        self.wn_g_off = np.asarray(g_off, dtype=np.int64)
        self.wn_cols = np.asarray(cols, dtype=np.int32)
        if g.layers > 1:
            self.layer_stride = self.offsets["wavenet.conv_layers.1.conv.bias"] - self.offsets["wavenet.conv_layers.0.conv.bias"]
        else:
            self.layer_stride = 0

    def off(self, name: str) -> int:
        return self.offsets[name]

    def numel(self, name: str) -> int:
        return int(np.prod(self.shapes[name]))


def _traits(dtype: int):
    if dtype in (BF16, F16):
        return dict(EPL=8, CK=64, KBU=2, MT2=4, ES=2)
    return dict(EPL=4, CK=32, KBU=4, MT2=2, ES=4)


def u_row_index(dtype: int, kb: np.ndarray, h: np.ndarray, j: np.ndarray) -> np.ndarray:
    """k index (row of the previous accumulator tile stack) held by element j of lane-half h in k-block kb."""
    t = _traits(dtype)
    ut, s = kb // t["KBU"], kb % t["KBU"]
    if dtype in (BF16, F16):
        return 32 * ut + 16 * s + 8 * (j >> 2) + 4 * h + (j & 3)
    return 32 * ut + 8 * s + 4 * h + j


def glu_pass_tiles(NP: int) -> int:
    """Gate-channel tiles per GEMM-1 pass (csrc/glu_fwd.hip dispatch_np must agree)."""
    if NP not in (1, 2, 3, 4, 6, 8):
        raise ValueError(f"gate_channels/2 padded to {32 * NP} is not supported by the fused layer kernel (Hp/32 must be 1,2,3,4,6,8)")
    return {1: 1, 2: 1, 3: 3, 4: 4, 6: 3, 8: 4}[NP]


def cond_weight_src(g: Geometry, lay: ParamLayout, row, cc):
    """arena offset (layer 0) of the 1x1 conditioning weight [gate row `row`][operand column `cc`], and its validity: columns
    [0, Cc) = conv1x1c; with g_local, [Cc, Cc + Cg) = conv1x1g (Geometry.g_local)."""
    Cc = max(g.Cc, 0)
    src = np.zeros(np.broadcast(row, cc).shape, dtype=np.int64)
    ok = np.zeros(src.shape, dtype=bool)
    if Cc > 0:
        c_off = lay.off("wavenet.conv_layers.0.conv1x1c.weight_v")
        isc = (cc >= 0) & (cc < Cc)
        src = np.where(isc, c_off + row * Cc + cc, src)
        ok |= isc
    if g.g_local:
        g_off = lay.off("wavenet.conv_layers.0.conv1x1g.weight_v")
        isg = (cc >= Cc) & (cc < Cc + g.Cg)
        src = np.where(isg, g_off + row * g.Cg + (cc - Cc), src)
        ok |= isg
    return src, ok


def glu_w1_map(g: Geometry, lay: ParamLayout, dtype: int) -> np.ndarray:
    """int32 map (relative to arena start, layer 0) for the GEMM-1 stream of one GLU layer."""
    t = _traits(dtype)
    EPL, CK = t["EPL"], t["CK"]
    NPH = glu_pass_tiles(g.NP)
    NM = 2 * NPH
    cpr = g.Rp // CK
    nq_conv = g.k * cpr
    nq1 = nq_conv + g.Ccp // CK
    ps, q, blk, m, lane, j = np.meshgrid(np.arange(g.NP // NPH), np.arange(nq1), np.arange(4), np.arange(NM), np.arange(64),
                                         np.arange(EPL), indexing="ij")
    i, h = lane & 31, lane >> 5
    half = (m >= NPH).astype(np.int64)
    ig = 32 * (ps * NPH + m - half * NPH) + i
    row = half * g.H + ig
    is_conv = q < nq_conv
    tap = np.where(is_conv, q % g.k, 0)               # taps of one column block back to back (csrc/glu_fwd.hip: b_src)
    cblk = np.where(is_conv, q // g.k, q - nq_conv)
    ch = cblk * CK + blk * 2 * EPL + h * EPL + j
    conv_off = lay.off("wavenet.conv_layers.0.conv.weight_v")
    src_conv = conv_off + (row * g.R + ch) * g.k + tap
    valid_conv = (ig < g.H) & (ch < g.R)
    src_c, valid_c = cond_weight_src(g, lay, row, ch)
    valid_c = valid_c & (ig < g.H)
    out = np.where(is_conv, np.where(valid_conv, src_conv, -1), np.where(valid_c, src_c, -1))
    return out.astype(np.int32).reshape(-1)


def glu_w2_map(g: Geometry, lay: ParamLayout, dtype: int) -> np.ndarray:
    """GEMM-2 stream of one layer: conv1x1_out rows only (the skip 1x1 is contracted in the head)."""
    t = _traits(dtype)
    EPL, KBU = t["EPL"], t["KBU"]
    NKB = g.NP * KBU
    n_mt = g.Rp // 32
    gm, kb, lane, j = np.meshgrid(np.arange(n_mt), np.arange(NKB), np.arange(64), np.arange(EPL), indexing="ij")
    i, h = lane & 31, lane >> 5
    ur = u_row_index(dtype, kb, h, j)
    r_out = 32 * gm + i
    src_out = lay.off("wavenet.conv_layers.0.conv1x1_out.weight_v") + r_out * g.H + ur
    out = np.where((r_out < g.R) & (ur < g.H), src_out, -1)
    return out.astype(np.int32).reshape(-1)


def glu_bias2_map(g: Geometry, lay: ParamLayout) -> np.ndarray:
    r = np.arange(g.Rp)
    o = lay.off("wavenet.conv_layers.0.conv1x1_out.bias")
    return np.where(r < g.R, o + r, -1).astype(np.int32)


def first_conv_maps(g: Geometry, lay: ParamLayout):
    """table (O or 1, Rp) with table[o][r] = W[r][o]; bias (Rp)."""
    nin = 1 if g.scalar_input else g.O
    o, r = np.meshgrid(np.arange(nin), np.arange(g.Rp), indexing="ij")
    tab = np.where(r < g.R, lay.off("wavenet.first_conv.weight_v") + r * nin + o, -1).astype(np.int32).reshape(-1)
    rr = np.arange(g.Rp)
    bias = np.where(rr < g.R, lay.off("wavenet.first_conv.bias") + rr, -1).astype(np.int32)
    return tab, bias


def head_w_map(g: Geometry, lay: ParamLayout, dtype: int) -> np.ndarray:
    """[GEMM 0: all layers' conv1x1_skip, GEMM-1 order over K = (layer, u channel)] +
       [GEMM 1: last_conv_layers.1, GEMM-2 order] + [GEMM 2: last_conv_layers.3, GEMM-2 order]."""
    t = _traits(dtype)
    EPL, CK, KBU = t["EPL"], t["CK"], t["KBU"]
    NT = g.Sp // 32
    nq0 = g.Ku // CK
    q, blk, m, lane, j = np.meshgrid(np.arange(nq0), np.arange(4), np.arange(NT), np.arange(64), np.arange(EPL),
                                     indexing="ij")
    i, h = lane & 31, lane >> 5
    row = 32 * m + i
    kk = q * CK + blk * 2 * EPL + h * EPL + j
    layer, ch = kk // g.Hp, kk % g.Hp
    w0 = np.where((row < g.S) & (layer < g.layers) & (ch < g.H),
                  lay.off("wavenet.conv_layers.0.conv1x1_skip.weight_v") + layer * lay.layer_stride + row * g.H + ch, -1)
    NKB = NT * KBU

    def second_gemm(name, rows, rows_p):
        gm, kb, lane, j = np.meshgrid(np.arange(rows_p // 32), np.arange(NKB), np.arange(64), np.arange(EPL), indexing="ij")
        i, h = lane & 31, lane >> 5
        ur = u_row_index(dtype, kb, h, j)
        r = 32 * gm + i
        return np.where((r < rows) & (ur < g.S), lay.off(name) + r * g.S + ur, -1)

    w1 = second_gemm("wavenet.last_conv_layers.1.weight_v", g.S, g.Sp)
    w3 = second_gemm("wavenet.last_conv_layers.3.weight_v", g.O, g.Op)
    return np.concatenate([w0.reshape(-1), w1.reshape(-1), w3.reshape(-1)]).astype(np.int32)


def head_bias_map(g: Geometry, lay: ParamLayout) -> np.ndarray:
    """[b1 (Sp) | b3 (Op)] -- stored after the Sp-float skip-bias sum produced by wae_sum_rows."""
    r = np.arange(g.Sp + g.Op)
    b1 = lay.off("wavenet.last_conv_layers.1.bias")
    b3 = lay.off("wavenet.last_conv_layers.3.bias")
    return np.where(r < g.Sp, np.where(r < g.S, b1 + r, -1), np.where(r - g.Sp < g.O, b3 + r - g.Sp, -1)).astype(np.int32)


def glu_packed_elems(g: Geometry, dtype: int) -> int:
    t = _traits(dtype)
    nph = glu_pass_tiles(g.NP)
    chb = 2 * nph * 4 * 1024
    nq1 = g.k * (g.Rp // t["CK"]) + g.Ccp // t["CK"]
    mt2 = t["MT2"] * nph // g.NP
    nq2 = (g.Rp // 32) // mt2
    return ((g.NP // nph) * nq1 + nq2) * chb // t["ES"]


def head_packed_elems(g: Geometry, dtype: int) -> int:
    t = _traits(dtype)
    chb = (g.Sp // 32) * 4 * 1024
    mt2 = 4 // t["KBU"]
    return (g.Ku // t["CK"] + (g.Sp // 32) // mt2 + (g.Op // 32) // mt2) * chb // t["ES"]


# ---------------------------------------------------------------------------------------------------
# autoregressive kernel (csrc/ar_fwd.hip): matrix-vector layout [k/EPL][rows padded to 64][EPL]
# ---------------------------------------------------------------------------------------------------
def _ar_block(rows, rows_valid_fn, K, src_fn, EPL):
    """generic blocked map: element (kb, r, j) -> src_fn(r, kb*EPL + j) or -1"""
    rp = _ru(rows, 64)
    nkb = (K + EPL - 1) // EPL
    kb, r, j = np.meshgrid(np.arange(nkb), np.arange(rp), np.arange(EPL), indexing="ij")
    kk = kb * EPL + j
    ok = (r < rows) & (kk < K)
    return np.where(ok, src_fn(np.minimum(r, rows - 1), np.minimum(kk, K - 1)), -1).reshape(-1), nkb * rp * EPL


def ar_layer_map(g: Geometry, lay: ParamLayout, dtype: int):
    """-> (map of one layer [W1 | W2], element offset of W2).  W1: G x (k*R + Cc) with k ordered
    [tap 0 .. tap k-1 | c] (tap j multiplies x[t-(k-1-j)d], conv.py:51-62); W2: [conv1x1_out (R) ; conv1x1_skip (S)] x H."""
    EPL = _traits(dtype)["EPL"]
    Cc = max(g.Cc, 0)
    K1 = g.k * g.R + Cc
    conv = lay.off("wavenet.conv_layers.0.conv.weight_v")
    cw = lay.off("wavenet.conv_layers.0.conv1x1c.weight_v") if Cc else 0

    def src1(r, kk):
        tap, ch = kk // g.R, kk % g.R
        return np.where(kk < g.k * g.R, conv + (r * g.R + ch) * g.k + np.minimum(tap, g.k - 1),
                        cw + r * max(Cc, 1) + (kk - g.k * g.R))

    m1, n1 = _ar_block(g.G, None, K1, src1, EPL)
    out = lay.off("wavenet.conv_layers.0.conv1x1_out.weight_v")
    skp = lay.off("wavenet.conv_layers.0.conv1x1_skip.weight_v")

    def src2(r, kk):
        return np.where(r < g.R, out + r * g.H + kk, skp + (r - g.R) * g.H + kk)

    m2, n2 = _ar_block(g.R + g.S, None, g.H, src2, EPL)
    return np.concatenate([m1, m2]).astype(np.int32), n1


def ar_bias2_map(g: Geometry, lay: ParamLayout) -> np.ndarray:
    r = np.arange(g.R + g.S)
    return np.where(r < g.R, lay.off("wavenet.conv_layers.0.conv1x1_out.bias") + r,
                    lay.off("wavenet.conv_layers.0.conv1x1_skip.bias") + r - g.R).astype(np.int32)


def ar_head_map(g: Geometry, lay: ParamLayout, dtype: int) -> np.ndarray:
    EPL = _traits(dtype)["EPL"]
    w1 = lay.off("wavenet.last_conv_layers.1.weight_v")
    w3 = lay.off("wavenet.last_conv_layers.3.weight_v")
    m1, _ = _ar_block(g.S, None, g.S, lambda r, kk: w1 + r * g.S + kk, EPL)
    m3, _ = _ar_block(g.O, None, g.S, lambda r, kk: w3 + r * g.S + kk, EPL)
    return np.concatenate([m1, m3]).astype(np.int32)


def ar_head_bias_map(g: Geometry, lay: ParamLayout) -> np.ndarray:
    r = np.arange(g.S + g.O)
    return np.where(r < g.S, lay.off("wavenet.last_conv_layers.1.bias") + r,
                    lay.off("wavenet.last_conv_layers.3.bias") + r - g.S).astype(np.int32)


def ar_ring_offsets(g: Geometry) -> np.ndarray:
    """float offsets of each layer's history ring ((k-1)*d+1 rows of R) inside one utterance's arena; last = total"""
    off = [0]
    for d in g.dilations:
        off.append(off[-1] + ((g.k - 1) * d + 1) * g.R)
    return np.asarray(off, dtype=np.int64)


# ---------------------------------------------------------------------------------------------------
# backward: transposed weight streams (csrc/gemm_tm.hip, csrc/head_bwd.hip) and the scatter maps that
# bring the dense weight-gradient tiles of csrc/gemm_tn.hip back into the flat gradient arena
# ---------------------------------------------------------------------------------------------------
def tm_slicing(Mp: int) -> Tuple[int, int]:
    """(tiles per slice, slices) of an Mp-row output of wae_gemm_tm (csrc/gemm_tm.hip: tm_slicing must agree)."""
    nt = Mp // 32
    if nt in (1, 2, 3, 4, 6, 8):
        return nt, 1
    for c in (8, 6, 4):
        if nt % c == 0:
            return c, nt // c
    raise ValueError(f"wae_gemm_tm: unsupported output width {Mp}")


def first_gemm_map(Mp: int, Kp: int, dtype: int, src_fn) -> np.ndarray:
    """[slice][q][blk][m][lane][j] stream of an (Mp x Kp) matrix; src_fn(row, kk) -> arena offset or -1 (vectorised).
    Outputs wider than 256 rows are cut into slices of tm_slicing(Mp) tiles, each with its own chunk stream."""
    t = _traits(dtype)
    EPL, CK = t["EPL"], t["CK"]
    assert Mp % 32 == 0 and Kp % CK == 0, (Mp, Kp)
    nts, nsl = tm_slicing(Mp)
    sl, q, blk, m, lane, j = np.meshgrid(np.arange(nsl), np.arange(Kp // CK), np.arange(4), np.arange(nts), np.arange(64),
                                         np.arange(EPL), indexing="ij")
    row = 32 * (sl * nts + m) + (lane & 31)
    kk = q * CK + blk * 2 * EPL + (lane >> 5) * EPL + j
    return src_fn(row, kk).astype(np.int32).reshape(-1)


def second_gemm_map(Mp: int, Kp: int, dtype: int, src_fn) -> np.ndarray:
    """[gm][kb][lane][j] stream; k index follows the accumulator-tile row order (u_row_index)."""
    t = _traits(dtype)
    EPL, KBU = t["EPL"], t["KBU"]
    assert Mp % 32 == 0 and Kp % 32 == 0
    gm, kb, lane, j = np.meshgrid(np.arange(Mp // 32), np.arange(Kp // 32 * KBU), np.arange(64), np.arange(EPL), indexing="ij")
    row = 32 * gm + (lane & 31)
    kk = u_row_index(dtype, kb, lane >> 5, j)
    return src_fn(row, kk).astype(np.int32).reshape(-1)


def _gate_row(g: Geometry, c2):
    """padded gate column (0..2Hp) -> (reference row of the (G, ...) conv weight, valid)"""
    half = (c2 >= g.Hp).astype(np.int64)
    i = c2 - half * g.Hp
    return half * g.H + i, i < g.H


def bwd_u_map(g: Geometry, lay: ParamLayout, dtype: int) -> np.ndarray:
    """du = W_out^T dx-hat + W_skip^T dskip: rows h (Hp), K = [Rp | Sp]."""
    o = lay.off("wavenet.conv_layers.0.conv1x1_out.weight_v")
    s = lay.off("wavenet.conv_layers.0.conv1x1_skip.weight_v")

    def src(h, kk):
        r, sk = kk, kk - g.Rp
        return np.where(kk < g.Rp, np.where((r < g.R) & (h < g.H), o + r * g.H + h, -1),
                        np.where((sk < g.S) & (h < g.H), s + sk * g.H + h, -1))
    return first_gemm_map(g.Hp, g.Rp + g.Sp, dtype, src)


def bwd_uo_map(g: Geometry, lay: ParamLayout, dtype: int) -> np.ndarray:
    """W_out^T for the fused boundary kernel (csrc/glu_bwd.hip): rows h (Hp), K = Rp in accumulator-row order (the
    operand is the just-computed dx-hat tile stack)."""
    o = lay.off("wavenet.conv_layers.0.conv1x1_out.weight_v")
    return second_gemm_map(g.Hp, g.Rp, dtype, lambda h, r: np.where((r < g.R) & (h < g.H), o + r * g.H + h, -1))


TM_INTERLEAVE = 1    # include/wae.h: WAE_TM_INTERLEAVE


def bwd_x_map(g: Geometry, lay: ParamLayout, dtype: int, interleave: bool = True) -> np.ndarray:
    """dx = sum_tap W1_tap^T dz[t + (k-1-tap) d]: rows r (Rp), K = k sources of 2Hp gate columns, visited round-robin per
    128-byte column block (WAE_TM_INTERLEAVE): chunk q = cblk * k + tap; interleave=False: tap by tap (csrc/glu_bwd.hip)."""
    conv = lay.off("wavenet.conv_layers.0.conv.weight_v")
    CK = _traits(dtype)["CK"]

    def src(r, kk):
        q, within = kk // CK, kk % CK
        if interleave:
            tap, c2 = q % g.k, (q // g.k) * CK + within
        else:
            tap, c2 = kk // (2 * g.Hp), kk % (2 * g.Hp)
        row, ok = _gate_row(g, c2)
        return np.where(ok & (r < g.R), conv + (row * g.R + r) * g.k + tap, -1)
    return first_gemm_map(g.Rp, g.k * 2 * g.Hp, dtype, src)


def bwd_c_map(g: Geometry, lay: ParamLayout, dtype: int) -> np.ndarray:
    """dc = sum_l Wc_l^T dz_l: rows cc (Ccp), K = L * 2Hp (absolute offsets).  (g_local: rows [Cc, Cc + Cg) are conv1x1g's.)"""
    def src(cc, kk):
        l, c2 = kk // (2 * g.Hp), kk % (2 * g.Hp)
        row, ok = _gate_row(g, c2)
        off, okc = cond_weight_src(g, lay, row, cc)
        return np.where(ok & okc, off + l * lay.layer_stride, -1)
    return first_gemm_map(g.Ccp, g.layers * 2 * g.Hp, dtype, src)


def head_bwd_map(g: Geometry, lay: ParamLayout, dtype: int) -> np.ndarray:
    w3 = lay.off("wavenet.last_conv_layers.3.weight_v")
    w1 = lay.off("wavenet.last_conv_layers.1.weight_v")
    a = first_gemm_map(g.Op, g.Sp, dtype, lambda o, kk: np.where((o < g.O) & (kk < g.S), w3 + o * g.S + kk, -1))
    b = second_gemm_map(g.Sp, g.Op, dtype, lambda s, kk: np.where((s < g.S) & (kk < g.O), w3 + kk * g.S + s, -1))
    c = second_gemm_map(g.Sp, g.Sp, dtype, lambda s, kk: np.where((s < g.S) & (kk < g.S), w1 + kk * g.S + s, -1))
    return np.concatenate([a, b, c]).astype(np.int32)


def head_wide_maps(g: Geometry, lay: ParamLayout, dtype: int) -> Dict[str, np.ndarray]:
    """Weight streams of the wide head (Sp > 256; wae_gemm_tm modes 3-6), all in first-GEMM order:
    skip = all layers' conv1x1_skip over K = Ku (layer, gated channel); w1 / w3 = last_conv_layers.1 / .3;
    w3t / w1t = their transposes for the backward data path."""
    ws = lay.off("wavenet.conv_layers.0.conv1x1_skip.weight_v")
    w1 = lay.off("wavenet.last_conv_layers.1.weight_v")
    w3 = lay.off("wavenet.last_conv_layers.3.weight_v")

    def skip_src(row, kk):
        layer, ch = kk // g.Hp, kk % g.Hp
        return np.where((row < g.S) & (layer < g.layers) & (ch < g.H), ws + layer * lay.layer_stride + row * g.H + ch, -1)
    return dict(
        skip=first_gemm_map(g.Sp, g.Ku, dtype, skip_src),
        w1=first_gemm_map(g.Sp, g.Sp, dtype, lambda r, kk: np.where((r < g.S) & (kk < g.S), w1 + r * g.S + kk, -1)),
        w3=first_gemm_map(g.Op, g.Sp, dtype, lambda o, kk: np.where((o < g.O) & (kk < g.S), w3 + o * g.S + kk, -1)),
        w3t=first_gemm_map(g.Sp, g.Op, dtype, lambda s_, kk: np.where((s_ < g.S) & (kk < g.O), w3 + kk * g.S + s_, -1)),
        w1t=first_gemm_map(g.Sp, g.Sp, dtype, lambda s_, kk: np.where((s_ < g.S) & (kk < g.S), w1 + kk * g.S + s_, -1)),
    )


def head_is_wide(g: Geometry) -> bool:
    """The register-chained head kernels hold Sp/32 <= 8 accumulator tiles per wave; wider heads (or EngineOptions.head_wide, for
    tests) run as separate wae_gemm_tm launches."""
    return g.Sp > 256


ONES_PAD = 128   # spare C columns that receive the per-clip "ones column" sums (B <= 128)


def grad_scatter_maps(g: Geometry, lay: ParamLayout) -> dict:
    """C-buffer element -> gradient-arena offset (or -1).  Layer maps are relative to layer 0.  ``*_w`` maps cover the
    weight sub-tile (rows x cols, every slot written once: plain adds); ``*_b`` maps cover the ONES_PAD ones-columns
    (rows x ONES_PAD, every clip column adds into the same bias slot: atomics)."""
    out = {}
    ones = np.zeros((1, ONES_PAD), dtype=np.int64)
    # dW1 | dWc : rows = padded gate rows (2Hp), cols = [tap*Rp + r | k*Rp + cc]  (the ones columns feed gproj_bwd)
    ld1 = g.k * g.Rp + g.Ccp + ONES_PAD
    ncol1 = g.k * g.Rp + g.Ccp
    row, col = np.meshgrid(np.arange(2 * g.Hp), np.arange(ncol1), indexing="ij")
    grow, ok = _gate_row(g, row)
    conv = lay.off("wavenet.conv_layers.0.conv.weight_v")
    tap, r = col // g.Rp, col % g.Rp
    m = np.where(ok & (col < g.k * g.Rp) & (r < g.R), conv + (grow * g.R + r) * g.k + np.minimum(tap, g.k - 1), -1)
    if g.Ccp:
        cc = col - g.k * g.Rp
        off, okc = cond_weight_src(g, lay, grow, cc)
        m = np.where(ok & okc, off, m)
    out["w1"], out["ld1"], out["ncol1"] = m.astype(np.int32).reshape(-1), ld1, ncol1
    # dW_out: rows r (Rp), cols h (Hp) ; ones -> bias
    ldo = g.Hp + ONES_PAD
    row, col = np.meshgrid(np.arange(g.Rp), np.arange(g.Hp), indexing="ij")
    wo = lay.off("wavenet.conv_layers.0.conv1x1_out.weight_v")
    bo = lay.off("wavenet.conv_layers.0.conv1x1_out.bias")
    out["wo"], out["ldo"] = np.where((row < g.R) & (col < g.H), wo + row * g.H + col, -1).astype(np.int32).reshape(-1), ldo
    rowb = np.arange(g.Rp)[:, None] + ones
    out["bo"] = np.where(rowb < g.R, bo + rowb, -1).astype(np.int32).reshape(-1)
    # dW_skip of all layers: rows s (Sp), cols l*Hp + h ; ones -> skip bias (applied per layer separately)
    lds = g.Ku + ONES_PAD
    row, col = np.meshgrid(np.arange(g.Sp), np.arange(g.Ku), indexing="ij")
    ws = lay.off("wavenet.conv_layers.0.conv1x1_skip.weight_v")
    l, hh = col // g.Hp, col % g.Hp
    m = np.where((row < g.S) & (col < g.layers * g.Hp) & (hh < g.H), ws + l * lay.layer_stride + row * g.H + hh, -1)
    out["ws"], out["lds"] = m.astype(np.int32).reshape(-1), lds
    bs = lay.off("wavenet.conv_layers.0.conv1x1_skip.bias")
    rowb = np.arange(g.Sp)[:, None] + ones
    out["bs"] = np.where(rowb < g.S, bs + rowb, -1).astype(np.int32).reshape(-1)   # (Sp, ONES_PAD) ones columns, layer 0
    # The static weight-gradient launch (csrc/gemm_tn_static.hip, kind OUTSKIP) writes dW_out and dW_skip of a layer TRANSPOSED --
    # rows = gated channel h (Hp of them), columns = residual / skip channel -- and the out bias as ONE row behind the Hp rows.
    hrow, rcol = np.meshgrid(np.arange(g.Hp), np.arange(g.Rp), indexing="ij")
    out["woT"] = np.where((rcol < g.R) & (hrow < g.H), wo + rcol * g.H + hrow, -1).astype(np.int32).reshape(-1)
    rr = np.arange(g.Rp)
    out["boT"] = np.where(rr < g.R, bo + rr, -1).astype(np.int32)
    out["ldoT_rows"] = g.Hp + 8               # rows of a layer's block: Hp weight rows, the bias row, padding
    hrow, scol = np.meshgrid(np.arange(g.Hp), np.arange(g.Sp), indexing="ij")
    out["wsT"] = np.where((scol < g.S) & (hrow < g.H), ws + scol * g.H + hrow, -1).astype(np.int32).reshape(-1)   # layer 0
    ss_ = np.arange(g.Sp)
    out["bsT"] = np.where(ss_ < g.S, bs + ss_, -1).astype(np.int32)          # skip bias from a plain (Sp,) vector of sums, layer 0
    # head
    ldh = g.Sp + ONES_PAD
    row, col = np.meshgrid(np.arange(g.Op), np.arange(g.Sp), indexing="ij")
    w3, b3 = lay.off("wavenet.last_conv_layers.3.weight_v"), lay.off("wavenet.last_conv_layers.3.bias")
    out["w3"] = np.where((row < g.O) & (col < g.S), w3 + row * g.S + col, -1).astype(np.int32).reshape(-1)
    rowb = np.arange(g.Op)[:, None] + ones
    out["b3"] = np.where(rowb < g.O, b3 + rowb, -1).astype(np.int32).reshape(-1)
    row, col = np.meshgrid(np.arange(g.Sp), np.arange(g.Sp), indexing="ij")
    w1, b1 = lay.off("wavenet.last_conv_layers.1.weight_v"), lay.off("wavenet.last_conv_layers.1.bias")
    out["w1h"] = np.where((row < g.S) & (col < g.S), w1 + row * g.S + col, -1).astype(np.int32).reshape(-1)
    rowb = np.arange(g.Sp)[:, None] + ones
    out["b1h"] = np.where(rowb < g.S, b1 + rowb, -1).astype(np.int32).reshape(-1)
    out["ldh"] = ldh
    # first conv table gradient (O or 1, Rp) -> weight_v (R, O, 1)
    nin = 1 if g.scalar_input else g.O
    o, r = np.meshgrid(np.arange(nin), np.arange(g.Rp), indexing="ij")
    out["tab"] = np.where(r < g.R, lay.off("wavenet.first_conv.weight_v") + r * nin + o, -1).astype(np.int32).reshape(-1)
    rr = np.arange(g.Rp)
    out["fb"] = np.where(rr < g.R, lay.off("wavenet.first_conv.bias") + rr, -1).astype(np.int32)
    return out
