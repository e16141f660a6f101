"""``wavenet_vocoder.modules.ResidualConv1dGLU`` (reference modules.py:71-169) as a stand-alone layer on the fused HIP
kernel, forward and backward (the reference's is an ordinary autograd module; here an autograd Function drives the kernels the
stack's backward uses: backward.layer_backward)."""
import ctypes
import math

import numpy as np
import torch

from .. import _lib as L
from .. import packing as P
from ._base import ArenaModel


class _LayerFn(torch.autograd.Function):
    """(x', s) = layer(x, c, g) with gradients for x, c and every parameter (g: a constant-over-time feature vector, no gradient)."""

    @staticmethod
    def forward(ctx, mod, x, c, g, *params):
        xo, so = mod._run(x, c, g, train=True)
        eng = mod._engine
        eng.fwd_gen = getattr(eng, "fwd_gen", 0) + 1           # the saved activations now belong to THIS forward
        ctx.mod, ctx.gen, ctx.shape, ctx.has_c = mod, eng.fwd_gen, tuple(x.shape), c is not None
        ctx.gvec = mod._keep[1]
        return xo, so

    @staticmethod
    def backward(ctx, dxo, dso):
        from .. import backward as BW
        mod = ctx.mod
        eng = mod._engine
        if getattr(eng, "fwd_gen", 0) != ctx.gen:
            raise RuntimeError("backward through a forward whose saved activations were overwritten by a later training-mode forward "
                               "of the same layer: call backward before the next forward")
        gm, lib, st = eng.g, eng.lib, eng.stream()
        B, R, T = ctx.shape
        dev = dxo.device if dxo is not None else dso.device
        gx = torch.zeros(B, T, gm.Rp, dtype=eng.tdtype, device=dev)
        ds = torch.zeros(B, T, gm.Sp, dtype=eng.tdtype, device=dev)
        if dxo is not None:   # x' = (conv1x1_out(u) + x) sqrt(.5): the kernels take sqrt(.5) * d loss / d x' (backward.py: "hat")
            L.check(lib.wae_to_btc(L.ptr((dxo.float() * (math.sqrt(0.5) * eng.grad_scale)).contiguous()), L.ptr(gx), B, gm.R, T, gm.Rp, eng.dt, st), "to_btc dx'")
        if dso is not None:
            L.check(lib.wae_to_btc(L.ptr((dso.float() * eng.grad_scale).contiguous()), L.ptr(ds), B, gm.S, T, gm.Sp, eng.dt, st), "to_btc ds")
        dx_btc, dc_btc = BW.layer_backward(eng, B, T, gx, ds, ctx.gvec)
        dx = torch.empty(B, gm.R, T, dtype=torch.float32, device=dev)
        L.check(lib.wae_from_btc_scaled(L.ptr(dx_btc), L.ptr(dx), B, gm.R, T, gm.Rp, eng.dt, 1.0 / eng.grad_scale, st), "from_btc dx")
        dc = None
        if ctx.has_c and dc_btc is not None:
            dc = torch.empty(B, gm.Cc, T, dtype=torch.float32, device=dev)
            L.check(lib.wae_from_btc_scaled(L.ptr(dc_btc), L.ptr(dc), B, gm.Cc, T, gm.Ccp, eng.dt, 1.0 / eng.grad_scale, st), "from_btc dc")
        _, views = mod._grad_views(eng)
        return (None, dx, dc, None) + tuple(v.clone() for v in views)


class ResidualConv1dGLU(ArenaModel):
    """x' , s = layer(x, c, g):  z = conv_dilated(x) + conv1x1c(c) + conv1x1g(g); u = tanh(z_a) * sigmoid(z_b);
    s = conv1x1_skip(u); x' = (conv1x1_out(u) + x) * sqrt(.5)   (modules.py:115-163).

    Same constructor as the reference (modules.py:71-75).  Supported: causal=True, bias=True (what the reference's WaveNet builds,
    wavenet.py:127-134), any dropout in eval mode and dropout 0 in train mode, global features constant over time (the reference expands one
    speaker vector, wavenet.py:185-194).  Trainable: gradients for x, c and every parameter (weight_g / weight_v / bias of conv,
    conv1x1c, conv1x1g, conv1x1_out, conv1x1_skip) through ``_LayerFn``."""

    def __init__(self, residual_channels, gate_channels, kernel_size, skip_out_channels=None, cin_channels=-1, gin_channels=-1,
                 dropout=1 - 0.95, padding=None, dilation=1, causal=True, bias=True, *args, **kwargs):
        super().__init__()
        self.dropout = float(dropout)          # identity in eval mode (modules.py:127-128); a train-mode call with p > 0 raises
        if not causal or not bias:
            raise NotImplementedError("only the causal, biased layer the reference's WaveNet builds is implemented")
        if padding is not None and padding != (kernel_size - 1) * dilation:
            raise NotImplementedError("padding must be the causal (kernel_size - 1) * dilation")
        if skip_out_channels is None:
            skip_out_channels = residual_channels                     # modules.py:80-81
        self.kernel_size, self.dilation = kernel_size, dilation
        geom = P.Geometry(layers=1, stacks=1, R=residual_channels, G=gate_channels, S=skip_out_channels, O=2,
                          Cc=cin_channels, Cg=gin_channels, k=kernel_size, n_speakers=None, use_speaker_embedding=False,
                          dilations_override=[dilation])
        self._init_arena(geom, "wavenet.conv_layers.0.")
        self._buf = None
        self._skip_w = None

    # ------------------------------------------------------------------ kernels
    def _run(self, x, c, g, train=False):
        if self.training and self.dropout > 0:
            raise NotImplementedError(f"training-mode forward with dropout={self.dropout} is not implemented (every preset uses 0.0; "
                                      "eval mode is exact for any value): pass dropout=0.0 or call .eval()")
        eng = self.engine()
        gm, lib, st = eng.g, eng.lib, eng.stream()
        B, R, T = x.shape
        assert R == gm.R, f"x has {R} channels, the layer {gm.R}"
        eng.prepare_weights()
        ws = eng.workspace(B, T, train)
        L.check(lib.wae_to_btc(L.ptr(x.contiguous().float()), L.ptr(ws["x"][0]), B, gm.R, T, gm.Rp, eng.dt, st), "to_btc x")
        if gm.Ccp:
            if c is None or c.shape[-1] != T:
                raise ValueError("local conditioning c must be (B, cin_channels, T)")
            L.check(lib.wae_to_btc(L.ptr(c.contiguous().float()), L.ptr(ws["c_up"]), B, gm.Cc, T, gm.Ccp, eng.dt, st), "to_btc c")
        gvec = None
        if gm.Cg > 0 and g is not None:
            if g.shape[-1] > 1 and not bool((g == g[:, :, :1]).all()):
                raise NotImplementedError("global features that vary over time are not supported (the reference expands one vector)")
            gvec = g[:, :, 0].contiguous().float()
        wg_off = eng.lay.off("wavenet.conv_layers.0.conv1x1g.weight_v") if gm.Cg > 0 else -1
        L.check(lib.wae_gproj_fwd(L.ptr(eng.eff), wg_off if gvec is not None else -1, eng.lay.off("wavenet.conv_layers.0.conv.bias"),
                                  eng.lay.layer_stride, None, 0, L.ptr(gvec), L.ptr(ws["zb"]), B, 1, gm.G, gm.Hp, max(gm.Cg, 0), 0, None,
                                  st),
                "gproj")
        ws["u"].zero_()
        d = L.GluDesc(eng.dt, B, T, gm.Rp, gm.Ccp, gm.Hp, gm.k, self.dilation, L.GLU_SAVE_Z if train else 0)
        L.check(lib.wae_glu_layer_fwd(ctypes.byref(d), L.ptr(ws["x"][0]), L.ptr(ws["x"][1]), L.ptr(ws["c_up"]), L.ptr(ws["u"]), gm.Ku,
                                      L.ptr(ws["zb"]), 2 * gm.Hp, L.ptr(ws["z"][0]) if train else None, L.ptr(eng.w_glu), L.ptr(eng.b_glu),
                                      st), "glu layer")
        # skip output: s = W_skip u + b as one time-major GEMM over the layer's gated activations
        if self._skip_w is None or self._skip_w[0] is not eng:
            lay = eng.lay
            so = lay.off("wavenet.conv_layers.0.conv1x1_skip.weight_v")
            m = P.first_gemm_map(gm.Sp, gm.Ku, eng.dt, lambda s, kk: np.where((s < gm.S) & (kk < gm.H), so + s * gm.H + kk, -1))
            mp = torch.from_numpy(m).to(eng.device)
            self._skip_w = (eng, mp, torch.zeros(mp.numel(), dtype=eng.tdtype, device=eng.device))
        _, mp, wbuf = self._skip_w
        L.check(lib.wae_pack_gather(L.ptr(eng.eff), L.ptr(mp), L.ptr(wbuf), mp.numel(), 1, 0, 0, eng.dt, st), "pack skip")
        sbt = torch.empty(B, T, gm.Sp, dtype=eng.tdtype, device=eng.device)
        td = L.TmDesc(eng.dt, B, T, gm.Sp, 1, 0, 1.0)
        ptrs = (ctypes.c_void_p * 1)(ws["u"].data_ptr())
        strides = (ctypes.c_int64 * 1)(gm.Ku)
        cols = (ctypes.c_int32 * 1)(gm.Ku)
        shifts = (ctypes.c_int32 * 1)(0)
        L.check(lib.wae_gemm_tm(ctypes.byref(td), ptrs, strides, cols, shifts, L.ptr(wbuf), L.ptr(sbt), gm.Sp, None, 0, st), "skip gemm")
        xo = torch.empty(B, gm.R, T, dtype=torch.float32, device=eng.device)
        so_ = torch.empty(B, gm.S, T, dtype=torch.float32, device=eng.device)
        L.check(lib.wae_from_btc(L.ptr(ws["x"][1]), L.ptr(xo), B, gm.R, T, gm.Rp, eng.dt, st), "from_btc x")
        L.check(lib.wae_from_btc(L.ptr(sbt), L.ptr(so_), B, gm.S, T, gm.Sp, eng.dt, st), "from_btc s")
        bias = dict(self.named_parameters())["conv1x1_skip.bias"].detach().float()
        self._keep = (sbt, gvec)
        return xo, so_ + bias.view(1, -1, 1)

    # ------------------------------------------------------------------ reference API
    def forward(self, x, c=None, g=None):
        """x (B, R, T), c (B, Cc, T), g (B, Cg, T) -> (x' (B, R, T), s (B, S, T))   (modules.py:109-110)."""
        if torch.is_grad_enabled():
            params = [p for _, p in sorted(((n, p) for n, p in self.named_parameters()), key=lambda kv: self._pnames.index(kv[0]))]
            wants = any(t is not None and t.requires_grad for t in (x, c)) or any(p.requires_grad for p in params)
            if g is not None and g.requires_grad:
                raise NotImplementedError("no gradient with respect to the global features g (a constant-over-time vector here); "
                                          "detach it, or train through wavenet_vocoder.WaveNet (speaker embedding)")
            if wants:
                return _LayerFn.apply(self, x, c, g, *params)
        with torch.no_grad():
            return self._run(x, c, g)

    def incremental_forward(self, x, c=None, g=None):
        """One time step: x (B, 1, R), c (B, 1, Cc), g (B, 1, Cg) -> (x' (B, 1, R), s (B, 1, S)); the last (k-1)*d inputs
        are kept between calls (conv.py:17-46 keeps the same window)."""
        if self.training:
            raise RuntimeError("incremental_forward only supports eval mode")          # conv.py:19-20
        win = (self.kernel_size - 1) * self.dilation + 1
        xt = x.transpose(1, 2).contiguous().float()                                     # (B, R, 1)
        if self._buf is None:
            self._buf = torch.zeros(x.shape[0], xt.shape[1], win, dtype=torch.float32, device=x.device)   # conv.py:35-36
        self._buf = torch.cat([self._buf[:, :, 1:], xt], dim=2)
        cw = c.transpose(1, 2).expand(-1, -1, win).contiguous() if c is not None else None
        gw = g.transpose(1, 2).expand(-1, -1, win).contiguous() if g is not None else None
        with torch.no_grad():
            xo, so = self._run(self._buf, cw, gw)
        return xo[:, :, -1:].transpose(1, 2).contiguous(), so[:, :, -1:].transpose(1, 2).contiguous()

    def clear_buffer(self):
        self._buf = None                                                                # modules.py:165-169
