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
    """(x', s) = layer(x, c, g) with gradients for x, c, every parameter and -- when the layer runs with the global features as a time
    series (Geometry.g_local) -- for g."""

    @staticmethod
    def forward(ctx, mod, x, c, g, *params):
        xo, so = mod._run(x, c, g, train=True)
        eng = mod._engine
        eng.fwd_gen = getattr(eng, "fwd_gen", 0) + 1           # the saved activations now belong to THIS forward
        ctx.mod, ctx.gen, ctx.shape, ctx.has_c = mod, eng.fwd_gen, tuple(x.shape), c is not None
        ctx.gvec = mod._keep[1]
        ctx.has_g = g is not None and eng.g.g_local
        ctx.seed = mod._keep[2]                 # dropout seed of THIS forward (None: no mask)
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
        B, R, T0 = ctx.shape
        lead = mod.lead                 # non-causal: the kernels' frame is `lead` steps longer, outputs sit at [lead, T0 + lead)
        T = T0 + lead
        dev = dxo.device if dxo is not None else dso.device

        def late(a):
            return a if not lead else torch.cat([a.new_zeros(a.shape[0], a.shape[1], lead), a], dim=2)
        gx = torch.zeros(B * T + lead, gm.Rp, dtype=eng.tdtype, device=dev)[:B * T].view(B, T, gm.Rp)   # (+ lead rows: read, never used)
        ds = torch.zeros(B, T, gm.Sp, dtype=eng.tdtype, device=dev)
        if dxo is not None:   # x' = (conv1x1_out(u) + x) sqrt(.5): the kernels take sqrt(.5) * d loss / d x' (backward.py: "hat")
            L.check(lib.wae_to_btc(L.ptr(late(dxo.float() * (math.sqrt(0.5) * eng.grad_scale)).contiguous()), L.ptr(gx), B, gm.R, T, gm.Rp, eng.dt, st), "to_btc dx'")
        if dso is not None:
            L.check(lib.wae_to_btc(L.ptr(late(dso.float() * eng.grad_scale).contiguous()), L.ptr(ds), B, gm.S, T, gm.Sp, eng.dt, st), "to_btc ds")
        dx_btc, dc_btc = BW.layer_backward(eng, B, T, gx, ds, ctx.gvec, drop_seed=ctx.seed, lead=lead)
        dx = torch.empty(B, gm.R, T, dtype=torch.float32, device=dev)
        L.check(lib.wae_from_btc_scaled(L.ptr(dx_btc), L.ptr(dx), B, gm.R, T, gm.Rp, eng.dt, 1.0 / eng.grad_scale, st), "from_btc dx")
        if lead:
            dx = dx[:, :, :T0].contiguous()        # (the convolution operand's frame: x sits at [0, T0))
        dc = dg = None
        if (ctx.has_c or ctx.has_g) and dc_btc is not None:
            # the conditioning operand's gradient: columns [0, Cc) are c's, with g as a time series [Cc, Cc + Cg) are g's
            dcg = torch.empty(B, gm.Cx, T, dtype=torch.float32, device=dev)
            L.check(lib.wae_from_btc_scaled(L.ptr(dc_btc), L.ptr(dcg), B, gm.Cx, T, gm.Ccp, eng.dt, 1.0 / eng.grad_scale, st), "from_btc dc")
            nc = max(gm.Cc, 0)
            if ctx.has_c:
                dc = dcg[:, :nc, lead:].contiguous()
            if ctx.has_g:
                dg = dcg[:, nc:nc + gm.Cg, lead:].contiguous()
        _, views = mod._grad_views(eng)
        return (None, dx, dc, dg) + tuple(v.clone() for v in views)


class ResidualConv1dGLU(ArenaModel):
    """x' , s = layer(x, c, g):  z = conv_dilated(x) + conv1x1c(c) + conv1x1g(g); u = tanh(z_a) * sigmoid(z_b);
    s = conv1x1_skip(u); x' = (conv1x1_out(u) + x) * sqrt(.5)   (modules.py:115-163).

    Same constructor as the reference (modules.py:71-75).  Supported: causal=True (what the reference's WaveNet builds,
    wavenet.py:127-134) and causal=False with an odd kernel size (symmetric padding: the causal kernels on a frame shifted by
    (k-1)/2 * d, forward and backward; no incremental_forward), bias=True or False; dropout (modules.py:127-128): the identity in eval mode, in training mode the engine's counter-based mask (one
    seed per forward call; parity against the oracle under the same mask, oracle.wae_oracle.dropout_keep); global features as ONE vector per
    clip (what the reference's WaveNet expands, wavenet.py:185-194: hoisted into a per-clip bias) or -- round 5 -- as any (B, Cg, T) time
    series, with a gradient for g (modules.py:148-152 convolves whatever it is given): the first call that sees a g that varies over
    time or requires a gradient switches the layer to Geometry.g_local, where conv1x1g's columns ride behind conv1x1c's in the kernel's
    conditioning operand.  Trainable: gradients for x, c, g and every parameter (weight_g / weight_v / bias of conv, conv1x1c, conv1x1g,
    conv1x1_out, conv1x1_skip) through ``_LayerFn``."""

    def __init__(self, residual_channels, gate_channels, kernel_size, skip_out_channels=None, cin_channels=-1, gin_channels=-1,
                 dropout=1 - 0.95, padding=None, dilation=1, causal=True, bias=True, *args, **kwargs):
        super().__init__()
        self.dropout = float(dropout)          # identity in eval mode (modules.py:127-128); the engine's hashed mask in train mode
        # causal=False (modules.py:82-88,134-136): the convolution pads (k - 1) // 2 * d on both sides and nothing is trimmed, i.e.
        # z[t] reads x[t + (j - (k-1)/2) d].  The kernels are causal; the non-causal layer runs them on a frame that is `lead` steps
        # longer: convolution operand [x ; 0], residual operand / conditioning / outputs [0 ; .] (see _run), so no kernel changes.
        self.causal = bool(causal)
        if not self.causal and kernel_size % 2 == 0:
            raise NotImplementedError("causal=False needs an odd kernel_size (the reference's own residual addition fails for even ones: "
                                      "the symmetric padding (k - 1) // 2 * d shortens the output)")
        self.lead = 0 if self.causal else (kernel_size - 1) // 2 * dilation
        self.bias = bool(bias)       # modules.py:88-107: bias=False builds conv, conv1x1_out and conv1x1_skip without one
        want_pad = (kernel_size - 1) * dilation if self.causal else (kernel_size - 1) // 2 * dilation
        if padding is not None and padding != want_pad:
            raise NotImplementedError("padding must be the layer's own default: (kernel_size - 1) * dilation (causal) or "
                                      "(kernel_size - 1) // 2 * dilation (causal=False)")
        if skip_out_channels is None:
            skip_out_channels = residual_channels                     # modules.py:80-81
        self.kernel_size, self.dilation = kernel_size, dilation
        geom = P.Geometry(layers=1, stacks=1, R=residual_channels, G=gate_channels, S=skip_out_channels, O=2,
                          Cc=cin_channels, Cg=gin_channels, k=kernel_size, n_speakers=None, use_speaker_embedding=False,
                          dilations_override=[dilation])
        # (bias=False: the three bias slots of the engine's arena exist, stay at zero and are not parameters of this module)
        self._init_arena(geom, "wavenet.conv_layers.0.", skip=None if self.bias else (lambda rel: rel.endswith(".bias")))
        self._buf = None
        self._skip_w = None

    # ------------------------------------------------------------------ kernels
    def _g_as_time_series(self):
        """Switch the layer to Geometry.g_local (a new engine over the same parameters; one-way)."""
        import dataclasses
        if self.geom.g_local:
            return
        sd = {k: v.detach().clone() for k, v in self.state_dict().items()}
        self._engine = None
        for k, p_ in self.named_parameters():
            p_.data = sd[k]
        self.geom = dataclasses.replace(self.geom, g_local=True)
        self._skip_w = None

    def _run(self, x, c, g, train=False):
        gm0 = self.geom
        if gm0.Cg > 0 and g is not None and not gm0.g_local:
            if g.requires_grad or (g.shape[-1] > 1 and not bool((g == g[:, :, :1]).all())):
                self._g_as_time_series()
        eng = self.engine()
        gm, lib, st = eng.g, eng.lib, eng.stream()
        B, R, T0 = x.shape
        assert R == gm.R, f"x has {R} channels, the layer {gm.R}"
        lead = self.lead
        xnc = None
        if lead:
            # non-causal: frame t' = t + lead.  The convolution operand keeps x at [0, T) (its causal taps t' - (k-1-j) d are then
            # x[t + (j - (k-1)/2) d], zeros beyond either end); everything that is point-wise in time moves to [lead, T + lead)
            def late(a):
                return None if a is None else torch.cat([a.new_zeros(a.shape[0], a.shape[1], lead), a.detach().float()], dim=2)
            if c is not None and c.shape[-1] != T0:
                raise ValueError("local conditioning c must be (B, cin_channels, T)")
            if g is not None and g.shape[-1] not in (1, T0):
                raise ValueError("global features g must be (B, gin_channels, T) or (B, gin_channels, 1)")
            xnc = torch.cat([x.detach().float(), x.new_zeros(B, R, lead, dtype=torch.float32)], dim=2)
            x, c = late(x), late(c)
            if g is not None and g.shape[-1] > 1:
                g = late(g) if gm0.g_local or self.geom.g_local else torch.cat([g[:, :, :1].expand(-1, -1, lead), g], dim=2)
        T = T0 + lead
        eng.prepare_weights()
        ws = eng.workspace(B, T, train)
        L.check(lib.wae_to_btc(L.ptr(x.contiguous().float()), L.ptr(ws["x"][0]), B, gm.R, T, gm.Rp, eng.dt, st), "to_btc x")
        if lead:
            if "xnc" not in ws:
                ws["xnc"] = torch.empty_like(ws["x"][0])
            L.check(lib.wae_to_btc(L.ptr(xnc.contiguous()), L.ptr(ws["xnc"]), B, gm.R, T, gm.Rp, eng.dt, st), "to_btc conv operand")
        gvec = None
        if gm.g_local:
            if g is None:
                raise ValueError("this layer has been run with global features as a time series: g (B, gin_channels, T) is required")
            gt = g.detach().float()
            if gt.shape[-1] == 1:
                gt = gt.expand(-1, -1, T)
            if gt.shape[-1] != T:
                raise ValueError("global features g must be (B, gin_channels, T) or (B, gin_channels, 1)")
            if gm.Cc > 0 and (c is None or c.shape[-1] != T):
                raise ValueError("local conditioning c must be (B, cin_channels, T)")
            cg = torch.cat([c.detach().float(), gt], dim=1) if gm.Cc > 0 else gt       # the kernel's conditioning operand [c ; g]
            L.check(lib.wae_to_btc(L.ptr(cg.contiguous()), L.ptr(ws["c_up"]), B, gm.Cx, T, gm.Ccp, eng.dt, st), "to_btc [c ; g]")
        else:
            if gm.Ccp:
                if c is None or c.shape[-1] != T:
                    raise ValueError("local conditioning c must be (B, cin_channels, T)")
                L.check(lib.wae_to_btc(L.ptr(c.contiguous().float()), L.ptr(ws["c_up"]), B, gm.Cc, T, gm.Ccp, eng.dt, st), "to_btc c")
            if gm.Cg > 0 and g is not None:
                gvec = g[:, :, 0].contiguous().float()
        wg_off = eng.lay.off("wavenet.conv_layers.0.conv1x1g.weight_v") if gm.Cg > 0 else -1
        L.check(lib.wae_gproj_fwd(L.ptr(eng.eff), wg_off if gvec is not None else -1, eng.lay.off("wavenet.conv_layers.0.conv.bias"),
                                  eng.lay.layer_stride, None, 0, L.ptr(gvec), L.ptr(ws["zb"]), B, 1, gm.G, gm.Hp, max(gm.Cg, 0), 0, None,
                                  st),
                "gproj")
        ws["u"].zero_()
        d = L.GluDesc(eng.dt, B, T, gm.Rp, gm.Ccp, gm.Hp, gm.k, self.dilation, L.GLU_SAVE_Z if train else 0)
        # dropout (modules.py:127-128): F.dropout on the convolution's operand in training mode, the residual path keeps x.  One seed per
        # forward call from the engine's call counter (engine.layer_drop_seed); backward regenerates the mask from the seed kept in ctx.
        seed = None
        xconv = xsrc = ws["xnc"] if lead else ws["x"][0]
        if self.training and self.dropout > 0:
            eng.drop_calls += 1
            seed = eng.layer_drop_seed(eng.drop_calls, 0)
        if seed is not None or (train and "xd" in ws):
            if "xd" not in ws:          # (an inference-shaped workspace: the masked operand lives for this call only)
                ws["xd_tmp"] = ws.get("xd_tmp") if ws.get("xd_tmp") is not None else torch.empty_like(ws["x"][0])
            xconv = ws["xd"][0] if "xd" in ws else ws["xd_tmp"]
            L.check(lib.wae_dropout_fwd(L.ptr(xsrc), L.ptr(xconv), B * T * gm.Rp, seed or 0, self.dropout if seed is not None else 0.0,
                                        eng.dt, st), "dropout")
        L.check(lib.wae_glu_layer_fwd_drop(ctypes.byref(d), L.ptr(ws["x"][0]), L.ptr(xconv), L.ptr(ws["x"][1]), L.ptr(ws["c_up"]),
                                           L.ptr(ws["u"]), gm.Ku, L.ptr(ws["zb"]), 2 * gm.Hp, L.ptr(ws["z"][0]) if train else None,
                                           L.ptr(eng.w_glu), L.ptr(eng.b_glu), st), "glu layer")
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
        self._keep = (sbt, gvec, seed)
        if lead:
            xo, so_ = xo[:, :, lead:].contiguous(), so_[:, :, lead:].contiguous()
        if not self.bias:
            return xo, so_
        bias = dict(self.named_parameters())["conv1x1_skip.bias"].detach().float()
        return xo, so_ + bias.view(1, -1, 1)

    # ------------------------------------------------------------------ reference API
    def forward(self, x, c=None, g=None):
        """x (B, R, T), c (B, Cc, T), g (B, Cg, T) -> (x' (B, R, T), s (B, S, T))   (modules.py:109-110)."""
        if torch.is_grad_enabled():
            params = [p for _, p in sorted(((n, p) for n, p in self.named_parameters()), key=lambda kv: self._pnames.index(kv[0]))]
            wants = any(t is not None and t.requires_grad for t in (x, c, g)) or any(p.requires_grad for p in params)
            if wants:
                return _LayerFn.apply(self, x, c, g, *params)
        with torch.no_grad():
            return self._run(x, c, g)

    def incremental_forward(self, x, c=None, g=None):
        """One time step: x (B, 1, R), c (B, 1, Cc), g (B, 1, Cg) -> (x' (B, 1, R), s (B, 1, S)); the last (k-1)*d inputs
        are kept between calls (conv.py:17-46 keeps the same window)."""
        if self.training:
            raise RuntimeError("incremental_forward only supports eval mode")          # conv.py:19-20
        if not self.causal:
            raise NotImplementedError("incremental_forward of a causal=False layer (it would need future inputs)")
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
