"""Drop-in ``WaveNet`` (reference: wavenet_vocoder/wavenet.py:63-364) on the MI355X engine.

Same constructor arguments, same ``state_dict`` keys (``first_conv.weight_g`` ...), same call conventions; the
arithmetic is libwae_hip.so.  Inputs must live on a ROCm GPU -- there is no CPU path.
"""
import math

import torch
from torch import nn

from .. import packing as P
from ._base import ArenaModel


def softmax_bct(y):
    """F.softmax(y, dim=1) of (B, C, T) logits on the HIP path (losses.softmax_bct; imported late: losses imports this package)."""
    from ..losses import softmax_bct as f
    return f(y)


def receptive_field_size(total_layers, num_cycles, kernel_size, dilation=lambda x: 2 ** x):
    """(kernel_size - 1) * sum(dilations) + 1 (wavenet.py:42-60)."""
    assert total_layers % num_cycles == 0
    per = total_layers // num_cycles
    return (kernel_size - 1) * sum(dilation(i % per) for i in range(total_layers)) + 1


def _ids_from_input(x, out_channels, scalar_input, eng=None):
    """Reference inputs are one-hot (B, C, T) floats (or (B, 1, T) scalars); the kernels take class ids.  wavenet.py:203 would run
    first_conv densely on soft labels too: a (B, C, T) input that is not one-hot is REFUSED (WaeEngine.check_errors raises at the end
    of the call), never arg-maxed."""
    if scalar_input:
        return x.reshape(x.shape[0], -1).float()
    if x.dim() == 3:
        if x.shape[1] != out_channels and x.shape[2] == out_channels:
            x = x.transpose(1, 2)
        if eng is None:
            raise RuntimeError("one-hot inputs are converted on the GPU: pass the engine")
        return eng.ids_from_onehot(x)
    return x.to(torch.int32)


def _start_classes(initial_input, out_channels, eng):
    """initial_input (B, C, 1) / (B, 1, C) one-hot rows -> one start class per utterance (wavenet.py:283-297); None -> 127 (:288)."""
    if initial_input is None:
        return 127
    ii = initial_input
    if ii.dim() == 3:
        ii = ii.reshape(ii.shape[0], -1)
    ids = eng.ids_from_onehot(ii.reshape(ii.shape[0], out_channels, 1)).reshape(-1)
    eng.check_errors()
    return ids


class _DecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, ids, c, g, c_is_up, train, *params):
        eng = model.engine()
        ctx.model, ctx.ids, ctx.g, ctx.c_is_up = model, ids, g, c_is_up
        ctx.c_shape = None if c is None else tuple(c.shape)
        ctx.train = train
        gid = g if (g is not None and g.dtype in (torch.int32, torch.int64)) else None
        gvec = None if (g is None or gid is not None) else g.reshape(g.shape[0], -1).float().contiguous()
        ctx.gid, ctx.gvec = gid, gvec
        out = eng.decoder_forward(ids, c, gid, want_logits=True, train=train, c_is_upsampled=c_is_up, gvec=gvec,
                                  dropout_on=model.training)
        ctx.gen = getattr(eng, "fwd_gen", 0)
        eng.check_errors()     # an out-of-range class / speaker id raises IndexError here, as nn.Embedding / the one-hot encoder do
        return out["logits"]

    @staticmethod
    def backward(ctx, dy):
        from .. import _lib as L
        from .. import backward as BW
        model = ctx.model
        eng = model._engine
        if ctx.train and getattr(eng, "fwd_gen", 0) != ctx.gen:
            # the activations live in the engine's per-(B, T) workspace, not in this autograd node
            raise RuntimeError("backward through a forward whose saved activations were overwritten by a later training-mode forward "
                               "of the same model: call backward before the next forward")
        g = eng.g
        B, O, T = dy.shape
        ext = torch.zeros(B, T, g.Op, dtype=eng.tdtype, device=dy.device)
        dyc = dy.contiguous().float()
        if eng.grad_scale != 1.0:               # fp16 stack: its backward runs on loss-scaled gradients (engine.py: grad_scale)
            dyc = dyc * eng.grad_scale
        L.check(eng.lib.wae_to_btc(L.ptr(dyc), L.ptr(ext), B, O, T, g.Op, eng.dt, eng.stream()), "to_btc")
        dc = BW.decoder_backward(eng, ctx.ids, None, None, ctx.gid, ctx.gvec, ext_dy=ext)
        dc_in = None
        if ctx.c_shape is not None:
            if ctx.c_is_up or not g.upsample_scales:
                dc_in = torch.empty(B, g.Cc, T, dtype=torch.float32, device=dy.device)
                L.check(eng.lib.wae_from_btc_scaled(L.ptr(dc), L.ptr(dc_in), B, g.Cc, T, g.Ccp, eng.dt, 1.0 / eng.grad_scale,
                                                    eng.stream()), "from_btc")
            else:
                dc_in = BW.frontend_backward(eng, dc, 1.0, stop_at_quant=True)
        grads = BW.finish_grads(eng)
        _, views = model._grad_views(eng)
        return (None, None, dc_in, None, None, None) + tuple(v.clone() for v in views)


class WaveNet(ArenaModel):
    def __init__(self, out_channels=256, layers=20, stacks=2, residual_channels=512, gate_channels=512,
                 skip_out_channels=512, kernel_size=3, dropout=1 - 0.95, cin_channels=-1, gin_channels=-1, n_speakers=None,
                 upsample_conditional_features=False, upsample_net="ConvInUpsampleNetwork",
                 upsample_params={"upsample_scales": [4, 4, 4, 4]}, scalar_input=False, use_speaker_embedding=False,
                 output_distribution="Logistic", cin_pad=0, _prefix=""):
        super().__init__()
        # modules.py:127-128 applies F.dropout(x, p, training=self.training) in front of every dilated convolution: identity in
        # eval mode for any p.  The reference's constructor default is 0.05 (wavenet.py:98-111); every shipped preset trains with
        # 0.0 (hps/*.json).  In training mode the engine applies a counter-based mask (wae_dropout_fwd / wae_dropout_bwd): same
        # distribution as torch's, not the same random stream.
        self.dropout = float(dropout)
        # wavenet.py:150-151: getattr(upsample, upsample_net)(**upsample_params) -- the two classes of upsample.py
        if upsample_conditional_features and upsample_net not in ("ConvInUpsampleNetwork", "UpsampleNetwork"):
            raise AttributeError(f"module 'wavenet_vocoder.upsample' has no attribute {upsample_net!r}")
        self.scalar_input = scalar_input
        self.out_channels = out_channels
        self.cin_channels = cin_channels
        self.output_distribution = output_distribution
        scales = None
        up_act, up_slope = "none", 0.01
        if upsample_conditional_features:
            # the reference builds ConvInUpsampleNetwork(**upsample_params) (wavenet.py:151, upsample.py:69-85): its signature is
            # (upsample_scales, upsample_activation="none", upsample_activation_params={}, mode="nearest",
            # freq_axis_kernel_size=1, cin_pad=0, cin_channels=80).  An option the kernels do not implement RAISES here -- it
            # would otherwise give silently different numbers (round-2 finding: every key but upsample_scales was dropped).
            known = {"upsample_scales", "upsample_activation", "upsample_activation_params", "mode", "freq_axis_kernel_size",
                     "cin_pad", "cin_channels"}
            unknown = sorted(set(upsample_params) - known)
            if unknown:
                raise TypeError(f"__init__() got an unexpected keyword argument {unknown[0]!r} (upsample_params)")
            if "upsample_scales" not in upsample_params:
                raise TypeError("__init__() missing 1 required positional argument: 'upsample_scales' (upsample_params)")
            up_act = upsample_params.get("upsample_activation", "none")
            act_params = dict(upsample_params.get("upsample_activation_params", {}) or {})
            act_params.pop("inplace", None)
            up_slope = 0.01
            if up_act == "LeakyReLU":
                up_slope = float(act_params.pop("negative_slope", 0.01))
            if up_act != "none" and (up_act not in P.UP_ACT_KINDS or act_params):
                raise NotImplementedError(f"upsample_activation={up_act!r} with parameters {act_params}: ReLU, LeakyReLU(negative_slope), Tanh "
                                          "and Sigmoid are implemented (upsample.py:44-46 takes any torch.nn module; no preset sets one)")
            if upsample_params.get("mode", "nearest") != "nearest":
                raise NotImplementedError("only nearest-neighbour stretching (Stretch2d mode='nearest', upsample.py:19-21) is implemented")
            if int(upsample_params.get("freq_axis_kernel_size", 1)) != 1:
                raise NotImplementedError("freq_axis_kernel_size must be 1 (a smoothing FIR along time only, upsample.py:36-40)")
            if int(upsample_params.get("cin_channels", cin_channels)) != cin_channels:
                raise ValueError(f"upsample_params cin_channels ({upsample_params['cin_channels']}) != cin_channels ({cin_channels}): "
                                 "conv_in would not accept the features")
            scales = list(upsample_params["upsample_scales"])
            # the network's own cin_pad is the one in upsample_params (the reference ignores the constructor argument for it)
            # -- ConvInUpsampleNetwork(**upsample_params) defaults it to 0 when the key is absent (upsample.py:72)
            cin_pad = int(upsample_params.get("cin_pad", 0))
        if not 1 <= int(kernel_size) <= 4:
            # the backward's tap sources are wae_gemm_tm's TM_MAX_SRC = 4; 1..4 taps are checked against the oracle (forward, backward,
            # incremental decoding: tests/test_gpu_parity.py::test_kernel_sizes_against_oracle); 3 is what every preset uses, and the
            # only tap count with static-schedule layer-kernel instantiations (the others run the run-time-scheduled kernel)
            raise NotImplementedError(f"kernel_size={kernel_size}: 1..4 taps are implemented and verified; wider kernels are refused")
        geom = P.Geometry(layers=layers, stacks=stacks, R=residual_channels, G=gate_channels, S=skip_out_channels,
                          O=out_channels, Cc=cin_channels, Cg=gin_channels, k=kernel_size,
                          n_speakers=n_speakers if (gin_channels > 0 and use_speaker_embedding) else None,
                          upsample_scales=scales, cin_pad=cin_pad, scalar_input=scalar_input,
                          use_speaker_embedding=bool(use_speaker_embedding), conv_in=upsample_net != "UpsampleNetwork",
                          up_act=up_act, up_act_slope=up_slope)
        self._init_arena(geom, "wavenet.")
        self.receptive_field = receptive_field_size(layers, stacks, kernel_size)

    # ------------------------------------------------------------------ reference API
    def has_speaker_embedding(self):
        return "embed_speakers" in self._modules

    def local_conditioning_enabled(self):
        return self.cin_channels > 0

    def forward(self, x, c=None, g=None, softmax=False):
        """x (B, C, T) one-hot / (B, 1, T) scalar / (B, T) ids; c (B, Cc, Tc); g (B,) ids or (B, Cg, 1) features.
        Returns (B, out_channels, T) logits (probabilities if softmax)  -- wavenet.py:164-216."""
        ids = _ids_from_input(x, self.out_channels, self.scalar_input, self.engine())
        if g is not None and g.dtype in (torch.int32, torch.int64):
            g = g.reshape(-1)
        c_is_up = not bool(self.geom.upsample_scales)
        params = [p for _, p in sorted(((n, p) for n, p in self.named_parameters()), key=lambda kv: self._pnames.index(kv[0]))]
        train = torch.is_grad_enabled() and (any(p.requires_grad for p in params) or (c is not None and c.requires_grad))
        if self.training and self.dropout > 0 and not train:
            # F.dropout is active whenever module.training is set, gradients or not: run the train-mode path (it is the one that
            # applies the mask) even when nothing requires a gradient
            train = True
        y = _DecoderFn.apply(self, ids, c, g, c_is_up, train, *params)
        return softmax_bct(y) if softmax else y

    def incremental_forward(self, initial_input=None, c=None, g=None, T=100, test_inputs=None, tqdm=lambda x: x,
                            softmax=True, quantize=True, log_scale_min=-50.0):
        """wavenet.py:218-346 as one persistent kernel launch: sampling (quantize=True) -> one-hot (B, C, T); quantize=False ->
        the logits / softmax rows (B, C, T), fed back as the next step's input once test_inputs is used up."""
        if self.training:
            raise RuntimeError("incremental_forward only supports eval mode")          # conv.py:19-20
        eng = self.engine()
        if self.scalar_input:
            # wavenet.py:284-285,325-333: every step draws from the mixture of logistics; the samples are returned (B, 1, T)
            gid = g.reshape(-1) if (g is not None and g.dtype in (torch.int32, torch.int64)) else None
            gvec = None if (g is None or gid is not None) else g.reshape(g.shape[0], -1).float().contiguous()
            tf = None
            if test_inputs is not None:
                tf = test_inputs
                if tf.dim() == 3:
                    tf = tf.reshape(tf.shape[0], -1)                                     # (B,1,T) or (B,T,1) -> (B,T)
                T = max(int(T or 0), tf.shape[1])                                        # wavenet.py:259-262
            c_is_up = c is not None and (not self.geom.upsample_scales or c.shape[-1] == int(T))
            B = c.shape[0] if c is not None else (tf.shape[0] if tf is not None else 1)
            dev = next(self.parameters()).device
            M = self.out_channels // 3
            with torch.no_grad():
                # every step draws from its mixture (wavenet.py:325-333); forced steps consume test_inputs instead of the draw
                out = eng.incremental_forward(c, gid, int(T), mode="sample", test_inputs=tf,
                                              c_is_upsampled=c_is_up, gvec=gvec, log_scale_min=log_scale_min,
                                              u_mix=torch.rand(B, int(T), M, device=dev) * (1 - 2e-5) + 1e-5,
                                              u_log=torch.rand(B, int(T), device=dev) * (1 - 2e-5) + 1e-5)
            return out["x"].unsqueeze(1)
        tf = None
        if test_inputs is not None:
            tf = _ids_from_input(test_inputs, self.out_channels, False, eng)
            eng.check_errors()                                                           # (rows that are not one-hot are refused)
            T = max(int(T or 0), tf.shape[1])                                            # wavenet.py:259-262
        T = int(T)
        nf = tf.shape[1] if tf is not None else 0
        init = _start_classes(initial_input, self.out_channels, eng)                    # wavenet.py:283-297 (None: class 127, :288)
        gid = g.reshape(-1) if (g is not None and g.dtype in (torch.int32, torch.int64)) else None
        gvec = None if (g is None or gid is not None) else g.reshape(g.shape[0], -1).float().contiguous()
        c_is_up = c is not None and (not self.geom.upsample_scales or c.shape[-1] == T)
        kw = dict(test_inputs=tf, n_forced=nf, init_idx=init, c_is_upsampled=c_is_up, gvec=gvec)
        with torch.no_grad():
            if quantize:
                if not softmax:
                    # the reference hands raw logits to OneHotCategorical(probs=...), which rejects negative entries
                    raise ValueError("quantize=True draws from the softmax probabilities: pass softmax=True")
                # every step's output is a draw from its distribution (wavenet.py:335-338), forced or not
                out = eng.incremental_forward(c, gid, T, mode="sample", **kw)
                return torch.nn.functional.one_hot(out["idx"].long(), self.out_channels).float().transpose(1, 2).contiguous()
            if nf >= T:                      # fully teacher-forced: logits (or probabilities) of every step
                y = eng.incremental_forward(c, gid, T, mode="logits", **kw)["logits"]
                return softmax_bct(y) if softmax else y
            # quantize=False, free-running: the probability (softmax=True) or logit vector itself is fed back (wavenet.py:303-305)
            return eng.incremental_forward(c, gid, T, mode="probs" if softmax else "raw", **kw)["logits"]

    def clear_buffer(self):
        """The per-layer history lives inside one kernel launch; nothing persists between calls (wavenet.py:348-356)."""

    def make_generation_fast_(self):
        """Weight norm is folded once into the packed weights at synthesis time (wavenet.py:358-364 removes the hooks);
        parameters keep their weight_g / weight_v form."""
        if self._engine is not None:
            self._engine._ar_packed = False
        return self
