"""WaeEngine: host-side driver of the HIP hot path (encoder -> VQ -> upsample -> gated stack -> head -> loss).

PyTorch is used for device memory, streams and (elsewhere) torch.distributed only; every arithmetic step below
is a call into libwae_hip.so through the C ABI of include/wae.h.  There is no CPU fallback.

Reference call stack being replaced: VQVAE.forward (vqvae_model.py:66-72) -> Encoder.forward (:48-51),
VectorQuantize.forward (vector_quantization.py:21-49), WaveNet.forward (wavenet.py:164-216).
"""
from __future__ import annotations

import contextlib
import ctypes
import math

from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L
from . import packing as P
from .options import two_chains


def _dt(dtype) -> int:
    if dtype in ("bf16", torch.bfloat16, L.WAE_BF16):
        return L.WAE_BF16
    if dtype in ("fp32", "f32", torch.float32, L.WAE_F32):
        return L.WAE_F32
    if dtype in ("fp16", "f16", "half", torch.float16, L.WAE_F16) and not isinstance(dtype, bool):
        return L.WAE_F16
    raise ValueError(f"unsupported compute dtype {dtype!r} (use 'fp32', 'bf16' or 'fp16')")


class WaeEngine:
    def __init__(self, geom: P.Geometry, dtype="bf16", device="cuda:0", dropout: float = 0.0, drop_seed: int = 0x5EED):
        """dtype: 'fp32' (exact, the parity mode), 'bf16' (default throughput mode) or 'fp16' (BASELINE config C5) storage of
        activations and packed weights; accumulation, biases, losses, gradients of parameters and the optimizer are fp32.
        dropout: probability of the F.dropout in front of every dilated convolution (modules.py:127-128); applied in
        train-mode forwards only, with a counter-based mask (seed, call number, layer) that backward regenerates."""
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f"dropout probability has to be between 0 and 1, but got {dropout}")     # F.dropout's own check
        self.dropout, self.drop_seed, self.drop_calls = float(dropout), int(drop_seed), 0
        if not torch.cuda.is_available():
            raise L.WaeError("WaeEngine needs a ROCm GPU: the hot path has no CPU implementation")
        self.lib = L.lib()
        from .options import EngineOptions
        self.opt = EngineOptions.from_env()       # every launch-path switch, read once (options.py)
        # GLU_PAIR (barrier on every second weight chunk of the fused layer kernel; bit-identical): measured -2.3 % per launch without
        # the z save, nothing with it -- so inference launches take it by default
        self.glu_flags = L.GLU_PAIR if self.opt.glu_pair == "1" else 0
        self.glu_pair = self.opt.glu_pair
        self.g = geom
        self.dt = _dt(dtype)
        self.tdtype = {L.WAE_BF16: torch.bfloat16, L.WAE_F16: torch.float16, L.WAE_F32: torch.float32}[self.dt]
        # fp16 has 5 exponent bits: the backward pass runs on gradients multiplied by grad_scale (a power of two: exact), the
        # weight-gradient contractions carry 1/grad_scale in their alpha, so everything fp32 sees is unscaled.  Bounds: the
        # largest 16-bit gradient is |dy| <= grad_scale / count <= 4096 << 65504; at C5 (count 8e4) dy ~ 2e-4 stays normal.
        self.grad_scale = 4096.0 if self.dt == L.WAE_F16 else 1.0
        self.device = torch.device(device)
        self.lay = P.ParamLayout(geom)
        dev = self.device
        self.params = torch.zeros(self.lay.total, dtype=torch.float32, device=dev)
        self.eff = torch.zeros_like(self.params)
        self.wn_v = torch.from_numpy(self.lay.wn_v_off).to(dev)
        self.wn_g = torch.from_numpy(self.lay.wn_g_off).to(dev)
        self.wn_c = torch.from_numpy(self.lay.wn_cols).to(dev)
        g = geom
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        self.m_w1 = up(P.glu_w1_map(g, self.lay, self.dt))
        self.m_w2 = up(P.glu_w2_map(g, self.lay, self.dt))
        self.m_b2 = up(P.glu_bias2_map(g, self.lay))
        tab, fb = P.first_conv_maps(g, self.lay)
        self.m_tab, self.m_fb = up(tab), up(fb)
        # head: register-chained kernels up to 256 skip channels; wider heads as separate GEMM launches (csrc/gemm_tm.hip)
        self.wide_head = P.head_is_wide(g) or self.opt.head_wide
        # 16-bit dtypes: GEMM 0 of the head as its own wae_gemm_tm launch (decoder_forward); WAE_HEAD_SPLIT=0 keeps the one-kernel head
        self.split_head = not self.wide_head and self.dt in (L.WAE_BF16, L.WAE_F16) and g.Sp in (128, 256) and self.opt.head_split
        if self.wide_head:
            hm = P.head_wide_maps(g, self.lay, self.dt)
            self.m_hwide = {k: up(v) for k, v in hm.items()}
            self.w_hwide = {k: torch.zeros(v.numel(), dtype=self.tdtype, device=dev) for k, v in self.m_hwide.items()}
            self.m_hw = torch.zeros(0, dtype=torch.int32, device=dev)
        else:
            self.m_hw = up(P.head_w_map(g, self.lay, self.dt))
        self.m_hb = up(P.head_bias_map(g, self.lay))
        self.n_w1 = self.m_w1.numel()
        self.n_w2 = self.m_w2.numel()
        self.glu_elems = self.n_w1 + self.n_w2
        assert self.glu_elems == P.glu_packed_elems(g, self.dt)
        self.w_glu = torch.zeros(g.layers * self.glu_elems, dtype=self.tdtype, device=dev)
        self.b_glu = torch.zeros(g.layers * g.Rp, dtype=torch.float32, device=dev)
        self.first_tab = torch.zeros(self.m_tab.numel(), dtype=torch.float32, device=dev)
        self.first_bias = torch.zeros(g.Rp, dtype=torch.float32, device=dev)
        self.w_head = torch.zeros(self.m_hw.numel(), dtype=self.tdtype, device=dev)
        self.b_head = torch.zeros(2 * g.Sp + g.Op, dtype=torch.float32, device=dev)   # [sum skip bias | b1 | b3]
        self._ws: Dict[tuple, dict] = {}
        # which form of the cooperative decode kernel runs (ar_path(); arguments of the C ABI, not environment variables)
        self.ar_generic, self.ar_one_handover, self.ar_resident = False, False, (0, 0)
        self._param_gen, self._prep_gen, self._ar_gen = 1, 0, -1
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)      # sticky WAE_ERR_* bits set by the kernels (include/wae.h)

    # Parameter generations.  Everything derived from the parameters -- the weight-normed arena and the fragment-packed buffers
    # (prepare_weights), the matrix-vector layout of the autoregressive kernels (pack_ar_weights) -- remembers the generation it was
    # made from.  `weights_dirty = True` (train_step, load_state_dict, the drop-in modules' parameter aliases, an EMA swap) starts a new
    # generation; a stale derivative is rebuilt where it is next needed.  (Round 3 kept two independent booleans: the decode weights
    # packed at the first in-training evaluation were reused by every later one.)
    @property
    def weights_dirty(self) -> bool:
        return self._prep_gen != self._param_gen

    @weights_dirty.setter
    def weights_dirty(self, dirty: bool):
        if dirty:
            self._param_gen += 1
        else:
            self._prep_gen = self._param_gen

    @property
    def _ar_packed(self) -> bool:
        return self._ar_gen == self._param_gen

    @_ar_packed.setter
    def _ar_packed(self, ok: bool):
        self._ar_gen = self._param_gen if ok else -1

    def check_errors(self):
        """Turn the kernels' sticky id-range flags into the IndexError the reference raises on the spot (nn.Embedding for a
        speaker id >= n_speakers, the one-hot encoder / CrossEntropyLoss for a class id outside [0, out_channels)).  Reads one
        device word (synchronises): the train script calls it where it already reads the logged scalars."""
        bits = int(self.err.item())
        if bits:
            self.err.zero_()
            if bits & L.ERR_NOT_ONEHOT:
                # wavenet.py:203 runs first_conv as a dense 1x1 on ANY (B, C, T) float tensor; the kernels gather rows of its weight by
                # class id -- identical for one-hot columns only.  Soft labels / probabilities are refused, not arg-maxed.
                raise NotImplementedError("decoder input (B, C, T) is not one-hot (every column exactly one 1.0 among zeros): dense inputs "
                                          "to first_conv are not implemented on the teacher-forced path; pass one-hot columns or (B, T) class ids")
            what = [n for b, n in ((L.ERR_CLASS_ID, f"input class id outside [0, {self.g.O})"),
                                   (L.ERR_SPEAKER_ID, f"speaker id outside [0, {self.g.n_speakers})"),
                                   (L.ERR_TARGET_ID, f"target class id outside [0, {self.g.O})")) if bits & b]
            raise IndexError("index out of range in self: " + "; ".join(what) + " (ids were clamped; results of that call are invalid)")

    def ids_from_onehot(self, x: torch.Tensor) -> torch.Tensor:
        """(B, C, T) one-hot floats (any strides: a transposed (B, T, C) view needs no copy) -> (B, T) int32 class ids.  A column that is
        not one-hot sets WAE_ERR_NOT_ONEHOT in the sticky error word; check_errors() raises (wae_onehot_to_ids, include/wae.h)."""
        x = x.to(self.device, torch.float32)
        B, C, T = x.shape
        ids = torch.empty(B, T, dtype=torch.int32, device=self.device)
        L.check(self.lib.wae_onehot_to_ids(L.ptr(x), B, C, T, x.stride(0), x.stride(1), x.stride(2), L.ptr(ids), L.ptr(self.err),
                                           L.ERR_NOT_ONEHOT, self.stream()), "onehot_to_ids")
        self._onehot_keep = x
        return ids

    # ------------------------------------------------------------------ parameters
    def stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ------------------------------------------------------------------ two chains of layer launches (include/wae.h: wae_stream_delay)
    def chain_plan(self, B: int, T: int, backward: bool = False):
        """None (one chain of full-batch launches), or (b_cut, delay_us): clips [0, b_cut) stay on the current stream, clips
        [b_cut, B) run as a second chain of the same launches on the engine's side stream, started delay_us late.
        Every workgroup of a layer launch stores its outputs at the same time -- at BASELINE C2 ~20 us of a 55-us forward launch and
        ~30 of a 77-us backward launch in which nothing computes -- and a launch with more workgroups than CUs (C5: 320) ends with a
        quarter-full second round.  One launch cannot be de-phased against itself (it ends with its late half: profiles/
        r06_pair8_experiments.txt); two chains of launches can, and two 160-workgroup launches side by side fill C5's machine.
        Measured on whole train steps (profiles/r06_chains.txt): C5 37.5 -> 33.4 ms (both directions), C2 5.26 -> 5.19 ms (the backward
        sweep only: `auto` leaves a forward stack of one round of workgroups on one chain); hps/vqwae.json's 160-workgroup launches gain
        nothing, so `auto` leaves them alone.  Same kernels, same arithmetic per clip: results are bit for bit those of one chain
        (tests/test_gpu_backward.py::test_two_chains_are_bitwise_one_chain)."""
        if not two_chains(self.opt.chains, self.dt in (L.WAE_BF16, L.WAE_F16), B, T, backward):      # (options.py: the rule and its numbers)
            return None
        g = self.g
        delay = getattr(self, "chain_delay_us", None)        # (tools)
        if delay is None:
            # The forward chains start together and stay in step (started half a launch apart they ran 64 us per layer at C2 until they
            # fell into step, 53 in step, 55.5 as one chain).  The backward sweep's second chain starts half a launch late: 35 us at
            # C2, scaled by the MFMA work of a workgroup.  Its launches then run 70-72 us per layer against 77 in step -- for about 14
            # layers, until the trailing chain (~3 us per launch faster) has caught up; holding it back with events on a third stream
            # costs more than it keeps (profiles/r06_chains.txt).
            work = (g.k * g.Rp + g.Ccp) * 2 * g.Hp + g.Hp * g.Rp
            delay = 35.0 * work / 368640.0 if backward else 0.0
        return (B + 1) // 2, float(delay)

    def side_stream(self, k: int = 0):
        ss = self.__dict__.setdefault("_side_streams", {})
        if k not in ss:
            ss[k] = torch.cuda.Stream(self.device)
        return ss[k]

    @contextlib.contextmanager
    def branch(self, k: int, after=None):
        """Independent side work of a step: the body's launches go to side stream k, which first waits for `after` (an event of the
        main stream; default: everything enqueued on the current stream so far).  Yields a one-element list that holds the branch's
        completion event afterwards: `self.join(box[0])` where the main stream needs the results.  (The step's small launches --
        weight packing, the upsampling network, the front end's backward, scatters -- fill a fraction of the machine each and run
        one after the other on one stream: DESIGN 3.4.)"""
        side, cur = self.side_stream(k), torch.cuda.current_stream(self.device)
        if after is None:
            after = torch.cuda.Event()
            after.record(cur)
        side.wait_event(after)
        box = [None]
        with torch.cuda.stream(side):
            yield box
        box[0] = torch.cuda.Event()
        box[0].record(side)

    def join(self, ev):
        if ev is not None:
            torch.cuda.current_stream(self.device).wait_event(ev)

    def chain_fork(self, delay_us: float):
        """The side stream waits for everything enqueued on the current stream so far, then for delay_us; returns it."""
        self._side_stream = self.side_stream(0)
        side, cur = self._side_stream, torch.cuda.current_stream(self.device)
        ev = torch.cuda.Event()
        ev.record(cur)
        side.wait_event(ev)
        L.check(self.lib.wae_stream_delay(delay_us, ctypes.c_void_p(side.cuda_stream)), "stream_delay")
        return side

    def chain_join(self):
        """The current stream waits for the side stream's chain."""
        ev = torch.cuda.Event()
        ev.record(self._side_stream)
        torch.cuda.current_stream(self.device).wait_event(ev)

    def view(self, name: str) -> torch.Tensor:
        off = self.lay.off(name)
        return self.params[off:off + self.lay.numel(name)].view(self.lay.shapes[name])

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        missing = [k for k in self.lay.offsets if k not in sd]
        extra = [k for k in sd if k not in self.lay.offsets]
        if strict and (missing or extra):
            raise KeyError(f"state_dict mismatch: missing {missing[:5]}, unexpected {extra[:5]}")
        host = torch.zeros(self.lay.total, dtype=torch.float32)
        for k in self.lay.offsets:
            if k in sd:
                t = sd[k].detach().to(torch.float32).cpu().reshape(-1)
                if t.numel() != self.lay.numel(k):
                    raise ValueError(f"{k}: expected {self.lay.shapes[k]}, got {tuple(sd[k].shape)}")
                host[self.lay.off(k):self.lay.off(k) + t.numel()] = t
        self.params.copy_(host.to(self.device))
        self.weights_dirty = True

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {k: self.view(k).detach().clone() for k in self.lay.offsets}

    def prepare_weights(self, side: bool = False):
        """K14 weight norm + fragment packing; call after every parameter update.  side (train_step): the packing runs on a side stream
        behind weight norm -- what needs the effective weights only (an encoder, the upsampling network, the hoisted global
        conditioning) goes ahead on the main stream beside it; decoder_forward joins in front of the first launch that reads a packed
        array."""
        lib, st, g, lay = self.lib, self.stream(), self.g, self.lay
        L.check(lib.wae_weight_norm_fwd(L.ptr(self.params), L.ptr(self.eff), lay.total, L.ptr(self.wn_v), L.ptr(self.wn_g),
                                        L.ptr(self.wn_c), len(lay.wn_cols), st), "weight_norm_fwd")
        self._ev_wn = None
        if side and self.opt.side and self.device.type == "cuda":      # the effective weights exist: what only needs them may start
            self._ev_wn = torch.cuda.Event()
            self._ev_wn.record(torch.cuda.current_stream(self.device))
        with (self.branch(1, after=self._ev_wn) if self._ev_wn is not None else contextlib.nullcontext([None])) as self._pack_done:
            self._pack_weights()
        self.weights_dirty = False

    def _pack_weights(self):
        lib, st, g, lay = self.lib, self.stream(), self.g, self.lay
        jobs = getattr(self, "_pack_jobs", None)
        if jobs is None:       # the pointers never change: one host array, one launch for every family
            es = self.w_glu.element_size()
            eff = self.eff.data_ptr()
            J = lambda mp, dst, n, nb, ss, ds, dt: L.GatherJob(eff, mp.data_ptr(), dst, n, ss, ds, nb, dt)
            lst = [J(self.m_w1, self.w_glu.data_ptr(), self.n_w1, g.layers, lay.layer_stride, self.glu_elems, self.dt),
                   J(self.m_w2, self.w_glu.data_ptr() + self.n_w1 * es, self.n_w2, g.layers, lay.layer_stride, self.glu_elems, self.dt),
                   J(self.m_b2, self.b_glu.data_ptr(), g.Rp, g.layers, lay.layer_stride, g.Rp, L.WAE_F32),
                   J(self.m_tab, self.first_tab.data_ptr(), self.m_tab.numel(), 1, 0, 0, L.WAE_F32),
                   J(self.m_fb, self.first_bias.data_ptr(), g.Rp, 1, 0, 0, L.WAE_F32)]
            if self.wide_head:
                lst += [J(self.m_hwide[k], self.w_hwide[k].data_ptr(), self.m_hwide[k].numel(), 1, 0, 0, self.dt)
                        for k in ("skip", "w1", "w3")]
            else:
                lst.append(J(self.m_hw, self.w_head.data_ptr(), self.m_hw.numel(), 1, 0, 0, self.dt))
            lst.append(J(self.m_hb, self.b_head.data_ptr() + g.Sp * 4, g.Sp + g.Op, 1, 0, 0, L.WAE_F32))
            jobs = self._pack_jobs = (L.GatherJob * len(lst))(*lst)
        L.check(lib.wae_pack_gather_multi(jobs, len(jobs), st), "pack weights")
        L.check(lib.wae_sum_rows(L.ptr(self.eff), lay.off("wavenet.conv_layers.0.conv1x1_skip.bias"), lay.layer_stride,
                                 g.layers, g.S, g.Sp, L.ptr(self.b_head), st), "sum skip bias")

    # ------------------------------------------------------------------ workspaces
    def workspace(self, B: int, T: int, train: bool = False) -> dict:
        key = (B, T, train)
        ws = self._ws.get(key)
        if ws is None:
            g, dev, td = self.g, self.device, self.tdtype
            ws = dict(
                x=[torch.empty(B, T, g.Rp, dtype=td, device=dev) for _ in range((g.layers + 1) if train else 2)],
                u=torch.zeros(B, T, g.Ku, dtype=td, device=dev),      # all layers' gated activations (pad columns stay 0)
                zb=torch.empty(B, g.layers, 2 * g.Hp, dtype=torch.float32, device=dev),
                c_up=torch.zeros(B, T, g.Ccp, dtype=td, device=dev) if g.Ccp else None,
                nll=torch.zeros(B, T, dtype=torch.float32, device=dev),
                loss=torch.zeros(2, dtype=torch.float32, device=dev),
            )
            if train:
                ws["z"] = [torch.empty(B, T, 2 * g.Hp, dtype=td, device=dev) for _ in range(g.layers)]
                if self.dropout > 0:       # dropout(x_l): operand of layer l's convolution and of its weight gradient
                    ws["xd"] = [torch.empty(B, T, g.Rp, dtype=td, device=dev) for _ in range(g.layers)]
                ws["lse"] = torch.zeros(B, T, dtype=torch.float32, device=dev)
            if train or self.wide_head or self.split_head:   # the wide / split head pass h0 (and h1) through HBM in inference too
                ws["h0"] = torch.empty(B, T, g.Sp, dtype=td, device=dev)
            if train or self.wide_head:
                ws["h1"] = torch.empty(B, T, g.Sp, dtype=td, device=dev)
            self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------ front end
    def encoder_forward(self, c: torch.Tensor) -> torch.Tensor:
        """a1: c (B, c_in, F) fp32 -> latents (B, Cc, F')  (vqvae_model.py:48-51)."""
        g, lib, st = self.g, self.lib, self.stream()
        x = c.contiguous().float()
        B = x.shape[0]
        acts = [x]
        for i, (k, s) in enumerate(P.ENCODER_BLOCKS):
            wname = f"encoder.net.{i}.conv.weight"
            co, ci, _ = self.lay.shapes[wname]
            Tin = x.shape[-1]
            Tout = (Tin + 2 * (k // 2) - k) // s + 1
            y = torch.empty(B, co, Tout, dtype=torch.float32, device=self.device)
            w = self.eff[self.lay.off(wname):]
            bb = self.eff[self.lay.off(f"encoder.net.{i}.conv.bias"):]
            L.check(lib.wae_enc_conv_fwd(L.ptr(x), L.ptr(w), L.ptr(bb), L.ptr(y), B, ci, Tin, co, k, s, k // 2, 1,
                                         int(s == 1 and ci == co), st), "enc_conv")
            x = y
            acts.append(x)
        Tq = x.shape[-1]
        lat = torch.empty(B, g.Cc, Tq, dtype=torch.float32, device=self.device)
        L.check(lib.wae_enc_conv_fwd(L.ptr(x), L.ptr(self.eff[self.lay.off("encoder.lin.weight"):]),
                                     L.ptr(self.eff[self.lay.off("encoder.lin.bias"):]), L.ptr(lat), B, g.encoder_hid, Tq,
                                     g.Cc, 1, 1, 0, 0, 0, st), "enc_lin")
        self._enc_acts = acts          # inputs of every block (+ the last block's output): backward needs them
        return lat

    def vq_forward(self, lat: torch.Tensor, beta: float = 0.25):
        """a2: -> (quant (B,Cc,Tq), idx int64 (B*Tq), stats[2] = (vq_loss, perplexity))."""
        g = self.g
        B, D, Tq = lat.shape
        idx = torch.empty(B * Tq, dtype=torch.int64, device=self.device)
        quant = torch.empty_like(lat)
        stats = torch.empty(2, dtype=torch.float32, device=self.device)
        hist = torch.empty(g.K + 1, dtype=torch.int32, device=self.device)
        L.check(self.lib.wae_vq_nearest(L.ptr(lat), L.ptr(self.eff[self.lay.off("vq.embedding.weight"):]), L.ptr(idx),
                                        L.ptr(quant), L.ptr(stats), L.ptr(hist), B, D, Tq, g.K, beta, self.stream()),
                "vq_nearest")
        return quant, idx, stats

    def upsample_forward(self, c: torch.Tensor, out: torch.Tensor):
        """a3: c (B,Cc,Tc) fp32 -> out (B, (Tc - 2 cin_pad) * prod(scales), Ccp) time-major compute dtype.  ConvInUpsampleNetwork
        (upsample.py:69-85): conv_in eats cin_pad frames at either end, then the stages; plain UpsampleNetwork (Geometry.conv_in False,
        upsample.py:29-66): the stages on all Tc frames, then cin_pad * prod(scales) samples trimmed at either end."""
        g, lib, st = self.g, self.lib, self.stream()
        B, Cc, Tc = c.shape
        c = c.contiguous()
        n = len(g.upsample_scales)
        trim = 0
        act = P.UP_ACT_KINDS.get(g.up_act, 0)          # upsample_activation behind every stage (upsample.py:44-46), 0 = none
        if g.conv_in:
            kin = 2 * g.cin_pad + 1
            Tin = Tc - 2 * g.cin_pad
            x = torch.empty(B, Cc, Tin, dtype=torch.float32, device=self.device)
            L.check(lib.wae_enc_conv_fwd(L.ptr(c), L.ptr(self.eff[self.lay.off("wavenet.upsample_net.conv_in.weight"):]),
                                         None, L.ptr(x), B, Cc, Tc, Cc, kin, 1, 0, 0, 0, st), "conv_in")
            self._up_acts = [c, x]          # conv_in input, then every stage's input
        else:
            x, Tin = c, Tc
            trim = g.cin_pad * int(np.prod(g.upsample_scales))
            self._up_acts = [None, x]
        for i, s in enumerate(g.upsample_scales):
            w = self.eff[self.lay.off(P.up_stage_name(g, i) + ".weight_v"):]
            last = i == n - 1 and trim == 0 and not act    # the last stage writes the time-major operand itself, unless something follows
            if last:
                assert out.shape[1] == Tin * s, (out.shape, Tin * s)
                y = out
            else:
                y = torch.empty(B, Cc, Tin * s, dtype=torch.float32, device=self.device)
            L.check(lib.wae_upsample_stage_fwd(L.ptr(x), L.ptr(w), L.ptr(y), B, Cc, Tin, s, int(last), g.Ccp, self.dt, st),
                    "upsample_stage")
            x, Tin = y, Tin * s
            if act:
                L.check(lib.wae_act_fwd(L.ptr(x), x.numel(), act, float(g.up_act_slope), st), "upsample activation")
            if i < n - 1:
                self._up_acts.append(x)
        self._up_last = x if act else None      # the last stage's activated output: backward forms act' from it
        if trim or act:       # upsample.py:64-65: c[:, :, indent:-indent]
            assert out.shape[1] == Tin - 2 * trim, (out.shape, Tin, trim)
            xt = x[:, :, trim:Tin - trim].contiguous() if trim else x
            L.check(lib.wae_to_btc(L.ptr(xt), L.ptr(out), B, Cc, Tin - 2 * trim, g.Ccp, self.dt, st), "to_btc (c_up)")
            self._up_keep = xt
        return out

    def layer_drop_seed(self, call: int, layer: int) -> int:
        """64-bit seed of layer `layer`'s dropout mask in the call-th train-mode forward (csrc/misc.hip: dropout_keep)"""
        # distinct (drop_seed, call, layer) triples give distinct seeds; the kernel finalises the seed (dropout_key) before it meets
        # the element index, so neighbouring seeds give unrelated masks.  drop_seed should differ per data-parallel rank and
        # drop_calls follow the global step across a resume (vqwae_train.py does both).
        return ((self.drop_seed * 0x100000001B3 + call) * 1024 + layer) & 0xFFFFFFFFFFFFFFFF

    # ------------------------------------------------------------------ decoder
    def decoder_forward(self, x: torch.Tensor, c: Optional[torch.Tensor], gid: Optional[torch.Tensor],
                        targets: Optional[torch.Tensor] = None, lengths: Optional[torch.Tensor] = None,
                        want_logits: bool = True, train: bool = False, c_is_upsampled: bool = False,
                        gvec: Optional[torch.Tensor] = None, layer_events: Optional[list] = None, dropout_on: bool = True):
        """WaveNet.forward (wavenet.py:164-216) on class ids.

        x: (B,T) int32 class ids (mulaw-quantize) or (B,T) fp32 scalars (scalar_input).
        c: (B,Cc,Tc) fp32 local conditioning (upsampled here) or, if c_is_upsampled, (B,Cc,T).
        gid: (B,) int32 speaker ids.  targets: (B,T) int32 -> fused shifted CE.
        Returns dict(logits (B,O,T) | None, nll (B,T) | None, loss | None).
        """
        if self.weights_dirty:
            self.prepare_weights()
        g, lib, st = self.g, self.lib, self.stream()
        B, T = x.shape
        ws = self.workspace(B, T, train)
        # global conditioning folded with the conv bias (first: in a train step the launches below run beside the weight packing)
        wg_off = self.lay.off("wavenet.conv_layers.0.conv1x1g.weight_v") if g.Cg > 0 else -1
        emb_off = self.lay.offsets.get("wavenet.embed_speakers.weight", 0)
        use_gid = gid is not None and "wavenet.embed_speakers.weight" in self.lay.offsets
        if gid is not None:
            gid = gid.to(torch.int32).contiguous()
        L.check(lib.wae_gproj_fwd(L.ptr(self.eff), wg_off if (gid is not None or gvec is not None) else -1,
                                  self.lay.off("wavenet.conv_layers.0.conv.bias"), self.lay.layer_stride,
                                  L.ptr(gid) if use_gid else None, emb_off, L.ptr(gvec) if gvec is not None else None,
                                  L.ptr(ws["zb"]), B, g.layers, g.G, g.Hp, max(g.Cg, 0), int(g.n_speakers or 0), L.ptr(self.err), self.stream()),
                "gproj")
        # local conditioning
        if g.Ccp:
            if c is None:
                raise ValueError("model has local conditioning but c is None")
            if c_is_upsampled or not g.upsample_scales:
                if c.shape[-1] != T:
                    raise Exception(f"c {tuple(c.shape)} x T={T}")           # wavenet.py:198-200
                L.check(lib.wae_to_btc(L.ptr(c.contiguous().float()), L.ptr(ws["c_up"]), B, g.Cc, T, g.Ccp, self.dt, self.stream()), "to_btc")
            else:
                Tup = (c.shape[-1] - 2 * g.cin_pad) * int(np.prod(g.upsample_scales))
                if Tup != T:
                    raise Exception(f"c {tuple(c.shape)} upsamples to {Tup} != T={T}")  # wavenet.py:198-200
                self.upsample_forward(c.float(), ws["c_up"])
        # (train_step: the packing of the weights runs on a side stream beside the launches above -- prepare_weights(side=True))
        self.join(self.__dict__.pop("_pack_done", [None])[0])
        # first conv
        if g.scalar_input:
            xs = x.contiguous().float()
            L.check(lib.wae_first_conv_fwd(None, L.ptr(xs), L.ptr(self.first_tab), L.ptr(self.first_bias), L.ptr(ws["x"][0]),
                                           B * T, g.Rp, 1, self.dt, None, st), "first_conv")
        else:
            xi = x.to(torch.int32).contiguous()
            L.check(lib.wae_first_conv_fwd(L.ptr(xi), None, L.ptr(self.first_tab), L.ptr(self.first_bias), L.ptr(ws["x"][0]),
                                           B * T, g.Rp, g.O, self.dt, L.ptr(self.err), st), "first_conv")
        # gated residual stack
        es = self.w_glu.element_size()
        # dropout (modules.py:127-128) only in a train-mode forward of a model in training mode; eval is the identity
        if train:
            self.fwd_gen = getattr(self, "fwd_gen", 0) + 1       # the saved activations now belong to THIS forward
        drop = self.dropout if (train and dropout_on) else 0.0
        self._drop_seeds = None
        if drop > 0:
            self.drop_calls += 1
            self._drop_seeds = [self.layer_drop_seed(self.drop_calls, i) for i in range(g.layers)]
        if layer_events is not None:   # HIP events on the launch stream around the whole gated stack (bench.py)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
        # two half-batch chains of the same launches (chain_plan): not with dropout (its mask generator counts elements of the full batch)
        plan = self.chain_plan(B, T) if (drop == 0 and not (train and "xd" in ws)) else None

        parts = [(0, B, st)]
        if plan is not None:
            side = self.chain_fork(plan[1])
            parts = [(0, plan[0], st), (plan[0], B - plan[0], ctypes.c_void_p(side.cuda_stream))]
        for i, dil in enumerate(g.dilations):
            last = i == g.layers - 1
            flags = (L.GLU_SAVE_Z if train else 0) | (L.GLU_NO_OUT if last else 0) | self.glu_flags
            if not train and self.glu_pair == "inference":
                flags |= L.GLU_PAIR
            xin = ws["x"][i if train else i % 2]
            xout = ws["x"][(i + 1) if train else (i + 1) % 2]
            xconv = xin
            if drop > 0:
                xconv = ws["xd"][i]
                L.check(lib.wae_dropout_fwd(L.ptr(xin), L.ptr(xconv), B * T * g.Rp, self._drop_seeds[i], drop, self.dt, st), "dropout")
            elif train and "xd" in ws:
                # an engine built with dropout > 0 running a train-mode forward of a model in eval mode: the weight-gradient
                # tables of this (B, T) point at xd, so it must hold the (undropped) operand
                L.check(lib.wae_dropout_fwd(L.ptr(xin), L.ptr(ws["xd"][i]), B * T * g.Rp, 0, 0.0, self.dt, st), "dropout (identity)")
            for b0, nb, stc in parts:      # every array is clip-major: a chain is the same call on its clips
                dd = L.GluDesc(self.dt, nb, T, g.Rp, g.Ccp, g.Hp, g.k, dil, flags)
                rx = b0 * T * g.Rp * es
                L.check(lib.wae_glu_layer_fwd_drop(ctypes.byref(dd), ctypes.c_void_p(xin.data_ptr() + rx), ctypes.c_void_p(xconv.data_ptr() + rx),
                                                   None if last else ctypes.c_void_p(xout.data_ptr() + rx),
                                                   ctypes.c_void_p(ws["c_up"].data_ptr() + b0 * T * g.Ccp * es) if ws["c_up"] is not None else None,
                                                   ctypes.c_void_p(ws["u"].data_ptr() + (b0 * T * g.Ku + i * g.Hp) * es), g.Ku,
                                                   ctypes.c_void_p(ws["zb"].data_ptr() + (b0 * g.layers + i) * 2 * g.Hp * 4),
                                                   g.layers * 2 * g.Hp,
                                                   ctypes.c_void_p(ws["z"][i].data_ptr() + b0 * T * 2 * g.Hp * es) if train else None,
                                                   ctypes.c_void_p(self.w_glu.data_ptr() + i * self.glu_elems * es),
                                                   ctypes.c_void_p(self.b_glu.data_ptr() + i * g.Rp * 4), stc), f"glu layer {i}")
        if plan is not None:
            self.chain_join()
        if layer_events is not None:
            e1.record(torch.cuda.current_stream(self.device))
            layer_events.append((e0, e1))
        # head (+ fused CE)
        hd = L.HeadDesc(self.dt, B, T, g.Ku, g.Sp, g.Op, g.O, math.sqrt(1.0 / g.layers))
        logits = torch.empty(B, g.O, T, dtype=torch.float32, device=self.device) if want_logits else None
        tg = targets.to(torch.int32).contiguous() if targets is not None else None
        if tg is not None and tg.data_ptr() != (xi.data_ptr() if not g.scalar_input else 0):   # targets = inputs: already checked
            L.check(lib.wae_check_ids(L.ptr(tg), B * T, 0, g.O, L.ptr(self.err), L.ERR_TARGET_ID, st), "check targets")
        if self.wide_head:
            self._head_fwd_wide(ws, B, T, logits, tg, train)
        elif self.split_head:
            # the skip contraction (K = Ku: 72 chunks at C2, its operand 590 MB of u) as a wae_gemm_tm launch -- 197 us against the ~210 us
            # the same loop takes inside the one-workgroup-per-CU head kernel -- then the rest from h0 (45 us)
            from . import backward as BW
            es = self.w_head.element_size()
            BW._tm(self, B, T, g.Sp, 3, math.sqrt(1.0 / g.layers), [(ws["u"].data_ptr(), g.Ku, g.Ku, 0)], self.w_head.data_ptr(),
                   ws["h0"].data_ptr(), g.Sp, self.b_head.data_ptr(), 0)
            ck = 64 if es == 2 else 32
            tail = self.w_head.data_ptr() + (g.Ku // ck) * (g.Sp // 32) * 4096
            L.check(lib.wae_head_fwd_from_h0(ctypes.byref(hd), L.ptr(ws["h0"]), ctypes.c_void_p(tail), L.ptr(self.b_head), L.ptr(logits),
                                             L.ptr(tg), L.ptr(ws["nll"]) if tg is not None else None,
                                             L.ptr(ws["lse"]) if (train and tg is not None) else None,
                                             L.ptr(ws["h1"]) if train else None, st), "head (from h0)")
        else:
            L.check(lib.wae_head_fwd(ctypes.byref(hd), L.ptr(ws["u"]), L.ptr(self.w_head), L.ptr(self.b_head), L.ptr(logits),
                                     L.ptr(tg), L.ptr(ws["nll"]) if tg is not None else None,
                                     L.ptr(ws["lse"]) if (train and tg is not None) else None,
                                     L.ptr(ws["h0"]) if train else None, L.ptr(ws["h1"]) if train else None, st), "head")
        out = dict(logits=logits, nll=None, loss=None)
        if tg is not None:
            ln = lengths.to(self.device, torch.int32).contiguous() if lengths is not None else None
            L.check(lib.wae_masked_mean(L.ptr(ws["nll"]), L.ptr(ln), L.ptr(ws["loss"]), B, T, st), "masked_mean")
            out["nll"] = ws["nll"]
            out["loss"] = ws["loss"][0]
        return out

    def _head_fwd_wide(self, ws, B, T, logits, tg, train):
        """wavenet.py:204-214 (+ the shifted CE of vqwae_train.py:363-379,:764) for skip widths above 256: three launches of
        wae_gemm_tm -- h0 = relu(sqrt(1/L) (sum_l b_skip_l + W_skip u)), h1 = relu(b1 + W1 h0), logits / nll from b3 + W3 h1."""
        from . import backward as BW
        g = self.g
        bh = self.b_head.data_ptr()
        BW._tm(self, B, T, g.Sp, 3, math.sqrt(1.0 / g.layers), [(ws["u"].data_ptr(), g.Ku, g.Ku, 0)],
               self.w_hwide["skip"].data_ptr(), ws["h0"].data_ptr(), g.Sp, bh, 0)
        BW._tm(self, B, T, g.Sp, 3, 1.0, [(ws["h0"].data_ptr(), g.Sp, g.Sp, 0)], self.w_hwide["w1"].data_ptr(),
               ws["h1"].data_ptr(), g.Sp, bh + g.Sp * 4, 0)
        ce = L.TmCe(logits.data_ptr() if logits is not None else None, tg.data_ptr() if tg is not None else None,
                    ws["nll"].data_ptr() if tg is not None else None,
                    ws["lse"].data_ptr() if (train and tg is not None) else None, None, 0.0, g.O)
        BW._tm_ce(self, B, T, g.Op, 5, [(ws["h1"].data_ptr(), g.Sp, g.Sp, 0)], self.w_hwide["w3"].data_ptr(), None, 0,
                  bh + 2 * g.Sp * 4, ce)
        self._ce_keep = (logits, tg)

    # ------------------------------------------------------------------ autoregressive synthesis
    def _prepare_ar(self):
        if getattr(self, "_ar_ready", False):
            return
        g, dev = self.g, self.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        lm, w2_off = P.ar_layer_map(g, self.lay, self.dt)
        self.m_ar_layer, self.ar_w2_off, self.ar_layer_elems = up(lm), int(w2_off), int(lm.size)
        self.m_ar_b2 = up(P.ar_bias2_map(g, self.lay))
        self.m_ar_head = up(P.ar_head_map(g, self.lay, self.dt))
        self.m_ar_hb = up(P.ar_head_bias_map(g, self.lay))
        self.ar_w = torch.zeros(g.layers * self.ar_layer_elems, dtype=self.tdtype, device=dev)
        self.ar_b2 = torch.zeros(g.layers * (g.R + g.S), dtype=torch.float32, device=dev)
        self.ar_wh = torch.zeros(self.m_ar_head.numel(), dtype=self.tdtype, device=dev)
        self.ar_hb = torch.zeros(g.S + g.O, dtype=torch.float32, device=dev)
        roff = P.ar_ring_offsets(g)
        self.ar_ring_off, self.ar_ring_total = up(roff), int(roff[-1])
        self.ar_dil = torch.tensor(g.dilations, dtype=torch.int32, device=dev)
        self._ar_ready = True
        self._ar_packed = False

    def pack_ar_weights(self):
        """Effective (weight-normed) weights -> the matrix-vector layout of ar_fwd.hip; this is the engine's
        make_generation_fast_() (wavenet.py:358-364): weight norm is folded once, not per sample."""
        self._prepare_ar()
        if self.weights_dirty:
            self.prepare_weights()
        lib, st, g, lay = self.lib, self.stream(), self.g, self.lay
        L.check(lib.wae_pack_gather(L.ptr(self.eff), L.ptr(self.m_ar_layer), L.ptr(self.ar_w), self.ar_layer_elems, g.layers,
                                    lay.layer_stride, self.ar_layer_elems, self.dt, st), "pack AR layers")
        L.check(lib.wae_pack_gather(L.ptr(self.eff), L.ptr(self.m_ar_b2), L.ptr(self.ar_b2), g.R + g.S, g.layers,
                                    lay.layer_stride, g.R + g.S, L.WAE_F32, st), "pack AR bias2")
        L.check(lib.wae_pack_gather(L.ptr(self.eff), L.ptr(self.m_ar_head), L.ptr(self.ar_wh), self.m_ar_head.numel(), 1, 0, 0,
                                    self.dt, st), "pack AR head")
        L.check(lib.wae_pack_gather(L.ptr(self.eff), L.ptr(self.m_ar_hb), L.ptr(self.ar_hb), g.S + g.O, 1, 0, 0, L.WAE_F32, st),
                "pack AR head bias")
        self.ar_wm = self._pack_ar_fused()
        self._ar_packed = True

    def ar_path(self, generic: bool = False, one_handover: bool = False, lds_layers: Optional[int] = None,
                reg_layers: Optional[int] = None):
        """Selects the form of the cooperative decode kernel (csrc/ar_coop.hip) for this engine's next incremental_forward calls -- the
        A/B and test handles that rounds 4-5 read from the environment inside the library: `generic` = the any-shape kernel on the
        reference's geometry too; `one_handover` = one exchange per layer on host-formed W1_cur . W_out products (measured slower:
        23.9 against 31.4 kHz; wae_ar_generate_coop_fused); lds_layers / reg_layers = how many layers' weight packets stay in LDS / in
        registers (None: as many as fit, 0: none -- (0, 0) is the streaming form; results are bitwise the same for every split)."""
        enc = lambda v: 0 if v is None else (-1 if int(v) == 0 else int(v))  # noqa: E731  (wae_ar_desc: 0 = default, < 0 = none)
        if bool(one_handover) != self.ar_one_handover:
            self._ar_packed = False                    # the products are formed by pack_ar_weights
        self.ar_generic, self.ar_one_handover, self.ar_resident = bool(generic), bool(one_handover), (enc(lds_layers), enc(reg_layers))
        return self

    def _pack_ar_fused(self):
        """include/wae.h: wae_ar_generate_coop_fused.  M_l = sqrt(.5) W1_cur[l] W_out[l-1] for the reference's own geometry (the one the
        cooperative kernel has with its sizes as constants), from the packed decode weights: one small matrix product per layer, once per
        weight update -- like make_generation_fast_ (wavenet.py:358-364) it is preparation, not the decode path.  Opt-in (ar_path(one_handover=True)):
        with the exchange's stores no longer waiting for their acknowledgement a hand-over costs ~1.3 k clocks, and the fused layer's
        longer window (its requests miss L2: 11.7 MB of weights cycle through a 4-MB L2 every sample) measures 19 kHz against 22.7."""
        g = self.g
        if not (g.R == 256 and g.S == 256 and g.O == 256 and g.G == 256 and g.k == 3 and g.layers >= 2 and not g.scalar_input
                and self.ar_one_handover):
            return None
        epl = 4 if self.dt == L.WAE_F32 else 8
        K1 = 3 * g.R + max(g.Cc, 0)
        nkb1, nkbh = (K1 + epl - 1) // epl, (g.H + epl - 1) // epl
        g_pad, w_pad = (g.G + 63) // 64 * 64, (g.R + g.S + 63) // 64 * 64
        w = self.ar_w.view(g.layers, self.ar_layer_elems).float()
        w1 = w[:, :nkb1 * g_pad * epl].view(g.layers, nkb1, g_pad, epl)                       # [l][k / epl][row][k % epl]
        w1c = w1[:, 2 * g.R // epl:3 * g.R // epl, :g.G].permute(0, 2, 1, 3).reshape(g.layers, g.G, g.R)      # current tap
        w2 = w[:, self.ar_w2_off:self.ar_w2_off + nkbh * w_pad * epl].view(g.layers, nkbh, w_pad, epl)
        wout = w2[:, :, :g.R].permute(0, 2, 1, 3).reshape(g.layers, g.R, nkbh * epl)[:, :, :g.H]
        wm = torch.zeros(g.layers, g.G, g.H, dtype=torch.float32, device=self.device)
        a, b = w1c[1:].contiguous(), wout[:-1].contiguous()        # (L-1, G, R), (L-1, R, H)
        L.check(self.lib.wae_bmm_f32(L.ptr(a), L.ptr(b), L.ptr(wm[1:]), g.layers - 1, g.G, g.R, g.H, g.R, g.H, g.H, g.G * g.R, g.R * g.H,
                                     g.G * g.H, math.sqrt(0.5), self.stream()), "bmm_f32")
        self._fused_keep = (a, b)
        return wm.to(self.tdtype).contiguous()

    def incremental_forward(self, c: Optional[torch.Tensor], gid: Optional[torch.Tensor], T: int, mode: str = "sample",
                            test_inputs: Optional[torch.Tensor] = None, uniforms: Optional[torch.Tensor] = None,
                            init_idx: int = 127, c_is_upsampled: bool = False, want_logits: bool = False,
                            gvec: Optional[torch.Tensor] = None, u_mix: Optional[torch.Tensor] = None,
                            u_log: Optional[torch.Tensor] = None, log_scale_min: float = -7.0, clamp_log_scale: bool = False,
                            n_forced: Optional[int] = None):
        """WaveNet.incremental_forward (wavenet.py:218-346) as one persistent launch.

        mode "logits": teacher-forced on test_inputs (B,T) class ids (softmax=False, quantize=False) -> logits (B,O,T);
        "argmax": greedy feedback; "sample": categorical draw from `uniforms` (B,T) in [0,1) (torch.rand if None).
        "probs" / "raw" (quantize=False, wavenet.py:335-338 skipped): the softmax probabilities / raw logits of a step are the
        dense input of the next; they come back as `logits` (B,O,T).
        n_forced: test_inputs covers only the first n_forced steps (default: its length), later steps run free in `mode`
        (wavenet.py:300-305).  init_idx: the class fed to step 0 when nothing is forced -- an int for every utterance (wavenet.py:288:
        127) or one id per utterance (wavenet.py:283-297 starts each batch item from its own row of initial_input); ids per utterance
        run as a forced first step.  Returns dict(idx (B,T) int32, logits (B,O,T) | None).
        Scalar-input decoders: test_inputs (B,T) fp32 teacher-forces the inputs (mode "logits" -> the mixture parameters
        (B,3M,T) as `logits`); mode "sample" draws every step from the mixture of logistics on the uniforms u_mix (B,T,M),
        u_log (B,T) (torch.rand in (1e-5, 1-1e-5) if None) -> dict(x (B,T) fp32, logits | None)."""
        g, lib = self.g, self.lib
        if not getattr(self, "_ar_packed", False) or self.weights_dirty:
            self.pack_ar_weights()
        st = self.stream()
        B = c.shape[0] if c is not None else (test_inputs.shape[0] if test_inputs is not None else 1)
        m = {"logits": 0, "argmax": 1, "sample": 2, "probs": 3, "raw": 4}[mode]
        dev = self.device
        if not isinstance(init_idx, int) and not g.scalar_input:
            ii = torch.as_tensor(init_idx).reshape(-1).to("cpu", torch.int64)
            if ii.numel() == 1:
                init_idx = int(ii[0])
            else:
                if ii.numel() != B:
                    raise ValueError(f"init_idx: {ii.numel()} start classes for {B} utterances")
                if int(ii.min()) < 0 or int(ii.max()) >= g.O:
                    raise IndexError(f"index {int(ii.max() if ii.max() >= g.O else ii.min())} is out of bounds for dimension 2 with size {g.O}")
                if test_inputs is None:     # (forced steps override the start class anyway: wavenet.py:300-302)
                    test_inputs, n_forced = ii.to(dev, torch.int32).reshape(B, 1), 1
                init_idx = int(ii[0])
        if test_inputs is not None:
            nf = int(test_inputs.shape[1]) if n_forced is None else int(n_forced)
            nf = max(0, min(nf, int(test_inputs.shape[1]), T))
            if nf < T:                      # the kernels index inputs as (B, T)
                pad = torch.zeros(test_inputs.shape[0], T, dtype=test_inputs.dtype, device=test_inputs.device)
                pad[:, :nf] = test_inputs[:, :nf]
                test_inputs = pad
            if nf == 0:
                test_inputs = None
        else:
            nf = 0
        if m == 0 and nf < T:
            raise ValueError("mode 'logits' is teacher-forced: test_inputs must cover all T steps (use 'raw' to feed logits back)")
        c_up = None
        if g.Ccp:
            c_up = torch.zeros(B, T, g.Ccp, dtype=self.tdtype, device=dev)
            if c_is_upsampled or not g.upsample_scales:
                assert c.shape[-1] == T, f"c {tuple(c.shape)} != T {T}"       # wavenet.py:278
                L.check(lib.wae_to_btc(L.ptr(c.contiguous().float()), L.ptr(c_up), B, g.Cc, T, g.Ccp, self.dt, st), "to_btc")
            else:
                assert (c.shape[-1] - 2 * g.cin_pad) * int(np.prod(g.upsample_scales)) == T, "c does not upsample to T"
                self.upsample_forward(c.float(), c_up)
        zb = torch.empty(B, g.layers, 2 * g.Hp, dtype=torch.float32, device=dev)
        wg_off = self.lay.off("wavenet.conv_layers.0.conv1x1g.weight_v") if g.Cg > 0 else -1
        emb_off = self.lay.offsets.get("wavenet.embed_speakers.weight", 0)
        use_gid = gid is not None and "wavenet.embed_speakers.weight" in self.lay.offsets
        gid32 = gid.to(torch.int32).contiguous() if gid is not None else None
        L.check(lib.wae_gproj_fwd(L.ptr(self.eff), wg_off if (gid is not None or gvec is not None) else -1,
                                  self.lay.off("wavenet.conv_layers.0.conv.bias"), self.lay.layer_stride,
                                  L.ptr(gid32) if use_gid else None, emb_off, L.ptr(gvec) if gvec is not None else None,
                                  L.ptr(zb), B, g.layers, g.G, g.Hp, max(g.Cg, 0), int(g.n_speakers or 0), L.ptr(self.err), st),
                "gproj")
        # one utterance per XCD, its gate rows split over up to 32 CUs (csrc/ar_coop.hip); bigger batches run one
        # utterance per CU (csrc/ar_fwd.hip): better aggregate throughput, 3-4x lower speed per utterance
        coop = (B <= 8 and g.R <= 256 and g.S <= 256 and g.O <= 256 and not g.scalar_input and m <= 2
                and self.opt.ar_coop)
        C = max(1, min(self.opt.ar_coop_c, 32, g.H, g.S)) if coop else 1
        # zeros: the rows read as history before their first write (t - d, t - 2d of the first samples) are the causal pad; the
        # cooperative kernel zero-fills its ring itself, but only when its members share an XCD (round-5 advisor finding)
        ring = torch.zeros(B * C * self.ar_ring_total, dtype=torch.float32, device=dev)
        if g.scalar_input:
            es = self.ar_w.element_size()
            M = g.O // 3
            tf = test_inputs.to(dev, torch.float32).contiguous() if test_inputs is not None else None
            if m == 2 and u_mix is None:
                u_mix = torch.rand(B, T, M, device=dev) * (1 - 2e-5) + 1e-5          # mixture.py:138,151
                u_log = torch.rand(B, T, device=dev) * (1 - 2e-5) + 1e-5
            um = u_mix.to(dev, torch.float32).contiguous() if u_mix is not None else None
            ul = u_log.to(dev, torch.float32).contiguous() if u_log is not None else None
            if m == 0 and tf is None:
                raise ValueError("mode 'logits' needs test_inputs")
            xs = torch.empty(B, T, dtype=torch.float32, device=dev) if um is not None else None
            params = torch.empty(B, g.O, T, dtype=torch.float32, device=dev) if (want_logits or m == 0) else None
            if m >= 3:
                raise ValueError("scalar-input decoders feed the drawn sample back: modes 'logits' and 'sample' only")
            d = L.ArDesc(self.dt, B, T, g.layers, g.R, g.Rp, g.G, g.Hp, g.S, g.O, max(g.Cc, 0), g.Ccp, g.k, m, 0, 1,
                         math.sqrt(1.0 / g.layers), nf)
            L.check(lib.wae_ar_generate_scalar(ctypes.byref(d), L.ptr(self.ar_dil), L.ptr(self.ar_ring_off), L.ptr(ring),
                                               self.ar_ring_total, L.ptr(self.ar_w), self.ar_layer_elems * es, self.ar_w2_off * es,
                                               L.ptr(self.ar_b2), L.ptr(zb), L.ptr(self.first_tab), L.ptr(self.first_bias),
                                               L.ptr(self.ar_wh), L.ptr(self.ar_hb), L.ptr(c_up), self.dt,
                                               L.ptr(tf), L.ptr(um), L.ptr(ul), float(log_scale_min),
                                               int(bool(clamp_log_scale)), L.ptr(xs), L.ptr(params), st), "ar_generate_scalar")
            self._ar_keep = (c_up, zb, ring, tf, um, ul, gid32)
            return dict(x=xs, logits=params)
        inputs = test_inputs.to(torch.int32).contiguous() if test_inputs is not None else None
        if inputs is None and not 0 <= int(init_idx) < g.O:
            # wavenet.py:288 writes a one at class 127 of the start vector: the same IndexError when there are fewer classes
            raise IndexError(f"index {int(init_idx)} is out of bounds for dimension 2 with size {g.O}")
        if m == 2 and uniforms is None:
            uniforms = torch.rand(B, T, device=dev)
        uni = uniforms.float().contiguous() if uniforms is not None else None
        out_idx = torch.empty(B, T, dtype=torch.int32, device=dev)
        logits = torch.empty(B, g.O, T, dtype=torch.float32, device=dev) if (want_logits or m == 0 or m >= 3) else None
        es = self.ar_w.element_size()
        d = L.ArDesc(self.dt, B, T, g.layers, g.R, g.Rp, g.G, g.Hp, g.S, g.O, max(g.Cc, 0), g.Ccp, g.k, m, int(init_idx), 0,
                     math.sqrt(1.0 / g.layers), nf, int(self.ar_generic), self.ar_resident[0], self.ar_resident[1])
        if coop:
            nv = lib.wae_ar_coop_msg_values(ctypes.byref(d), C)
            msg = torch.zeros(B * 2 * C * nv, dtype=torch.int64, device=dev)
            acc = torch.zeros(B * lib.wae_ar_coop_acc_floats(ctypes.byref(d)), dtype=torch.float32, device=dev)
            err = torch.zeros(64, dtype=torch.int32, device=dev)   # [0] = time-out flag; the rest: profile counters of a -DWAE_ARC_PROFILE build
            L.check(lib.wae_ar_generate_coop_fused(ctypes.byref(d), C, L.ptr(self.ar_dil), L.ptr(self.ar_ring_off), L.ptr(ring),
                                                   self.ar_ring_total, L.ptr(self.ar_w), self.ar_layer_elems * es, self.ar_w2_off * es,
                                                   L.ptr(self.ar_b2), L.ptr(zb), L.ptr(self.first_tab), L.ptr(self.first_bias),
                                                   L.ptr(self.ar_wh), L.ptr(self.ar_hb), L.ptr(c_up), self.dt, L.ptr(inputs),
                                                   L.ptr(uni), L.ptr(out_idx), L.ptr(logits), L.ptr(msg), L.ptr(acc), L.ptr(err),
                                                   L.ptr(getattr(self, "ar_wm", None) if self.ar_one_handover else None), st),
                    "ar_generate_coop")
            self._ar_profile = err
            if int(err[0].item()) != 0:  # synchronises: generation is a blocking call for its callers anyway
                raise L.WaeError("ar_generate_coop: an exchange between the cooperating workgroups timed out")
        else:
            L.check(lib.wae_ar_generate(ctypes.byref(d), L.ptr(self.ar_dil), L.ptr(self.ar_ring_off), L.ptr(ring), self.ar_ring_total,
                                        L.ptr(self.ar_w), self.ar_layer_elems * es, self.ar_w2_off * es, L.ptr(self.ar_b2), L.ptr(zb),
                                        L.ptr(self.first_tab), L.ptr(self.first_bias), L.ptr(self.ar_wh), L.ptr(self.ar_hb),
                                        L.ptr(c_up), self.dt, L.ptr(inputs), L.ptr(uni), L.ptr(out_idx), L.ptr(logits), st),
                    "ar_generate")
        self._ar_keep = (c_up, zb, ring, inputs, uni, gid32)   # keep device buffers alive until the stream has run
        return dict(idx=out_idx, logits=logits)

    # ------------------------------------------------------------------ full autoencoder
    def forward(self, x: torch.Tensor, c: torch.Tensor, gid: Optional[torch.Tensor], targets=None, lengths=None,
                want_logits=True, train=False, beta: float = 0.25, dropout_on: bool = True, layer_events: Optional[list] = None):
        """VQVAE.forward (vqvae_model.py:66-72) -> dict(logits, vq_loss, perp, latents, idx, quant, loss)."""
        if self.weights_dirty:
            self.prepare_weights()
        lat = self.encoder_forward(c)
        quant, idx, stats = self.vq_forward(lat, beta)
        out = self.decoder_forward(x, quant, gid, targets, lengths, want_logits, train, dropout_on=dropout_on, layer_events=layer_events)
        out.update(latents=lat, quant=quant, idx=idx, vq_loss=stats[0], perp=stats[1])
        self._fe = dict(lat=lat, quant=quant, idx=idx, beta=beta)
        return out

    # ------------------------------------------------------------------ training step (vqwae_train.py:709-798)
    def init_optimizer(self, ema: bool = True):
        self.exp_avg = torch.zeros_like(self.params)
        self.exp_avg_sq = torch.zeros_like(self.params)
        self.shadow = self.params.clone() if ema else None        # ExponentialMovingAverage.register (:343-344)
        self.opt_scratch = torch.zeros(1, dtype=torch.float64, device=self.device)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=self.device)
        self.opt_step = 0

    def backward(self, x, gid, targets, lengths, gvec=None, loss_scale: float = 1.0, ext_dy=None, vq_scale: float = None,
                 grad_sync=None):
        """Gradients of (masked CE or, with ext_dy (B,T,Op) = d loss / d logits, any output loss [+ vq_loss]) of the last
        train-mode forward -> self.grads (flat arena).  loss_scale multiplies the CE term, vq_scale (default: loss_scale) the
        vq_loss term; grad_sync: see backward.decoder_backward."""
        from . import backward as BW
        self._ev_dc = None
        dc = BW.decoder_backward(self, x, targets, lengths, gid, gvec, ext_dy=ext_dy, loss_scale=loss_scale, grad_sync=grad_sync)
        if self.g.Ccp and self.g.upsample_scales:
            if self._ev_dc is not None:
                # dc was complete when the sweep ended (the event): the front end's backward runs on a side stream beside the scatter of
                # the layers' weight gradients that decoder_backward queued behind the sweep (disjoint slices of the gradient arena)
                with self.branch(0, after=self._ev_dc) as tail:
                    BW.frontend_backward(self, dc, loss_scale if vq_scale is None else vq_scale)
                self.join(tail[0])
            else:
                BW.frontend_backward(self, dc, loss_scale if vq_scale is None else vq_scale)
        return BW.finish_grads(self)

    def dmol_loss_and_grad(self, y_hat: torch.Tensor, y: torch.Tensor, lengths, num_classes: int = 65536,
                           log_scale_min: float = -7.0, scale: float = 1.0):
        """DiscretizedMixturelogisticLoss (vqwae_train.py:382-401 with the shift of :766): y_hat (B,3M,T) fp32, y (B,T) fp32
        -> (masked mean loss, scale * d loss / d y_hat as (B,T,Op) in the compute dtype for decoder_backward's ext_dy).
        Per-step loss and its gradient: wae_dmol_loss_fwd; masked mean: wae_masked_mean; mask, 1/count and the transposition
        of the gradient: wae_to_btc_masked."""
        g, lib, st = self.g, self.lib, self.stream()
        B, _, T = y_hat.shape
        yf = y.contiguous().float()
        nll = torch.empty(B, T, dtype=torch.float32, device=self.device)
        dy = torch.empty_like(y_hat)
        L.check(lib.wae_dmol_loss_fwd(L.ptr(y_hat), L.ptr(yf), L.ptr(nll), L.ptr(dy), B, g.O // 3, T, int(num_classes),
                                      float(log_scale_min), 1, st), "dmol_loss")
        if lengths is None:
            count, ln = B * (T - 1), None
        else:
            count = int(torch.clamp(lengths.detach().to("cpu", torch.int64).clamp(max=T) - 1, min=0).sum())
            ln = lengths.to(self.device, torch.int32).contiguous()
        loss = torch.empty(2, dtype=torch.float32, device=self.device)
        L.check(lib.wae_masked_mean(L.ptr(nll), L.ptr(ln), L.ptr(loss), B, T, st), "masked_mean")
        dyt = torch.zeros(B, T, g.Op, dtype=self.tdtype, device=self.device)
        L.check(lib.wae_to_btc_masked(L.ptr(dy), L.ptr(dyt), B, g.O, T, g.Op, self.dt, L.ptr(ln),
                                      float(scale) * self.grad_scale / max(count, 1), st),
                "to_btc dy")
        self._dmol_keep = (yf, nll, dy, ln)
        return loss[0], dyt

    def train_step(self, x, c, gid, lengths=None, lr: float = 4e-4, betas=(0.9, 0.999), eps: float = 1e-8,
                   weight_decay: float = 0.0, clip_thresh: float = 100.0, ema_decay: float = 0.9999, grad_hook=None,
                   quantize_channels: int = 65536, log_scale_min: float = -7.0, grad_sync=None, ce_scale: float = 1.0):
        """One optimisation step: forward (teacher forced, targets = x shifted by one), backward, [data parallel: grad_sync, a
        distributed.GradSync -- its all-reduce starts inside backward and is awaited here; or grad_hook(grads)],
        clip_grad_norm_ + Adam + EMA.  Returns dict(loss, ce, vq_loss, perp, grad_norm).
        ce_scale: factor on the CE gradient (ragged data-parallel shards: n_local * world / n_global, so that the rank mean is
        the gradient of the global masked mean of vqwae_train.py:374-379); vq_loss keeps weight 1 (rank mean, :759).
        Class-id input: masked cross-entropy; scalar input (hparams input_type "raw"): discretized mixture of logistics."""
        if not hasattr(self, "exp_avg"):
            self.init_optimizer()
        self.prepare_weights(side=True)
        # one int32 copy of the ids for forward, targets and backward (each of them converts what it is handed: four 5-us launches)
        if not self.g.scalar_input and x.dtype != torch.int32:
            x = x.to(self.device, torch.int32).contiguous()
        if gid is not None and gid.dtype != torch.int32:
            # (the same speaker-id tensor step after step -- a data loader's batch, a benchmark's fixture -- is converted once)
            key = (gid.data_ptr(), gid._version, tuple(gid.shape), gid.dtype)
            if getattr(self, "_gid32_key", None) != key:
                self._gid32_key, self._gid32 = key, gid.to(self.device, torch.int32).contiguous()
            gid = self._gid32
        from . import backward as BW
        if self._ev_wn is not None:
            # the backward's weight packing (+ the clearing of the step's gradient accumulators) needs the effective weights only:
            # behind the forward's packing on the same side stream, instead of between the head's forward and backward;
            # decoder_backward waits for it.  (Queued when the first conv is enqueued instead -- beside the first layer -- it made that
            # layer's launch 45 us longer: 20 us per step worse.  Two side streams carry everything: a process has few hardware
            # queues, and streams that share one run one after the other -- four chains of layer launches ran at half the speed of two.)
            with self.branch(1, after=self._ev_wn) as self._early_pack:
                BW.pack_bwd_weights(self)
        try:
            if self.g.scalar_input:
                fwd = self.forward if self.g.has_encoder else self.decoder_forward
                out = fwd(x, c, gid, targets=None, lengths=None, want_logits=True, train=True)
                if not self.g.has_encoder:
                    self._fe = None
                loss, dyt = self.dmol_loss_and_grad(out["logits"], x, lengths, quantize_channels, log_scale_min, scale=ce_scale)
                out["loss"] = loss
                grads = self.backward(x, gid, None, lengths, ext_dy=dyt, vq_scale=1.0, grad_sync=grad_sync)
            elif self.g.has_encoder:
                out = self.forward(x, c, gid, targets=x, lengths=lengths, want_logits=False, train=True,
                                   layer_events=getattr(self, "_layer_events", None))
            else:
                out = self.decoder_forward(x, c, gid, targets=x, lengths=lengths, want_logits=False, train=True,
                                           layer_events=getattr(self, "_layer_events", None))
                self._fe = None
            if not self.g.scalar_input:
                grads = self.backward(x, gid, x, lengths, loss_scale=ce_scale, vq_scale=1.0, grad_sync=grad_sync)
        except BaseException:
            self.__dict__.pop("_early_pack", None)      # (a refused input: nothing of this step may be taken for the next call's)
            self.join(self.__dict__.pop("_pack_done", [None])[0])
            raise
        if grad_sync is not None:
            grad_sync.finish()
        if grad_hook is not None:
            grad_hook(grads)
        self.opt_step += 1
        L.check(self.lib.wae_clip_adam_ema(L.ptr(self.params), L.ptr(grads), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                           L.ptr(self.shadow), self.lay.total, L.ptr(self.opt_scratch), L.ptr(self.grad_norm),
                                           self.opt_step, lr, betas[0], betas[1], eps, weight_decay, clip_thresh, ema_decay,
                                           self.stream()), "clip_adam_ema")
        self.weights_dirty = True
        res = dict(ce=out["loss"], grad_norm=self.grad_norm[0])
        if self.g.has_encoder:
            res.update(vq_loss=out["vq_loss"], perp=out["perp"], loss=out["loss"] + out["vq_loss"])
        else:
            res["loss"] = out["loss"]
        return res
