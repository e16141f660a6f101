"""Backward pass of the decoder hot path (host orchestration; all arithmetic in libwae_hip.so).

Gradient flow (autograd of wavenet.py:164-216 + vqwae_train.py:758-766), with "hat" = gradient already carrying the
layer's sqrt(.5):
    head_bwd:   dy, dh1, dskip                                                    (csrc/head_bwd.hip)
    layer l = L-1 .. 0:
        du/dz:  dz_l  = gate'(z_l) * (W_out_l^T dxhat_{l+1} + W_skip_l^T dskip)   (wae_gemm_tm, GATE_BWD)
        wgrad:  dW1_l, dWc_l, dzb_l (ones columns), dW_out_l                      (wae_gemm_tn)
        dx:     dxhat_l = sqrt(.5) * (dxhat_{l+1} + sum_tap W1_tap^T dz_l[t+s])   (wae_gemm_tm, RESIDUAL)
    dc = sum_l Wc_l^T dz_l  (one GEMM over all layers),  dW_skip of all layers (one GEMM), head + first-conv weights,
    scatter into the gradient arena, gproj backward, weight-norm backward.
"""
from __future__ import annotations

import ctypes
import math

from typing import Optional

import numpy as np
import torch

from . import _lib as L
from . import packing as P

RS = math.sqrt(0.5)


def _prepare_bwd(eng):
    if getattr(eng, "_bwd_ready", False):
        return
    g, dev, lay = eng.g, eng.device, eng.lay
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    # Which weight-gradient launches this engine uses follows from eng.opt (options.py: read once per engine): the arenas sized below, the
    # job tables of every (B, T) workspace and the scatter lists all follow from it.
    eng.opt_tn_stream = eng.dt in (L.WAE_BF16, L.WAE_F16) and eng.opt.tn_stream
    eng.opt_tn_static = eng.opt.tn_static
    eng.opt_tn_static_head = eng.opt.tn_static_head
    eng.m_bu = up(P.bwd_u_map(g, lay, eng.dt))
    eng.m_bx = up(P.bwd_x_map(g, lay, eng.dt))
    # K_X(l) + K_U(l-1) in one launch (csrc/glu_bwd.hip).  16-bit storage: the round-5 kernel (two workgroups per CU, the residual
    # launch's own weight stream and chunk order) is the DEFAULT where it has an instantiation -- 5.72 -> 5.52 ms per C2 train step once
    # its epilogues stopped reloading spilled addresses (profiles/EXPERIMENT_LOG.md); EngineOptions.bwd_fused "0" keeps the two launches.
    # fp32 keeps the two launches unless asked ("1"): its fused form is the round-1 kernel (118 us against 60 + 50).
    is16 = eng.dt in (L.WAE_BF16, L.WAE_F16)
    sup = eng.lib.wae_glu_bwd_fused_supported16(g.Rp, g.Hp) if is16 else eng.lib.wae_glu_bwd_fused_supported(g.Rp, g.Hp)
    want = eng.opt.bwd_fused == "1" or (eng.opt.bwd_fused == "auto" and is16)
    eng.fused_bwd = bool(sup) and g.Sp % (64 if is16 else 32) == 0 and want
    if eng.fused_bwd:
        eng.m_buo = up(P.bwd_uo_map(g, lay, eng.dt))
        eng.n_buo = eng.m_buo.numel()
        eng.w_buo = torch.zeros(g.layers * eng.n_buo, dtype=eng.tdtype, device=dev)
        if not is16:
            eng.m_bxf = up(P.bwd_x_map(g, lay, eng.dt, interleave=False))     # the fp32 kernel walks the taps one by one
            eng.w_bxf = torch.zeros(g.layers * eng.m_bxf.numel(), dtype=eng.tdtype, device=dev)
    eng.m_bc = up(P.bwd_c_map(g, lay, eng.dt)) if g.Ccp else None
    eng.m_hb_w = up(P.head_bwd_map(g, lay, eng.dt)) if not eng.wide_head else torch.zeros(0, dtype=torch.int32, device=dev)
    eng.n_bu, eng.n_bx = eng.m_bu.numel(), eng.m_bx.numel()
    eng.w_bu = torch.zeros(g.layers * eng.n_bu, dtype=eng.tdtype, device=dev)
    eng.w_bx = torch.zeros(g.layers * eng.n_bx, dtype=eng.tdtype, device=dev)
    eng.w_bc = torch.zeros(eng.m_bc.numel(), dtype=eng.tdtype, device=dev) if g.Ccp else None
    eng.w_hb = torch.zeros(eng.m_hb_w.numel(), dtype=eng.tdtype, device=dev)
    sm = P.grad_scatter_maps(g, lay)
    eng.sm = {k: (up(v) if isinstance(v, np.ndarray) else v) for k, v in sm.items()}
    eng.d_eff = torch.zeros_like(eng.params)
    eng.grads = torch.zeros_like(eng.params)
    # dense weight-gradient tiles (fp32), one allocation so a single memset clears them
    sizes = dict(c1=g.layers * 2 * g.Hp * sm["ld1"], co=g.layers * g.Rp * sm["ldo"], cs=g.Sp * sm["lds"],
                 c3=g.Op * sm["ldh"], c1h=g.Sp * sm["ldh"], ctab=P._ru(g.O, 128) * g.Rp, fb=g.Rp)
    if static_tn_geometry(eng):
        # wae_gemm_tn_static (kind OUTSKIP) writes dW_out / dW_skip of a layer transposed: rows = gated channel, and the out bias
        # as one more row; the skip bias (column sums of dS, the same for every layer) is the Cb of the LAST layer's OUTSKIP job, whose
        # first operand is dS (its conv1x1_out gradient is dead): a plain (Sp,) vector in the first floats of `cbs`
        sizes.update(coT=g.layers * sm["ldoT_rows"] * g.Rp, csT=g.layers * g.Hp * g.Sp, cbs=g.Sp * P.ONES_PAD)
    total = sum(sizes.values())
    eng.cbuf = torch.zeros(total, dtype=torch.float32, device=dev)
    eng.cview, off = {}, 0
    for k, n in sizes.items():
        eng.cview[k] = eng.cbuf[off:off + n]
        off += n
    eng._bwd_ready = True


def pack_bwd_weights(eng):
    _prepare_bwd(eng)
    lib, st, g, lay = eng.lib, eng.stream(), eng.g, eng.lay
    jobs = getattr(eng, "_pack_bwd_jobs", None)
    if jobs is None:
        eff = eng.eff.data_ptr()
        J = lambda mp, dst, n, nb, ss, ds: L.GatherJob(eff, mp.data_ptr(), dst.data_ptr(), n, ss, ds, nb, eng.dt)
        lst = [J(eng.m_bu, eng.w_bu, eng.n_bu, g.layers, lay.layer_stride, eng.n_bu),
               J(eng.m_bx, eng.w_bx, eng.n_bx, g.layers, lay.layer_stride, eng.n_bx)]
        if eng.fused_bwd:
            lst.append(J(eng.m_buo, eng.w_buo, eng.n_buo, g.layers, lay.layer_stride, eng.n_buo))
            if hasattr(eng, "m_bxf"):
                lst.append(J(eng.m_bxf, eng.w_bxf, eng.n_bx, g.layers, lay.layer_stride, eng.n_bx))       # tap by tap (fp32)
        if g.Ccp:
            lst.append(J(eng.m_bc, eng.w_bc, eng.m_bc.numel(), 1, 0, 0))
        if eng.wide_head:
            lst += [J(eng.m_hwide[k], eng.w_hwide[k], eng.m_hwide[k].numel(), 1, 0, 0) for k in ("w3t", "w1t")]
        else:
            lst.append(J(eng.m_hb_w, eng.w_hb, eng.m_hb_w.numel(), 1, 0, 0))
        # ... and the step's two gradient accumulators cleared by the same launch (fill jobs: no map; round 5 cleared them with two
        # torch fills of 43 + 50 MB between launches, 19 us that this launch's idle store bandwidth absorbs)
        lst += [L.GatherJob(None, None, t_.data_ptr(), t_.numel(), 0, 0, 1, L.WAE_F32) for t_ in (eng.d_eff, eng.cbuf)]
        jobs = eng._pack_bwd_jobs = (L.GatherJob * len(lst))(*lst)
    L.check(lib.wae_pack_gather_multi(jobs, len(jobs), st), "pack backward weights + clear the gradient accumulators")


def _arr(ctype, vals):
    return (ctype * len(vals))(*vals)


def _tm(eng, B, T, M, mode, alpha, srcs, w_ptr, out_ptr, out_stride, aux_ptr=None, aux_stride=0, flags=0, st=None):
    """srcs: list of (device ptr int, row stride elems, cols, shift)"""
    d = L.TmDesc(eng.dt, B, T, M, len(srcs), mode, alpha, flags)
    ptrs = _arr(ctypes.c_void_p, [s[0] for s in srcs])
    strides = _arr(ctypes.c_int64, [s[1] for s in srcs])
    cols = _arr(ctypes.c_int32, [s[2] for s in srcs])
    shifts = _arr(ctypes.c_int32, [s[3] for s in srcs])
    L.check(eng.lib.wae_gemm_tm(ctypes.byref(d), ptrs, strides, cols, shifts, ctypes.c_void_p(w_ptr), ctypes.c_void_p(out_ptr),
                                out_stride, ctypes.c_void_p(aux_ptr) if aux_ptr else None, aux_stride, st if st is not None else eng.stream()), "gemm_tm")


def _tm_ce(eng, B, T, M, mode, srcs, w_ptr, out_ptr, out_stride, bias_ptr, ce):
    """wae_gemm_tm_ce (modes 5 / 6 of the wide head): ce is an L.TmCe"""
    d = L.TmDesc(eng.dt, B, T, M, len(srcs), mode, 1.0, 0)
    ptrs = _arr(ctypes.c_void_p, [s[0] for s in srcs])
    strides = _arr(ctypes.c_int64, [s[1] for s in srcs])
    cols = _arr(ctypes.c_int32, [s[2] for s in srcs])
    shifts = _arr(ctypes.c_int32, [s[3] for s in srcs])
    L.check(eng.lib.wae_gemm_tm_ce(ctypes.byref(d), ptrs, strides, cols, shifts, ctypes.c_void_p(w_ptr),
                                   ctypes.c_void_p(out_ptr) if out_ptr else None, out_stride, ctypes.c_void_p(bias_ptr),
                                   ctypes.byref(ce), eng.stream()), "gemm_tm_ce")


class TileTable:
    """Host builder of a device array of wae_tn_tile (include/wae.h): each add() is one contraction
    C[M][N (+ones)] += alpha * P^T Q, cut into 128x128 output tiles."""

    def __init__(self, eng):
        self.eng, self.tiles = eng, []

    def add(self, M, N, shift, ones_col, alpha, p_ptr, p_stride, q_ptr, q_stride, c_ptr, ldc, onehot_ptr=0):
        es = self.eng.w_glu.element_size()
        nmax = max(N, ones_col + 1) if ones_col >= 0 else N
        TN = 128                                     # csrc/gemm_tn.hip: 128 x 128 output tiles
        for mt in range((M + 127) // 128):
            for nt in range((nmax + TN - 1) // TN):
                oc = ones_col - TN * nt if (ones_col >= 0 and 0 <= ones_col - TN * nt < TN) else -1
                self.tiles.append(L.TnTile(
                    (p_ptr + mt * 128 * es) if p_ptr else None, q_ptr + nt * TN * es, onehot_ptr or None,
                    c_ptr + (mt * 128 * ldc + nt * TN) * 4, p_stride, q_stride, ldc,
                    min(128, M - 128 * mt), max(0, min(TN, N - TN * nt)), 128 * mt, shift, oc, alpha))

    def finalize(self, B):
        arr = (L.TnTile * len(self.tiles))(*self.tiles)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.dev = host.to(self.eng.device)
        self.n = len(self.tiles)
        # measured on C2 (tools/ablate_tn.py): ~1200 workgroups (2-3 per slot) beat fewer, longer ones; keep >= 512 steps each
        self.splits = max(1, min(16, round(1200 / max(1, self.n * B))))
        return self

    def launch(self, B, T):
        eng = self.eng
        L.check(eng.lib.wae_gemm_tn_tiles(eng.dt, L.ptr(self.dev), self.n, B, T, self.splits, eng.stream()), "gemm_tn_tiles")


class StreamTable:
    """Host builder for wae_gemm_tn_stream (csrc/gemm_tn_stream.hip): the weight-gradient contractions of every layer as
    ONE launch.  A *group* is the list of jobs of one layer (every dilated-conv tap, the conditioning 1x1 with the per-clip
    zb sums, conv1x1_out with its bias), each cut into 384 x 256 output regions; all groups have the same number of jobs.
    The (group, 32-row time slab) list is cut into equal contiguous shares, one per *team* of len(group) workgroups."""
    RM, RN, KT = 384, 256, 32

    def __init__(self, eng, B, T):
        self.eng, self.B, self.T = eng, B, T
        self.groups = []          # list of lists of L.TsJob
        self.lead_jobs = 0        # the first lead_jobs jobs of every group are full-size (the taps): the pacing reference
        self.shifts = []          # most negative shift per group (slabs that pair only with rows before the clip are skipped)

    def begin_group(self):
        self.groups.append([])

    def add(self, M, N, shift, ones_col, alpha, p_ptr, p_stride, q_ptr, q_stride, c_ptr, ldc):
        """C[M][N (+ B ones columns at ones_col)] += alpha * P^T Q ; p_ptr == 0: placeholder jobs that do nothing."""
        es = 2
        nmax = max(N, ones_col + self.B) if ones_col >= 0 else N
        for mt in range((M + self.RM - 1) // self.RM):
            for nt in range((nmax + self.RN - 1) // self.RN):
                nv = max(0, min(self.RN, N - self.RN * nt))
                oc = -1
                if ones_col >= 0 and 0 <= ones_col - self.RN * nt < self.RN:
                    oc = ones_col - self.RN * nt
                    assert oc + self.B <= self.RN and nv % 8 == 0 and oc >= nv, "ones columns must fit behind the region's data"
                if nv == 0 and oc < 0:
                    continue
                mv = min(self.RM, M - self.RM * mt) if p_ptr else 0
                self.groups[-1].append(L.TsJob((p_ptr + mt * self.RM * es) if p_ptr else None, q_ptr + nt * self.RN * es,
                                               c_ptr + (mt * self.RM * ldc + nt * self.RN) * 4, p_stride, q_stride, ldc,
                                               mv, nv, shift, oc, alpha, 0))

    def finalize(self):
        eng, B, T = self.eng, self.B, self.T
        gs = len(self.groups[0])
        assert all(len(g) == gs for g in self.groups), "every layer must contribute the same list of jobs"
        ncu = torch.cuda.get_device_properties(eng.device).multi_processor_count
        spc = (T + self.KT - 1) // self.KT
        segs, team_seg = [], [0]
        self.team_size = gs
        self.nteams = max(1, ncu // gs)
        self.nwg = ncu if ncu % 8 == 0 and ncu >= self.nteams * gs else self.nteams * gs
        # Every member of a team sweeps the same slab range; a slab is skipped by a member when all of its rows pair with
        # rows before the clip (kernel: useful()).  Shares are cut on the raw slab count: the skipped slabs differ by tap.
        total = len(self.groups) * B * spc
        for t in range(self.nteams):
            lo, hi = total * t // self.nteams, total * (t + 1) // self.nteams
            while lo < hi:
                grp = lo // (B * spc)
                end = min(hi, (grp + 1) * B * spc)
                segs.append(L.TsSeg(grp * gs, lo - grp * B * spc, end - grp * B * spc))
                lo = end
            team_seg.append(len(segs))
        jobs = [j for g in self.groups for j in g]
        dev = eng.device
        self.jobs_dev = torch.frombuffer(bytearray(bytes((L.TsJob * len(jobs))(*jobs))), dtype=torch.uint8).to(dev)
        self.segs_dev = torch.frombuffer(bytearray(bytes((L.TsSeg * len(segs))(*segs))), dtype=torch.uint8).to(dev)
        self.team_seg_dev = torch.tensor(team_seg, dtype=torch.int32, device=dev)
        # (the kernel's team pacing and equal-time shares -- rounds 2-3, measured slower: profiles/EXPERIMENT_LOG.md -- stay reachable
        #  through wae_gemm_tn_stream's own arguments; the product launches unpaced teams)
        self.pace = torch.zeros(self.nteams * 8, dtype=torch.int32, device=dev)
        self.window, self.pace_from = 0, self.lead_jobs
        return self

    def launch(self):
        eng = self.eng
        L.check(eng.lib.wae_gemm_tn_stream(eng.dt, L.ptr(self.jobs_dev), L.ptr(self.segs_dev), L.ptr(self.team_seg_dev), self.nteams,
                                           self.team_size, self.nwg, self.B, self.T, L.ptr(self.pace), self.window, self.pace_from, eng.stream()),
                "gemm_tn_stream")


def _cuts(n, cap, gran):
    """[0, n) in the fewest equal parts of at most `cap` columns, each a multiple of `gran` -> [(first column, width)]"""
    parts = -(-n // cap)
    w = -(-(-(-n // parts)) // gran) * gran
    out, c = [], 0
    while c < n:
        out.append((c, min(w, n - c)))
        c += w
    return out


def static_tn_geometry(eng):
    """The static-schedule weight-gradient launch (csrc/gemm_tn_static.hip): 16-bit operands, Ccp = 64.  A job of the kernel is one
    output region -- TAPS: dz (<= 384 columns) x x (<= 256), COND: dz (<= 384) x [c (64) | ones], OUTSKIP: u (<= 192) x [Ghat (<= 256) |
    dS (<= 256)] -- and a layer whose matrices are wider is cut into several jobs per kind (round 5: C5's 512-wide layers are 18 jobs
    in three groups of six; C2 and hps/vqwae.json stay at the five jobs of one region each).  WAE_TN_STATIC=0 keeps the any-shape
    stream-K kernel (csrc/gemm_tn_stream.hip)."""
    g = eng.g
    return (eng.dt in (L.WAE_BF16, L.WAE_F16) and eng.opt_tn_static and g.Ccp == 64 and (2 * g.Hp) % 64 == 0 and g.Rp % 128 == 0
            and g.Sp % 128 == 0)


def static_tn_split(eng):
    """Does a layer need more than one job per kind (see static_tn_geometry)?"""
    g = eng.g
    return 2 * g.Hp > 384 or g.Rp > 256 or g.Sp > 256 or g.Hp > 192


def static_tn_shape(eng, B, T):
    """... and shapes it can address: one ones column per clip inside one 32-column tile, every operand clip below 2^30 bytes
    (the per-clip buffer descriptors carry 32-bit offsets; rows before a clip wrap to offsets beyond num_records)."""
    g = eng.g
    # bytes per time row of the widest operand array (the head's dy and the first conv's one-hot operand ride in the launch too)
    return static_tn_geometry(eng) and use_stream_tn(eng) and B <= 32 and static_tn_clip_bytes(g, T) < (1 << 30)


def static_tn_clip_bytes(g, T):
    """The largest operand clip the static launch addresses, in bytes (what wae_gemm_tn_static takes as max_clip_bytes)."""
    widest = max(g.layers * 2 * g.Hp, g.Ku, g.Rp, g.Sp, g.Ccp, g.Op, P._ru(g.O, 128)) * 2
    return (T + (g.k - 1) * max(g.dilations) + 32) * widest


def static_head(eng, B, T):
    """The head's weight gradients (dW3 = dy^T h1, dW1 = dh1^T h0, their biases = column sums of dy / dh1) and -- class-id input -- the
    first conv's (onehot^T dx0) ride in the static launch as one more group of jobs when they fit its regions (P <= 384 columns,
    Q <= 256) and a layer is one group of at least five jobs (k >= 3, no cut matrices); otherwise they stay on the 128 x 128 tile
    launches (csrc/gemm_tn.hip)."""
    g = eng.g
    return (static_tn_shape(eng, B, T) and not static_tn_split(eng) and not eng.wide_head and g.k >= 3 and g.Op <= 384 and g.Sp <= 256
            and P._ru(g.O, 128) <= 384 and eng.opt_tn_static_head)


class StaticStreamTable:
    """Host builder for wae_gemm_tn_static: per layer one job per dilated-conv tap (TAPS), one for conv1x1c + the per-clip sums of
    dz (COND) and one for conv1x1_out + conv1x1_skip (OUTSKIP).  Teams and segments as StreamTable."""
    KT = 32

    def __init__(self, eng, B, T):
        self.eng, self.B, self.T = eng, B, T
        self.groups = []

    def begin_group(self):
        self.groups.append([])

    def add(self, **kw):
        f = dict(P=None, Q0=None, Q1=None, C0=None, C1=None, Cb=None, p_stride=0, q0_stride=0, q1_stride=0, ldc0=0, ldc1=0,
                 m_valid=0, n0_valid=0, n1_valid=0, shift=0, ones_col=-1, kind=0, alpha=1.0, pad_=0)
        f.update(kw)
        assert f["shift"] <= 0 and f["m_valid"] % 8 == 0 and f["n0_valid"] % 8 == 0 and f["n1_valid"] % 8 == 0
        self.groups[-1].append(L.TqJob(*[f[n] for n, _ in L.TqJob._fields_]))

    def finalize(self, ncu=None):
        """ncu: the CUs this launch may count on (default: all) -- a launch that runs BESIDE an under-filled sweep takes the idle ones"""
        eng, B, T = self.eng, self.B, self.T
        gs = len(self.groups[0])
        assert all(len(g) == gs for g in self.groups), "every layer must contribute the same list of jobs"
        ncu = ncu or torch.cuda.get_device_properties(eng.device).multi_processor_count
        spc = (T + self.KT - 1) // self.KT
        segs, team_seg = [], [0]
        self.team_size = gs
        self.nteams = max(1, ncu // gs)
        self.nwg = ncu if ncu % 8 == 0 and ncu >= self.nteams * gs else self.nteams * gs
        total = len(self.groups) * B * spc
        # The members of a team walk the same slabs, but a slab costs the out + skip member 704 operand columns (96 tiles), a tap member
        # 640 and the conditioning member 448 (36 tiles): by workgroup lives (tools/tq_member_lives.py, C2) 1 274 / 1 150-1 176 / 954 us,
        # and the launch ends with the slowest.  A member's job kind is a field of the job record, so the second half of every team's
        # share runs on a copy of the groups in which the last two slots have changed places: the two members each do half of both.
        # (One more segment per team = one more prologue + flush per member; the partial sums are added atomically either way.)
        ng = len(self.groups)
        swap = eng.opt.tn_swap and gs >= 2
        if swap:
            def swapped(grp):
                k = [j.kind if j.m_valid > 0 else -1 for j in grp]
                if k[-2:] == [L.TQ_COND, L.TQ_OUTSKIP]:
                    return grp[:-2] + [grp[-1], grp[-2]]
                return grp
            self.groups = self.groups + [swapped(grp) for grp in self.groups]
        for t in range(self.nteams):
            lo, hi = total * t // self.nteams, total * (t + 1) // self.nteams
            mid = (lo + hi) // 2 if swap else hi
            for a, b_, var in ((lo, mid, 0), (mid, hi, 1)):
                while a < b_:
                    grp = a // (B * spc)
                    end = min(b_, (grp + 1) * B * spc)
                    segs.append(L.TsSeg((var * ng + grp) * gs, a - grp * B * spc, end - grp * B * spc))
                    a = end
            team_seg.append(len(segs))
        jobs = [j for g in self.groups for j in g]
        dev = eng.device
        self.jobs_dev = torch.frombuffer(bytearray(bytes((L.TqJob * len(jobs))(*jobs))), dtype=torch.uint8).to(dev)
        self.segs_dev = torch.frombuffer(bytearray(bytes((L.TsSeg * len(segs))(*segs))), dtype=torch.uint8).to(dev)
        self.team_seg_dev = torch.tensor(team_seg, dtype=torch.int32, device=dev)
        self.stamps = None          # diagnostic builds (-DWAE_TQ_STAMPS): an int64 [nwg][16][4] tensor, zeroed before the launch
        # (team pacing of the tap members -- round 4: 6.7 -> 4.7 GB of HBM traffic at identical time, profiles/r04_tq_experiments.txt --
        #  stays reachable through wae_gemm_tn_static's own arguments; the product launches unpaced: window 0)
        self.ntaps = gs - 2
        self.window = self.window_cond = 0
        self.pace = torch.zeros(self.nteams, dtype=torch.int32, device=dev)
        return self

    def launch(self):
        eng = self.eng
        L.check(eng.lib.wae_gemm_tn_static(eng.dt, L.ptr(self.jobs_dev), L.ptr(self.segs_dev), L.ptr(self.team_seg_dev), self.nteams,
                                           self.team_size, self.nwg, self.B, self.T, L.ptr(self.stamps), L.ptr(self.pace), self.window,
                                           self.window_cond, self.ntaps, static_tn_clip_bytes(eng.g, self.T), eng.stream()),
                "gemm_tn_static")


def use_stream_tn(eng):
    """16-bit runs take every layer's weight gradients in one launch after the backward sweep (needs the residual-stream gradient
    of every layer kept); fp32 (the parity mode) keeps one wae_gemm_tn_tiles launch per layer."""
    _prepare_bwd(eng)
    return eng.opt_tn_stream


def bwd_workspace(eng, B, T):
    key = ("bwd", B, T)
    ws = eng._ws.get(key)
    if ws is None:
        g, dev, td = eng.g, eng.device, eng.tdtype
        ws = dict(dz=torch.zeros(B, T, g.layers * 2 * g.Hp, dtype=td, device=dev),
                  gx=[torch.zeros(B, T, g.Rp, dtype=td, device=dev) for _ in range(g.layers if use_stream_tn(eng) else 2)],
                  gzero=torch.zeros(B, T, g.Rp, dtype=td, device=dev),
                  gtmp=torch.zeros(B, T, g.Rp, dtype=td, device=dev) if eng.dropout > 0 else None,
                  dy=torch.zeros(B, T, g.Op, dtype=td, device=dev),
                  dh1=torch.zeros(B, T, g.Sp, dtype=td, device=dev),
                  dskip=torch.zeros(B, T, g.Sp, dtype=td, device=dev),
                  dc=torch.zeros(B, T, max(g.Ccp, 64), dtype=td, device=dev),
                  ids=torch.zeros(B, T, dtype=torch.int32, device=dev),
                  onehot=(torch.zeros(B, T, P._ru(g.O, 128), dtype=td, device=dev)
                          if static_head(eng, B, T) and not g.scalar_input else None),
                  # scalar input (first_conv is a 1x1 on one channel): operand [x | 1 | 0 ...] of its weight/bias gradient
                  xs1=torch.zeros(B, T, 64, dtype=td, device=dev) if g.scalar_input else None)
        _build_tile_tables(eng, ws, eng._ws[(B, T, True)], B, T)
        eng._ws[key] = ws
    return ws


def _build_stream_table(eng, ws, fw, B, T, l0, l1, ncu=None):
    """The stream-K launch (csrc/gemm_tn_stream.hip) over the weight gradients of layers [l0, l1): dW1 taps, dWc + zb sums, dW_out +
    bias of every layer of the range.  The whole stack is one launch; data-parallel steps cut it into an upper and a lower half so
    that the upper half's slice of the gradient arena reaches the all-reduce in the middle of backward (decoder_backward)."""
    g, sm = eng.g, eng.sm
    es = eng.w_glu.element_size()
    Z2 = 2 * g.Hp
    dzs = g.layers * Z2
    c1, co = eng.cview["c1"], eng.cview["co"]
    ia = 1.0 / eng.grad_scale
    if static_tn_shape(eng, B, T):
        coT, csT = eng.cview["coT"], eng.cview["csT"]
        rows = sm["ldoT_rows"]
        stt = StaticStreamTable(eng, B, T)
        m_taps, n_taps = _cuts(Z2, 384, 64), _cuts(g.Rp, 256, 128)
        m_cond = _cuts(Z2, 384, 32)
        m_os, n_os0, n_os1 = _cuts(g.Hp, 192, 64), _cuts(g.Rp, 256, 128), _cuts(g.Sp, 256, 128)

        def layer_jobs(l):
            d = g.dilations[l]
            dz_ptr = ws["dz"].data_ptr() + l * Z2 * es
            c1l = c1.data_ptr() + l * Z2 * sm["ld1"] * 4
            xl = fw["xd"][l] if "xd" in fw else fw["x"][l]      # dW1 contracts dz against the convolution's operand
            jobs = []
            for tap in range(g.k):
                for m0, mw in m_taps:
                    for n0, nw in n_taps:
                        jobs.append(dict(kind=L.TQ_TAPS, P=dz_ptr + m0 * es, p_stride=dzs, m_valid=mw, Q0=xl.data_ptr() + n0 * es,
                                         q0_stride=g.Rp, n0_valid=nw, shift=-(g.k - 1 - tap) * d,
                                         C0=c1l + (m0 * sm["ld1"] + tap * g.Rp + n0) * 4, ldc0=sm["ld1"], alpha=ia))
            for m0, mw in m_cond:
                jobs.append(dict(kind=L.TQ_COND, P=dz_ptr + m0 * es, p_stride=dzs, m_valid=mw, Q0=fw["c_up"].data_ptr(), q0_stride=g.Ccp,
                                 n0_valid=g.Ccp, ones_col=g.Ccp, C0=c1l + (m0 * sm["ld1"] + g.k * g.Rp) * 4, ldc0=sm["ld1"], alpha=ia))
            has_out = l < g.layers - 1       # the last layer's x' is dead (wavenet.py:205-207): no conv1x1_out gradient
            coTl = coT.data_ptr() + l * rows * g.Rp * 4
            csTl = csT.data_ptr() + l * g.Hp * g.Sp * 4
            u_ptr = fw["u"].data_ptr() + l * g.Hp * es
            for m0, mw in m_os:
                if has_out:
                    for i in range(max(len(n_os0), len(n_os1))):
                        j = dict(kind=L.TQ_OUTSKIP, P=u_ptr + m0 * es, p_stride=g.Ku, m_valid=mw, alpha=ia)
                        if i < len(n_os0):
                            n0, nw = n_os0[i]
                            j.update(Q0=ws["gx"][l + 1].data_ptr() + n0 * es, q0_stride=g.Rp, n0_valid=nw,
                                     C0=coTl + (m0 * g.Rp + n0) * 4, ldc0=g.Rp)
                            if m0 == 0:          # the out bias: column sums of Ghat, formed by the job that holds rows 0..63 of u
                                j.update(Cb=coTl + (g.Hp * g.Rp + n0) * 4)
                        if i < len(n_os1):
                            n1, nw1 = n_os1[i]
                            j.update(Q1=ws["dskip"].data_ptr() + n1 * es, q1_stride=g.Sp, n1_valid=nw1,
                                     C1=csTl + (m0 * g.Sp + n1) * 4, ldc1=g.Sp)
                        jobs.append(j)
                else:
                    # ... which frees the job's first operand: dS takes its place, and the column sums the kernel forms of that operand
                    # are the skip bias gradient (the same for every layer, modules.py:157-160)
                    for n1, nw1 in n_os1:
                        j = dict(kind=L.TQ_OUTSKIP, P=u_ptr + m0 * es, p_stride=g.Ku, m_valid=mw, Q0=ws["dskip"].data_ptr() + n1 * es,
                                 q0_stride=g.Sp, n0_valid=nw1, C0=csTl + (m0 * g.Sp + n1) * 4, ldc0=g.Sp, alpha=ia)
                        if m0 == 0:
                            j.update(Cb=eng.cview["cbs"].data_ptr() + n1 * 4)
                        jobs.append(j)
            return jobs

        # a layer's jobs form one group (a team of that many workgroups walks the layer's slabs) up to six jobs; more are dealt into
        # equal groups of at most six -- jobs that share operands (the cuts of one tap) side by side --, padded with null jobs
        njobs = max(len(layer_jobs(l)) for l in range(l0, l1))
        ngrp = -(-njobs // 6) if njobs > 6 else 1
        gsz = -(-njobs // ngrp)
        for l in range(l0, l1):
            jobs = layer_jobs(l)
            jobs += [{}] * (ngrp * gsz - len(jobs))
            for gi in range(ngrp):
                stt.begin_group()
                for j in jobs[gi * gsz:(gi + 1) * gsz]:
                    stt.add(**j)
        if static_head(eng, B, T):
            gs = g.k + 2
            first = None
            if not g.scalar_input and l0 == 0:
                W1 = P._ru(g.O, 128)
                first = dict(kind=L.TQ_TAPS, P=ws["onehot"].data_ptr(), p_stride=W1, m_valid=W1, Q0=ws["gx"][0].data_ptr(),
                             q0_stride=g.Rp, n0_valid=g.Rp, C0=eng.cview["ctab"].data_ptr(), ldc0=g.Rp, alpha=ia / RS)
            if l1 == g.layers:
                # the head: two contractions and two column-sum jobs (kind COND without a Q operand: per-clip sums of P's columns,
                # the layout the ones columns of the tile launches had); the first conv's takes the free third slot
                c3, c1h = eng.cview["c3"].data_ptr(), eng.cview["c1h"].data_ptr()
                dy, dh1 = ws["dy"].data_ptr(), ws["dh1"].data_ptr()
                grp = [dict(kind=L.TQ_TAPS, P=dy, p_stride=g.Op, m_valid=g.Op, Q0=fw["h1"].data_ptr(), q0_stride=g.Sp, n0_valid=g.Sp,
                            C0=c3, ldc0=sm["ldh"], alpha=ia),
                       dict(kind=L.TQ_TAPS, P=dh1, p_stride=g.Sp, m_valid=g.Sp, Q0=fw["h0"].data_ptr(), q0_stride=g.Sp, n0_valid=g.Sp,
                            C0=c1h, ldc0=sm["ldh"], alpha=ia)]
                grp += [first or {}] + [{}] * (gs - 5)
                grp += [dict(kind=L.TQ_COND, P=dy, p_stride=g.Op, m_valid=g.Op, ones_col=g.Sp, C0=c3, ldc0=sm["ldh"], alpha=ia),
                        dict(kind=L.TQ_COND, P=dh1, p_stride=g.Sp, m_valid=g.Sp, ones_col=g.Sp, C0=c1h, ldc0=sm["ldh"], alpha=ia)]
            else:
                grp = ([first] + [{}] * (gs - 1)) if first else []
            if grp:
                stt.begin_group()
                for j in grp:
                    stt.add(**j)
        return stt.finalize(ncu)
    assert ncu is None, "a CU budget is a feature of the static launch"
    stt = StreamTable(eng, B, T)
    for l in range(l0, l1):
        d = g.dilations[l]
        stt.begin_group()
        dz_ptr = ws["dz"].data_ptr() + l * Z2 * es
        c1l = c1.data_ptr() + l * Z2 * sm["ld1"] * 4
        xl = fw["xd"][l] if "xd" in fw else fw["x"][l]      # dW1 contracts dz against the convolution's operand
        for tap in range(g.k):
            last = tap == g.k - 1 and not g.Ccp
            stt.add(Z2, g.Rp, -(g.k - 1 - tap) * d, (g.Rp if last else -1), ia, dz_ptr, dzs, xl.data_ptr(), g.Rp,
                    c1l + tap * g.Rp * 4, sm["ld1"])
        if g.Ccp:
            stt.add(Z2, g.Ccp, 0, g.Ccp, ia, dz_ptr, dzs, fw["c_up"].data_ptr(), g.Ccp, c1l + g.k * g.Rp * 4, sm["ld1"])
        has_out = l < g.layers - 1
        stt.add(g.Rp, g.Hp, 0, g.Hp, ia, ws["gx"][l + 1].data_ptr() if has_out else 0, g.Rp,
                fw["u"].data_ptr() + l * g.Hp * es, g.Ku, co.data_ptr() + l * g.Rp * sm["ldo"] * 4, sm["ldo"])
    stt.lead_jobs = g.k
    return stt.finalize()


def _build_tile_tables(eng, ws, fw, B, T):
    """All weight-gradient contractions of a step as two kinds of launches: one table per layer (dW1 taps, dWc + zb sums,
    dW_out + bias) and one global table (dW_skip of every layer + bias, head matrices + biases, first-conv table)."""
    g, sm = eng.g, eng.sm
    es = eng.w_glu.element_size()
    Z2 = 2 * g.Hp
    dzs = g.layers * Z2
    c1, co = eng.cview["c1"], eng.cview["co"]
    ia = 1.0 / eng.grad_scale       # fp16: the 16-bit gradients are loss-scaled, the fp32 weight gradients are not
    ws["tt_layer"] = []
    ws["stream"] = None
    if use_stream_tn(eng):
        ws["stream"] = _build_stream_table(eng, ws, fw, B, T, 0, g.layers)
    for l in range(g.layers if ws["stream"] is None else 0):
        d = g.dilations[l]
        tt = TileTable(eng)
        dz_ptr = ws["dz"].data_ptr() + l * Z2 * es
        c1l = c1.data_ptr() + l * Z2 * sm["ld1"] * 4
        xl = fw["xd"][l] if "xd" in fw else fw["x"][l]
        for tap in range(g.k):
            last = tap == g.k - 1 and not g.Ccp
            tt.add(Z2, g.Rp, -(g.k - 1 - tap) * d, (g.Rp if last else -1), ia, dz_ptr, dzs, xl.data_ptr(), g.Rp,
                   c1l + tap * g.Rp * 4, sm["ld1"])
        if g.Ccp:
            tt.add(Z2, g.Ccp, 0, g.Ccp, ia, dz_ptr, dzs, fw["c_up"].data_ptr(), g.Ccp, c1l + g.k * g.Rp * 4, sm["ld1"])
        if l < g.layers - 1:
            g_next = ws["gx"][(l + 1) % len(ws["gx"])]
            tt.add(g.Rp, g.Hp, 0, g.Hp, ia, g_next.data_ptr(), g.Rp, fw["u"].data_ptr() + l * g.Hp * es, g.Ku,
                   co.data_ptr() + l * g.Rp * sm["ldo"] * 4, sm["ldo"])
        ws["tt_layer"].append(tt.finalize(B))
    tt = TileTable(eng)
    c3, c1h, cs, ctab = eng.cview["c3"], eng.cview["c1h"], eng.cview["cs"], eng.cview["ctab"]
    ws["tt_head"] = ws["tt_first"] = None
    if not static_head(eng, B, T):
        tt.add(g.Op, g.Sp, 0, g.Sp, ia, ws["dy"].data_ptr(), g.Op, fw["h1"].data_ptr(), g.Sp, c3.data_ptr(), sm["ldh"])
        tt.add(g.Sp, g.Sp, 0, g.Sp, ia, ws["dh1"].data_ptr(), g.Sp, fw["h0"].data_ptr(), g.Sp, c1h.data_ptr(), sm["ldh"])
    if not static_tn_shape(eng, B, T):    # (static: dW_skip of every layer and the skip bias ride in the stream launch)
        tt.add(g.Sp, g.Ku, 0, g.Ku, ia, ws["dskip"].data_ptr(), g.Sp, fw["u"].data_ptr(), g.Ku, cs.data_ptr(), sm["lds"])
    if tt.tiles:
        ws["tt_head"] = tt.finalize(B)
    if static_head(eng, B, T) and not g.scalar_input:
        return
    tt = TileTable(eng)
    g0 = ws["gx"][0]                                    # dxhat_0 lands in gx[0 % 2]
    if g.scalar_input:     # rows 0 / 1 of the tile = d weight / d bias:  sum_t [x[t] | 1] (x) dx0[t]
        tt.add(64, g.Rp, 0, -1, ia / RS, ws["xs1"].data_ptr(), 64, g0.data_ptr(), g.Rp, ctab.data_ptr(), g.Rp)
    else:
        tt.add(g.O, g.Rp, 0, -1, ia / RS, 0, 0, g0.data_ptr(), g.Rp, ctab.data_ptr(), g.Rp, onehot_ptr=ws["ids"].data_ptr())
    ws["tt_first"] = tt.finalize(B)


def layer_segment(eng):
    """[lo, hi) of the flat arena that holds the gated layers and the head -- ~95 % of the parameters, and final as soon as the
    weight-gradient launches and gproj_bwd have run (first_conv sits in front of it, the speaker embedding, the upsampling
    network, the encoder and the codebook behind it)."""
    lay = eng.lay
    keys = list(lay.offsets)
    last = keys.index("wavenet.last_conv_layers.3.weight_v")
    hi = lay.offsets[keys[last + 1]] if last + 1 < len(keys) else lay.total
    return lay.off("wavenet.conv_layers.0.conv.bias"), hi


def layer_segment_mid(eng):
    """First arena element of layer L/2: [layer_segment lo, mid) = the lower half of the gated layers, [mid, hi) = the upper half
    and the head -- the slice a data-parallel backward hands to the all-reduce first."""
    return eng.lay.off("wavenet.conv_layers.0.conv.bias") + (eng.g.layers // 2) * eng.lay.layer_stride


def _wn_bwd_range(eng, lo, hi):
    """weight-norm backward (d_eff -> grads) of the arena slice [lo, hi)"""
    lay = eng.lay
    r0, r1 = int(np.searchsorted(lay.wn_v_off, lo)), int(np.searchsorted(lay.wn_v_off, hi))
    L.check(eng.lib.wae_weight_norm_bwd_range(L.ptr(eng.params), L.ptr(eng.d_eff), L.ptr(eng.grads), lo, hi, L.ptr(eng.wn_v),
                                              L.ptr(eng.wn_g), L.ptr(eng.wn_c), r0, r1, eng.stream()), "weight_norm_bwd")


def decoder_backward(eng, x_ids: torch.Tensor, targets: torch.Tensor, lengths: Optional[torch.Tensor],
                     gid: Optional[torch.Tensor], gvec: Optional[torch.Tensor] = None, ext_dy: Optional[torch.Tensor] = None,
                     loss_scale: float = 1.0, grad_sync=None):
    """Backward of the last ``decoder_forward(..., train=True)`` with the same (B, T).  Fills ``eng.grads`` (flat arena,
    reference parameter layout incl. weight_g / weight_v) for every decoder parameter and returns dc (B,T,Ccp): the
    gradient wrt the upsampled local conditioning.  ``ext_dy`` (B,T,Op): external d loss / d logits (DMoL).
    ``grad_sync`` (distributed.GradSync): the layers' + head's slice of the gradient arena is finished (weight-norm backward
    of that slice) and handed to the all-reduce BEFORE the conditioning / first-conv / front-end gradients are computed, so
    the collective runs under them."""
    _prepare_bwd(eng)
    g, lib, lay, st = eng.g, eng.lib, eng.lay, eng.stream()
    B, T = x_ids.shape
    fw = eng._ws[(B, T, True)]
    ws = bwd_workspace(eng, B, T)
    es = eng.w_glu.element_size()
    sm = eng.sm
    early = eng.__dict__.pop("_early_pack", None)
    if early is not None:          # train_step queued it on a side stream right behind weight norm
        eng.join(early[0])
    else:
        pack_bwd_weights(eng)      # (also clears eng.d_eff and eng.cbuf)
    if lengths is None:
        count = B * (T - 1)
    else:
        count = int(torch.clamp(lengths.detach().to("cpu", torch.int64).clamp(max=T) - 1, min=0).sum())
    inv_count = loss_scale * eng.grad_scale / max(count, 1)
    xi = x_ids.to(torch.int32).contiguous() if not g.scalar_input else None
    tg = targets.to(torch.int32).contiguous() if targets is not None else None
    ln = lengths.to(torch.int32).to(eng.device).contiguous() if lengths is not None else None
    keep = [xi, tg, ln]

    # ---- head --------------------------------------------------------------------------------------------------
    hd = L.HeadDesc(eng.dt, B, T, g.Ku, g.Sp, g.Op, g.O, math.sqrt(1.0 / g.layers))
    b3 = ctypes.c_void_p(eng.b_head.data_ptr() + 2 * g.Sp * 4)
    if eng.wide_head:
        # the same three steps as csrc/head_bwd.hip, one wae_gemm_tm launch each (dy, dh1, dskip through HBM)
        if ext_dy is not None:
            ws["dy"].copy_(ext_dy)
        else:
            ce = L.TmCe(None, tg.data_ptr(), None, fw["lse"].data_ptr(), ln.data_ptr() if ln is not None else None, inv_count, g.O)
            _tm_ce(eng, B, T, g.Op, 6, [(fw["h1"].data_ptr(), g.Sp, g.Sp, 0)], eng.w_hwide["w3"].data_ptr(), ws["dy"].data_ptr(),
                   g.Op, b3.value, ce)
        _tm(eng, B, T, g.Sp, 4, 1.0, [(ws["dy"].data_ptr(), g.Op, g.Op, 0)], eng.w_hwide["w3t"].data_ptr(), ws["dh1"].data_ptr(),
            g.Sp, fw["h1"].data_ptr(), g.Sp)
        _tm(eng, B, T, g.Sp, 4, math.sqrt(1.0 / g.layers), [(ws["dh1"].data_ptr(), g.Sp, g.Sp, 0)], eng.w_hwide["w1t"].data_ptr(),
            ws["dskip"].data_ptr(), g.Sp, fw["h0"].data_ptr(), g.Sp)
    else:
        L.check(lib.wae_head_bwd(ctypes.byref(hd), L.ptr(fw["h0"]), L.ptr(fw["h1"]), L.ptr(eng.w_hb), b3, L.ptr(fw["lse"]), L.ptr(tg),
                                 L.ptr(ln), inv_count, L.ptr(ext_dy), L.ptr(ws["dy"]), L.ptr(ws["dh1"]), L.ptr(ws["dskip"]), st),
                "head_bwd")
        if ext_dy is not None:
            ws["dy"].copy_(ext_dy)                # the tile table points at ws["dy"]
    c3, c1h, cs = eng.cview["c3"], eng.cview["c1h"], eng.cview["cs"]
    if ws["tt_head"] is not None:
        ws["tt_head"].launch(B, T)
    if ws["onehot"] is not None:        # operand of the first conv's weight gradient (a job of the stream launch)
        L.check(lib.wae_onehot_rows(L.ptr(xi), L.ptr(ws["onehot"]), B * T, ws["onehot"].shape[-1], eng.dt, st), "onehot_rows")

    # ---- gated stack, last layer first ---------------------------------------------------------------------------
    Z2 = 2 * g.Hp
    dzs = g.layers * Z2
    c1, co = eng.cview["c1"], eng.cview["co"]
    g_next = ws["gzero"]                      # dxhat_{L} = 0: the last layer's x' is dead (wavenet.py:205-207)
    ngx = len(ws["gx"])
    ck = 64 if eng.dt in (L.WAE_BF16, L.WAE_F16) else 32
    us_off = (g.Rp // ck) * g.NP * 4 * 1024   # bytes: the W_skip chunks follow the W_out chunks in the mode-2 stream

    tm_ev = getattr(eng, "_tm_events", None)   # bench.py: {"gate": [(e0, e1), ...], "res": [...], "pair": [...]} -- HIP events around every launch

    def timed(kind, fn, stc=None):
        if tm_ev is None or stc is not None:   # (HIP events bracket the current stream's launches only)
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(eng.device))
        fn()
        e1.record(torch.cuda.current_stream(eng.device))
        tm_ev.setdefault(kind, []).append((e0, e1))

    # (b0, nb, stc): the clips of a chain and its stream -- every array of the sweep is clip-major, so a chain is the same call on its
    #  clips (engine.chain_plan); (0, B, None) = the whole batch on the current stream
    WHOLE = (0, B, None)

    def k_u(l, gn, part=WHOLE):                # du -> dz of layer l
        b0, nb, stc = part
        r = b0 * T * es
        if gn.data_ptr() == ws["gzero"].data_ptr():
            # the top layer: dx-hat above it is zero (its x' is dead), so W_out^T . 0 is left out -- the skip half of the weight stream
            # on dS alone: the same sums (+ 0 exactly), half the chunks, 32 MB of zeros not read
            timed("gate", lambda: _tm(eng, nb, T, g.Hp, 2, 1.0, [(ws["dskip"].data_ptr() + r * g.Sp, g.Sp, g.Sp, 0)],
                                      eng.w_bu.data_ptr() + l * eng.n_bu * es + us_off, ws["dz"].data_ptr() + l * Z2 * es + r * dzs, dzs,
                                      fw["z"][l].data_ptr() + r * Z2, Z2, flags=0, st=stc), stc)
            return
        timed("gate", lambda: _tm(eng, nb, T, g.Hp, 2, 1.0, [(gn.data_ptr() + r * g.Rp, g.Rp, g.Rp, 0),
                                                            (ws["dskip"].data_ptr() + r * g.Sp, g.Sp, g.Sp, 0)],
                                  eng.w_bu.data_ptr() + l * eng.n_bu * es, ws["dz"].data_ptr() + l * Z2 * es + r * dzs, dzs,
                                  fw["z"][l].data_ptr() + r * Z2, Z2, flags=0, st=stc), stc)

    seeds = getattr(eng, "_drop_seeds", None)   # set by the train-mode forward when dropout is active

    def k_x(l, gn, gc, part=WHOLE):            # dx-hat of layer l
        b0, nb, stc = part
        r = b0 * T * es
        srcs = [(ws["dz"].data_ptr() + l * Z2 * es + r * dzs, dzs, Z2, (g.k - 1 - tap) * g.dilations[l]) for tap in range(g.k)]
        if seeds is None:
            timed("res", lambda: _tm(eng, nb, T, g.Rp, 1, RS, srcs, eng.w_bx.data_ptr() + l * eng.n_bx * es, gc.data_ptr() + r * g.Rp, g.Rp,
                                     gn.data_ptr() + r * g.Rp, g.Rp, flags=P.TM_INTERLEAVE, st=stc), stc)
        else:
            # dropout: the tap contraction alone (mode 0), then out = sqrt(.5) * (g_next + keep * acc / (1 - p)) with the mask
            # the forward applied to this layer's convolution operand
            _tm(eng, B, T, g.Rp, 0, 1.0, srcs, eng.w_bx.data_ptr() + l * eng.n_bx * es, ws["gtmp"].data_ptr(), g.Rp,
                flags=P.TM_INTERLEAVE)
            L.check(lib.wae_dropout_bwd(L.ptr(ws["gtmp"]), L.ptr(gn), L.ptr(gc), B * T * g.Rp, seeds[l], eng.dropout, RS, eng.dt, st),
                    "dropout_bwd")

    def tn_layer(l):                           # per-layer weight gradients (fp32 path; bf16 takes them all at the end)
        if ws["stream"] is not None:
            return
        ev = getattr(eng, "_tn_events", None)
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(eng.device))
        ws["tt_layer"][l].launch(B, T)
        if ev is not None:
            e1.record(torch.cuda.current_stream(eng.device))
            ev.append((e0, e1))

    def launch_stream(tab):
        ev = getattr(eng, "_tn_events", None)
        if ev is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(eng.device))
        tab.launch()
        if ev is not None:
            e1.record(torch.cuda.current_stream(eng.device))
            ev.append((e0, e1))

    OP = P.ONES_PAD

    def finish_layers(l0, l1, with_head):
        """Scatter the dense gradient tiles of layers [l0, l1) (and of the head) into the effective-weight arena, then the zb chain
        (conv bias + hoisted global conditioning, modules.py:148-152) of those layers.  Disjoint slots; the tables never move."""
        key = ("scatter_jobs", l0, l1, with_head)
        jobs = ws.get(key)
        static = isinstance(ws["stream"], StaticStreamTable)
        if jobs is None:
            nb, ls = l1 - l0, lay.layer_stride

            def sjob(src, mp, rows, cols, ld, off=0, nb=1, ss=0, ds=0, unique=1, doff=0):
                return L.ScatterJob(src.data_ptr() + off * 4, mp.data_ptr(), eng.d_eff.data_ptr() + doff * 4, rows * cols, ss, ds, ld, nb,
                                    cols, unique, 0)
            lst = [sjob(c1, sm["w1"], Z2, sm["ncol1"], sm["ld1"], off=l0 * Z2 * sm["ld1"], nb=nb, ss=Z2 * sm["ld1"], ds=ls, doff=l0 * ls)]
            if static:      # transposed dW_out (+ its bias row) and dW_skip blocks of the static stream launch, per layer
                coT, csT, cbs, rows = eng.cview["coT"], eng.cview["csT"], eng.cview["cbs"], sm["ldoT_rows"]
                lst += [sjob(coT, sm["woT"], g.Hp, g.Rp, g.Rp, off=l0 * rows * g.Rp, nb=nb, ss=rows * g.Rp, ds=ls, doff=l0 * ls),
                        sjob(coT, sm["boT"], 1, g.Rp, g.Rp, off=l0 * rows * g.Rp + g.Hp * g.Rp, nb=nb, ss=rows * g.Rp, ds=ls, doff=l0 * ls),
                        sjob(csT, sm["wsT"], g.Hp, g.Sp, g.Sp, off=l0 * g.Hp * g.Sp, nb=nb, ss=g.Hp * g.Sp, ds=ls, doff=l0 * ls),
                        sjob(cbs, sm["bsT"], 1, g.Sp, g.Sp, nb=nb, ss=0, ds=ls, doff=l0 * ls)]
            else:
                lst += [sjob(co, sm["wo"], g.Rp, g.Hp, sm["ldo"], off=l0 * g.Rp * sm["ldo"], nb=nb, ss=g.Rp * sm["ldo"], ds=ls, doff=l0 * ls),
                        sjob(co, sm["bo"], g.Rp, OP, sm["ldo"], off=l0 * g.Rp * sm["ldo"] + g.Hp, nb=nb, ss=g.Rp * sm["ldo"], ds=ls, unique=2,
                             doff=l0 * ls),
                        sjob(cs, sm["bs"], g.Sp, OP, sm["lds"], off=g.Ku, nb=nb, ss=0, ds=ls, unique=2, doff=l0 * ls)]
            if with_head:
                if not static:
                    lst.append(sjob(cs, sm["ws"], g.Sp, g.Ku, sm["lds"]))
                lst += [sjob(c3, sm["w3"], g.Op, g.Sp, sm["ldh"]),
                        sjob(c3, sm["b3"], g.Op, OP, sm["ldh"], off=g.Sp, unique=2),
                        sjob(c1h, sm["w1h"], g.Sp, g.Sp, sm["ldh"]),
                        sjob(c1h, sm["b1h"], g.Sp, OP, sm["ldh"], off=g.Sp, unique=2)]
            jobs = ws[key] = (L.ScatterJob * len(lst))(*lst)
        L.check(lib.wae_unpack_scatter_add_multi(jobs, len(jobs), st), "scatter layer and head gradients")
        wg_off = lay.off("wavenet.conv_layers.0.conv1x1g.weight_v") + l0 * lay.layer_stride if g.Cg > 0 else -1
        L.check(lib.wae_gproj_bwd(L.ptr(eng.eff), L.ptr(eng.d_eff), wg_off,
                                  lay.off("wavenet.conv_layers.0.conv.bias") + l0 * lay.layer_stride, lay.layer_stride,
                                  L.ptr(gid32) if use_gid else None, emb_off, L.ptr(gvec),
                                  ctypes.c_void_p(c1.data_ptr() + l0 * Z2 * sm["ld1"] * 4), Z2 * sm["ld1"], sm["ld1"],
                                  g.k * g.Rp + g.Ccp, B, l1 - l0, g.G, g.Hp, max(g.Cg, 0), int(g.n_speakers or 0), st),
                "gproj_bwd")

    emb_off = lay.offsets.get("wavenet.embed_speakers.weight", 0)
    use_gid = gid is not None and "wavenet.embed_speakers.weight" in lay.offsets
    gid32 = gid.to(torch.int32).contiguous() if gid is not None else None
    keep.append(gid32)
    # ---- data parallel: where the arena is cut.  [seg_lo, seg_hi) = the gated layers + the head (backward.layer_segment); with the
    #      stream-K launch and >= 4 layers the sweep hands over [seg_mid, seg_hi) -- the upper half of the layers and the head -- as
    #      soon as it has passed layer L/2, and [seg_lo, seg_mid) at its end (WAE_DP_SPLIT=0: one hand-over at the end)
    eng._grads_done = None
    seg_lo, seg_hi = layer_segment(eng)
    split = None
    if grad_sync is not None and ws["stream"] is not None and g.layers >= 4 and eng.opt.dp_split:
        split = g.layers // 2
        seg_mid = layer_segment_mid(eng)
        if "stream_hi" not in ws:
            ws["stream_hi"] = _build_stream_table(eng, ws, fw, B, T, split, g.layers)
            ws["stream_lo"] = _build_stream_table(eng, ws, fw, B, T, 0, split)

    # 16-bit fused sweep, three taps, 64 conditioning columns: dc = sum_l Wc_l^T dz_l is folded into the pair launches (phase A's
    # shift-0 tap already holds dz_l[t] as MFMA operand): an fp32 running sum over the layers, written in the storage dtype by the
    # launch of layer 0 (which runs the pair kernel's first half only) -- the K = L * 2Hp launch that re-read every dz is gone
    fold_dc = bool(eng.fused_bwd and seeds is None and eng.dt in (L.WAE_BF16, L.WAE_F16) and g.k == 3 and g.Ccp == 64
                   and eng.opt.bwd_fold_dc)
    if fold_dc and "dc32" not in ws:
        ws["dc32"] = torch.empty(B, T, 64, dtype=torch.float32, device=eng.device)
    cbytes = (Z2 // 64) * 8192                # bytes of one layer's chunks in the dc weight stream (packing.bwd_c_map)
    # ---- an UNDER-FILLED sweep (hps/vqwae.json's shard: 160 workgroups per launch on 256 CUs): the weight gradients of the upper layers
    #      (+ the head's) run BESIDE the lower part of the sweep, as a launch sized for the idle CUs on a side stream; the lower layers'
    #      follow the sweep as before.  `beside` = the first layer of the upper part.  Measured at that shard (20 layers, 96 idle CUs;
    #      ms per train step, one box): 6 / 8 / 10 / 12 / 14 layers beside the sweep 3.247 / 3.226-3.239 / 3.223-3.230 / 3.230 / 3.287
    #      against 3.37-3.39 without.  The side launch runs a layer's share in ~114 us on 96 CUs (all 256: 31 -- its teams no longer sit
    #      on one XCD each, and the sweep's launches share L2 and HBM with it) and slows the sweep's launches by ~5 %; what it has not
    #      finished when the sweep ends runs beside the lower layers' launch, so the split is not critical: L / (1 + ncu / (2 idle)).
    beside = None
    if (grad_sync is None and eng.opt.side and isinstance(ws["stream"], StaticStreamTable) and seeds is None and g.layers >= 8):
        ncu_all = torch.cuda.get_device_properties(eng.device).multi_processor_count
        idle = (ncu_all - B * ((T + 255) // 256)) // 8 * 8
        if idle >= 64:
            nup = int(g.layers / (1.0 + 0.5 * ncu_all / idle))
            if nup >= 2:
                beside = g.layers - nup
                if "stream_beside" not in ws:
                    ws["stream_beside"] = _build_stream_table(eng, ws, fw, B, T, beside, g.layers, ncu=idle)
                    # with an encoder in front, the front end's backward is a chain of ~20 small launches (~0.3 ms) that starts at the end of
                    # the sweep on its own side stream (engine.backward): the lower layers' launch leaves it 32 CUs instead of taking
                    # every SIMD's registers (0 / 32 / 48 / 64 CUs left: 3.20-3.21 / 3.08-3.14 / 3.10 / 3.11 ms per step at hps/vqwae.json)
                    reserve = 32 if g.has_encoder else 0
                    ws["stream_below"] = _build_stream_table(eng, ws, fw, B, T, 0, beside, ncu=(ncu_all - reserve) if reserve else None)
    beside_done = [None]
    k_u(g.layers - 1, g_next)
    # two half-batch chains of the sweep's launches (engine.chain_plan), the second half a launch late; not with dropout (its mask
    # generator counts elements of the full batch) and not with per-layer weight-gradient launches (they read both chains' rows)
    plan = eng.chain_plan(B, T, backward=True) if (seeds is None and ws["stream"] is not None) else None

    def fork():
        if plan is None:
            return [WHOLE]
        side = eng.chain_fork(plan[1])
        return [(0, plan[0], None), (plan[0], B - plan[0], ctypes.c_void_p(side.cuda_stream))]

    if tm_ev is not None:                      # bench.py: HIP events around the whole sweep (both chains)
        sweep_e0, sweep_e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sweep_e0.record(torch.cuda.current_stream(eng.device))
    parts = fork()
    for l in range(g.layers - 1, -1, -1):
        assert l == g.layers - 1 or g_next.data_ptr() == ws["gx"][(l + 1) % ngx].data_ptr()
        tn_layer(l)                            # needs dz_l and dx_{l+1}-hat
        g_cur = ws["gx"][l % ngx]
        if (l > 0 or fold_dc) and eng.fused_bwd and seeds is None:
            # K_X of layer l and K_U of layer l-1 in one launch (csrc/glu_bwd.hip); layer 0 (fold_dc): K_X + the last dc term only
            lp = max(l - 1, 0)
            for b0, nb, stc in parts:
                d = L.GluBwdDesc(eng.dt, nb, T, g.Rp, g.Hp, g.Sp, g.k, g.dilations[l], RS)
                r = b0 * T * es
                args = [ctypes.byref(d), ctypes.c_void_p(ws["dz"].data_ptr() + l * Z2 * es + r * dzs), dzs,
                        ctypes.c_void_p(g_next.data_ptr() + r * g.Rp), ctypes.c_void_p(g_cur.data_ptr() + r * g.Rp),
                        ctypes.c_void_p(ws["dskip"].data_ptr() + r * g.Sp), ctypes.c_void_p(fw["z"][lp].data_ptr() + r * Z2),
                        ctypes.c_void_p(ws["dz"].data_ptr() + lp * Z2 * es + r * dzs),
                        ctypes.c_void_p((eng.w_bxf if hasattr(eng, "w_bxf") else eng.w_bx).data_ptr() + l * eng.n_bx * es),
                        ctypes.c_void_p(eng.w_buo.data_ptr() + lp * eng.n_buo * es),
                        ctypes.c_void_p(eng.w_bu.data_ptr() + lp * eng.n_bu * es + us_off)]
                sq = stc if stc is not None else st
                if fold_dc:
                    mode = (0 if l == g.layers - 1 else 1) | (2 if l == 0 else 0) | (4 if getattr(eng, "bwd_pair4", False) else 0)
                    # (bit 2: keep the 4-wave kernel where the 8-wave one, csrc/glu_bwd8.hip, has an instantiation -- tests and tools)
                    args += [ctypes.c_void_p(eng.w_bc.data_ptr() + l * cbytes), ctypes.c_void_p(ws["dc32"].data_ptr() + b0 * T * 64 * 4),
                             ctypes.c_void_p(ws["dc"].data_ptr() + r * ws["dc"].shape[-1]), mode, int(l == 0)]
                    timed("pair", lambda: L.check(lib.wae_glu_bwd_fused_dc(*args, sq), "glu_bwd_fused_dc"), stc)
                else:
                    timed("pair", lambda: L.check(lib.wae_glu_bwd_fused(*args, sq), "glu_bwd_fused"), stc)
        else:
            for part in parts:
                k_x(l, g_next, g_cur, part)
                if l > 0:
                    k_u(l - 1, g_cur, part)
        g_next = g_cur
        if beside is not None and l == beside:
            # dz, dx-hat and the saved activations of layers [beside, L) are complete: their weight gradients start on the idle CUs
            with eng.branch(1) as beside_done:
                launch_stream(ws["stream_beside"])
        if split is not None and l == split:
            # data parallel: dz, dx-hat and the saved activations of layers [split, L) are complete -> their weight gradients now
            # (one stream-K launch over the upper half), the head's and theirs into the arena, weight-norm backward of that slice,
            # and the all-reduce of ~half the arena starts while the lower half of the sweep still runs
            if plan is not None:
                eng.chain_join()
            launch_stream(ws["stream_hi"])
            finish_layers(split, g.layers, with_head=True)
            _wn_bwd_range(eng, seg_mid, seg_hi)
            grad_sync.ready_range(seg_mid, seg_hi)
            parts = fork()
    if plan is not None:
        eng.chain_join()
    if tm_ev is not None:
        sweep_e1.record(torch.cuda.current_stream(eng.device))
        tm_ev.setdefault("sweep", []).append((sweep_e0, sweep_e1))
    if fold_dc and eng.opt.side:
        # dc is complete: the front end's backward may start (engine.backward: a side stream).  Its launches then sit behind the
        # weight-gradient launch below, which fills every SIMD's registers, and run in that launch's ragged end (its workgroups finish
        # 60-90 us apart) and beside the scatters.  (Started behind the weight-gradient launch instead they ran beside the scatters
        # only, and both took longer: 54 us per step gained instead of 90.)
        eng._ev_dc = torch.cuda.Event()
        eng._ev_dc.record(torch.cuda.current_stream(eng.device))
    if split is not None:
        launch_stream(ws["stream_lo"])
    elif beside is not None:              # the lower layers' (the upper ones' launch has been running beside the sweep)
        launch_stream(ws["stream_below"])
        eng.join(beside_done[0])
    elif ws["stream"] is not None:        # every layer's dW1 taps, dWc + zb sums, dW_out + bias: one launch
        launch_stream(ws["stream"])
    if split is not None:
        finish_layers(0, split, with_head=False)
        _wn_bwd_range(eng, seg_lo, seg_mid)
        eng._grads_done = (seg_lo, seg_hi)
        grad_sync.ready_range(seg_lo, seg_mid)
    else:
        finish_layers(0, g.layers, with_head=True)
        # ---- data parallel without the split (too few layers, or the per-layer tile launches of fp32): the layers' + head's
        #      gradients are final -> weight-norm backward of that slice, then the all-reduce starts on its side stream while the
        #      launches below (and the front end's backward) still run
        eng._grads_done = None
        if grad_sync is not None:
            _wn_bwd_range(eng, seg_lo, seg_hi)
            eng._grads_done = (seg_lo, seg_hi)
            grad_sync.ready_range(seg_lo, seg_hi)
    # ---- local-conditioning gradient over all layers at once ---------------------------------------------------------
    if g.Ccp and not fold_dc:
        _tm(eng, B, T, g.Ccp, 0, 1.0, [(ws["dz"].data_ptr(), dzs, dzs, 0)], eng.w_bc.data_ptr(), ws["dc"].data_ptr(), g.Ccp)
    # ---- first conv: dW[r][class] = sum_t dx0[t][r] onehot(id[t])[class];  dx0 = dxhat_0 / sqrt(.5) -----------------------
    ctab, fb = eng.cview["ctab"], eng.cview["fb"]
    assert g_next.data_ptr() == ws["gx"][0].data_ptr()
    if g.scalar_input:
        ws["xs1"][:, :, 0] = x_ids.to(ws["xs1"].dtype)
        ws["xs1"][:, :, 1] = 1.0
        ws["tt_first"].launch(B, T)
        fb.copy_(ctab[g.Rp:2 * g.Rp])            # row 1: bias gradient; row 0 (weight) is scattered below
    else:
        if ws["tt_first"] is not None:
            ws["ids"].copy_(xi)
            ws["tt_first"].launch(B, T)
        L.check(lib.wae_sum_rows(L.ptr(ctab), 0, g.Rp, g.O, g.Rp, g.Rp, L.ptr(fb), st), "first bias grad")
    jobs = ws.get("scatter_jobs_first")
    if jobs is None:
        def sjob(src, mp, rows, cols, ld):
            return L.ScatterJob(src.data_ptr(), mp.data_ptr(), eng.d_eff.data_ptr(), rows * cols, 0, 0, ld, 1, cols, 1, 0)
        lst = [sjob(ctab, sm["tab"], 1 if g.scalar_input else g.O, g.Rp, g.Rp), sjob(fb, sm["fb"], 1, g.Rp, g.Rp)]
        jobs = ws["scatter_jobs_first"] = (L.ScatterJob * len(lst))(*lst)
    L.check(lib.wae_unpack_scatter_add_multi(jobs, len(jobs), st), "scatter first-conv gradients")
    eng._bwd_keep = keep
    return ws["dc"]



def layer_backward(eng, B, T, gx_hat, ds, gvec, drop_seed=None, lead=0):
    """Backward of ONE ResidualConv1dGLU layer (an engine of geometry layers == 1, wavenet_vocoder.modules.ResidualConv1dGLU) -- the
    autograd of modules.py:115-163 from the same kernels the stack uses, last train-mode forward of that (B, T):
        dz  = gate'(z) * (W_out^T gx_hat + W_skip^T ds)                   (wae_gemm_tm GATE_BWD)
        dx  = gx_hat + sum_tap W1_tap^T dz[t + (k-1-tap) d]               (wae_gemm_tm RESIDUAL, alpha 1: the true gradient)
        dc  = Wc^T dz                                                     (wae_gemm_tm PLAIN)
        dW1, dWc, per-clip sums of dz, dW_out + bias, dW_skip + bias      (wae_gemm_tn_tiles), gproj / weight-norm backward
    gx_hat (B,T,Rp) = sqrt(.5) * d loss / d x' (x' = (conv1x1_out(u) + x) sqrt(.5)), ds (B,T,Sp) = d loss / d s, both in the engine's
    storage dtype (times eng.grad_scale for fp16).  drop_seed: the seed of the dropout mask the forward applied to the convolution's
    operand (modules.py:127-128), or None.  lead > 0: a causal=False layer run on a frame `lead` steps longer (modules.ResidualConv1dGLU):
    the convolution operand's rows [0, T - lead) are x, the residual operand's rows [lead, T) are x, so the residual path's gradient of
    x[t] is gx_hat[t + lead] (gx_hat must own `lead` readable rows behind its end) and dx is returned in the convolution operand's frame.
    Fills eng.grads (finish_grads) and returns (dx (B,T,Rp), dc (B,T,Ccp) | None)."""
    _prepare_bwd(eng)
    g, lib, lay, st, sm = eng.g, eng.lib, eng.lay, eng.stream(), eng.sm
    assert g.layers == 1
    fw = eng._ws[(B, T, True)]
    es = eng.w_glu.element_size()
    Z2 = 2 * g.Hp
    key = ("lbwd", B, T)
    ws = eng._ws.get(key)
    ia = 1.0 / eng.grad_scale
    c1, co, cs = eng.cview["c1"], eng.cview["co"], eng.cview["cs"]
    if ws is None:
        dev, td = eng.device, eng.tdtype
        ws = dict(dz=torch.zeros(B, T, Z2, dtype=td, device=dev), gx=torch.zeros(B, T, g.Rp, dtype=td, device=dev),
                  gn=torch.zeros(B * T + lead, g.Rp, dtype=td, device=dev)[:B * T].view(B, T, g.Rp),
                  dskip=torch.zeros(B, T, g.Sp, dtype=td, device=dev),
                  dc=torch.zeros(B, T, max(g.Ccp, 64), dtype=td, device=dev))
        d = g.dilations[0]
        tt = TileTable(eng)
        for tap in range(g.k):
            last = tap == g.k - 1 and not g.Ccp
            # dW1 contracts dz against the convolution's operand: the masked one, the non-causal layer's [x ; 0], or x itself
            xop = fw["xd"][0] if "xd" in fw else (fw["xnc"] if lead else fw["x"][0])
            tt.add(Z2, g.Rp, -(g.k - 1 - tap) * d, (g.Rp if last else -1), ia, ws["dz"].data_ptr(), Z2, xop.data_ptr(), g.Rp,
                   c1.data_ptr() + tap * g.Rp * 4, sm["ld1"])
        if g.Ccp:
            tt.add(Z2, g.Ccp, 0, g.Ccp, ia, ws["dz"].data_ptr(), Z2, fw["c_up"].data_ptr(), g.Ccp, c1.data_ptr() + g.k * g.Rp * 4, sm["ld1"])
        tt.add(g.Rp, g.Hp, 0, g.Hp, ia, ws["gn"].data_ptr(), g.Rp, fw["u"].data_ptr(), g.Ku, co.data_ptr(), sm["ldo"])
        tt.add(g.Sp, g.Ku, 0, g.Ku, ia, ws["dskip"].data_ptr(), g.Sp, fw["u"].data_ptr(), g.Ku, cs.data_ptr(), sm["lds"])
        ws["tt"] = tt.finalize(B)
        eng._ws[key] = ws
    ws["gn"].copy_(gx_hat)
    ws["dskip"].copy_(ds)
    early = eng.__dict__.pop("_early_pack", None)
    if early is not None:          # train_step queued it on a side stream right behind weight norm
        eng.join(early[0])
    else:
        pack_bwd_weights(eng)      # (also clears eng.d_eff and eng.cbuf)
    _tm(eng, B, T, g.Hp, 2, 1.0, [(ws["gn"].data_ptr(), g.Rp, g.Rp, 0), (ws["dskip"].data_ptr(), g.Sp, g.Sp, 0)], eng.w_bu.data_ptr(),
        ws["dz"].data_ptr(), Z2, fw["z"][0].data_ptr(), Z2)
    ws["tt"].launch(B, T)
    srcs = [(ws["dz"].data_ptr(), Z2, Z2, (g.k - 1 - tap) * g.dilations[0]) for tap in range(g.k)]
    if drop_seed is None and not lead:
        _tm(eng, B, T, g.Rp, 1, 1.0, srcs, eng.w_bx.data_ptr(), ws["gx"].data_ptr(), g.Rp, ws["gn"].data_ptr(), g.Rp,
            flags=P.TM_INTERLEAVE)
    else:
        # dropout and / or the non-causal frame: the tap contraction alone, then dx[t] = gx_hat[t + lead] + keep * acc[t] / (1 - p) with
        # the forward's mask (no dropout: p = 0 keeps everything)
        if "gtmp" not in ws:
            ws["gtmp"] = torch.zeros(B, T, g.Rp, dtype=eng.tdtype, device=eng.device)
        _tm(eng, B, T, g.Rp, 0, 1.0, srcs, eng.w_bx.data_ptr(), ws["gtmp"].data_ptr(), g.Rp, flags=P.TM_INTERLEAVE)
        gn_late = ctypes.c_void_p(ws["gn"].data_ptr() + lead * g.Rp * ws["gn"].element_size())
        L.check(lib.wae_dropout_bwd(L.ptr(ws["gtmp"]), gn_late, L.ptr(ws["gx"]), B * T * g.Rp, drop_seed or 0,
                                    eng.dropout if drop_seed is not None else 0.0, 1.0, eng.dt, st), "dropout_bwd")
    if g.Ccp:
        _tm(eng, B, T, g.Ccp, 0, 1.0, [(ws["dz"].data_ptr(), Z2, Z2, 0)], eng.w_bc.data_ptr(), ws["dc"].data_ptr(), g.Ccp)
    OP = P.ONES_PAD
    jobs = ws.get("scatter")
    if jobs is None:
        def sjob(src, mp, rows, cols, ld, off=0, unique=1):
            return L.ScatterJob(src.data_ptr() + off * 4, mp.data_ptr(), eng.d_eff.data_ptr(), rows * cols, 0, 0, ld, 1, cols, unique, 0)
        lst = [sjob(c1, sm["w1"], Z2, sm["ncol1"], sm["ld1"]), sjob(co, sm["wo"], g.Rp, g.Hp, sm["ldo"]),
               sjob(co, sm["bo"], g.Rp, OP, sm["ldo"], off=g.Hp, unique=2), sjob(cs, sm["ws"], g.Sp, g.Ku, sm["lds"]),
               sjob(cs, sm["bs"], g.Sp, OP, sm["lds"], off=g.Ku, unique=2)]
        jobs = ws["scatter"] = (L.ScatterJob * len(lst))(*lst)
    L.check(lib.wae_unpack_scatter_add_multi(jobs, len(jobs), st), "scatter layer gradients")
    wg_off = lay.off("wavenet.conv_layers.0.conv1x1g.weight_v") if (g.Cg > 0 and gvec is not None) else -1
    L.check(lib.wae_gproj_bwd(L.ptr(eng.eff), L.ptr(eng.d_eff), wg_off, lay.off("wavenet.conv_layers.0.conv.bias"), lay.layer_stride,
                              None, 0, L.ptr(gvec), L.ptr(c1), Z2 * sm["ld1"], sm["ld1"], g.k * g.Rp + g.Ccp, B, 1, g.G, g.Hp,
                              max(g.Cg, 0), 0, st), "gproj_bwd")
    eng._grads_done = None
    finish_grads(eng)
    return ws["gx"], (ws["dc"] if g.Ccp else None)

def _debug_kernels(eng, B, T, l, flags_u=0, flags_x=0):
    """(du/dz launch, dx launch) of layer l as closures over the workspaces of the last train step (tools/ablate_tm.py,
    flags_*: extra wae_tm_desc flags (L.TM_ONE_WG)."""
    g, ws, fw = eng.g, eng._ws[("bwd", B, T)], eng._ws[(B, T, True)]
    es = eng.w_glu.element_size()
    Z2 = 2 * g.Hp
    dzs = g.layers * Z2
    ngx = len(ws["gx"])
    gn, gc = ws["gx"][(l + 1) % ngx], ws["gx"][l % ngx]

    def k_u():
        _tm(eng, B, T, g.Hp, 2, 1.0, [(gn.data_ptr(), g.Rp, g.Rp, 0), (ws["dskip"].data_ptr(), g.Sp, g.Sp, 0)],
            eng.w_bu.data_ptr() + l * eng.n_bu * es, ws["dz"].data_ptr() + l * Z2 * es, dzs, fw["z"][l].data_ptr(), Z2, flags=flags_u)

    def k_x():
        srcs = [(ws["dz"].data_ptr() + l * Z2 * es, dzs, Z2, (g.k - 1 - tap) * g.dilations[l]) for tap in range(g.k)]
        _tm(eng, B, T, g.Rp, 1, RS, srcs, eng.w_bx.data_ptr() + l * eng.n_bx * es, gc.data_ptr(), g.Rp, gn.data_ptr(), g.Rp,
            flags=P.TM_INTERLEAVE | flags_x)
    return k_u, k_x


def finish_grads(eng):
    """d_eff (gradient wrt effective weights) -> grads (wrt weight_g / weight_v and plain parameters); the slice that
    decoder_backward already finished for the all-reduce (eng._grads_done) is left alone."""
    lay = eng.lay
    done = getattr(eng, "_grads_done", None)
    if done is None:
        _wn_bwd_range(eng, 0, lay.total)
    else:
        if done[0] > 0:
            _wn_bwd_range(eng, 0, done[0])
        if done[1] < lay.total:
            _wn_bwd_range(eng, done[1], lay.total)
        eng._grads_done = None
    return eng.grads


def frontend_backward(eng, dc: torch.Tensor, loss_scale: float = 1.0, stop_at_quant: bool = False):
    """dc (B,T,Ccp) -> upsample stages -> conv_in -> [VQ straight-through + vq_loss -> encoder]; adds the weight
    gradients into eng.d_eff.  Uses the activations kept by the last train-mode forward."""
    g, lib, lay, st = eng.g, eng.lib, eng.lay, eng.stream()
    B, T = dc.shape[0], dc.shape[1]
    dev = eng.device
    d = torch.empty(B, g.Cc, T, dtype=torch.float32, device=dev)
    L.check(lib.wae_from_btc_scaled(L.ptr(dc), L.ptr(d), B, g.Cc, T, g.Ccp, eng.dt, 1.0 / eng.grad_scale, st), "from_btc")
    acts = eng._up_acts                      # [conv_in input | None, stage-0 input, stage-1 input, ...]
    keep = [d]
    if not g.conv_in and g.cin_pad > 0:      # plain UpsampleNetwork: the forward trimmed `indent` samples at either end (upsample.py:64-65)
        trim = g.cin_pad * int(np.prod(g.upsample_scales))
        d = torch.nn.functional.pad(d, (trim, trim))
        keep.append(d)
    act = P.UP_ACT_KINDS.get(g.up_act, 0)
    for i in range(len(g.upsample_scales) - 1, -1, -1):
        s = g.upsample_scales[i]
        xin = acts[1 + i]
        name = P.up_stage_name(g, i) + ".weight_v"
        if act:        # through the stage's activation, whose output is the next stage's input (the last stage's: kept by the forward)
            yout = acts[2 + i] if i + 1 < len(g.upsample_scales) else eng._up_last
            d = d.contiguous()
            L.check(lib.wae_act_bwd(L.ptr(yout), L.ptr(d), d.numel(), act, float(g.up_act_slope), st), "upsample activation bwd")
        din = torch.empty_like(xin)
        L.check(lib.wae_upsample_stage_bwd(L.ptr(d), L.ptr(xin), L.ptr(eng.eff[lay.off(name):]), L.ptr(din),
                                           L.ptr(eng.d_eff[lay.off(name):]), B, g.Cc, xin.shape[-1], s, st), "upsample_stage_bwd")
        d = din
        keep.append(d)
    if g.conv_in:
        cin = acts[0]
        kin = 2 * g.cin_pad + 1
        name = "wavenet.upsample_net.conv_in.weight"
        dq = torch.empty_like(cin)
        L.check(lib.wae_enc_conv_bwd(L.ptr(cin), L.ptr(eng.eff[lay.off(name):]), None, L.ptr(d), L.ptr(dq),
                                     L.ptr(eng.d_eff[lay.off(name):]), None, B, g.Cc, cin.shape[-1], g.Cc, kin, 1, 0, 0, 0, st), "conv_in bwd")
    else:
        dq = d                               # no conv_in: the stages' input gradient IS the gradient of the features
    keep.append(dq)
    eng._fe_keep = keep
    fe = getattr(eng, "_fe", None)
    if stop_at_quant:
        return dq
    if fe is not None and g.has_encoder:
        lat, quant, idx = fe["lat"], fe["quant"], fe["idx"]
        Tq = lat.shape[-1]
        dlat = torch.empty_like(lat)
        en = "vq.embedding.weight"
        L.check(lib.wae_vq_bwd(L.ptr(lat), L.ptr(quant), L.ptr(idx), L.ptr(dq), L.ptr(dlat), L.ptr(eng.d_eff[lay.off(en):]), B, g.Cc,
                               Tq, fe["beta"], loss_scale, st), "vq_bwd")
        ea = eng._enc_acts                   # ea[i] = input of block i, ea[10] = output of block 9
        dx = torch.empty_like(ea[10])
        L.check(lib.wae_enc_conv_bwd(L.ptr(ea[10]), L.ptr(eng.eff[lay.off("encoder.lin.weight"):]), None, L.ptr(dlat), L.ptr(dx),
                                     L.ptr(eng.d_eff[lay.off("encoder.lin.weight"):]), L.ptr(eng.d_eff[lay.off("encoder.lin.bias"):]),
                                     B, g.encoder_hid, Tq, g.Cc, 1, 1, 0, 0, 0, st), "lin bwd")
        keep += [dlat, dx]
        dcur = dx
        for i in range(len(P.ENCODER_BLOCKS) - 1, -1, -1):
            k, s = P.ENCODER_BLOCKS[i]
            wn_, bn_ = f"encoder.net.{i}.conv.weight", f"encoder.net.{i}.conv.bias"
            co, ci, _ = lay.shapes[wn_]
            xin, yout = ea[i], ea[i + 1]
            dxi = torch.empty_like(xin) if i > 0 else None
            L.check(lib.wae_enc_conv_bwd(L.ptr(xin), L.ptr(eng.eff[lay.off(wn_):]), L.ptr(yout), L.ptr(dcur), L.ptr(dxi),
                                         L.ptr(eng.d_eff[lay.off(wn_):]), L.ptr(eng.d_eff[lay.off(bn_):]), B, ci, xin.shape[-1], co,
                                         k, s, k // 2, 1, int(s == 1 and ci == co), st), "enc block bwd")
            keep.append(dxi)
            dcur = dxi
    eng._fe_keep = keep
    return dq
