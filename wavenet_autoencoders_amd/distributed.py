"""Data-parallel training over the GPUs of one node: one process per GPU, persistent replicas, RCCL all-reduce of the
flat gradient arena over xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests).

Replaces the reference's single-process DataParallel step (vqwae_train.py:698-706: per-step `replicate` of 30 MB of
parameters, `scatter` of one-hot inputs, `gather` of 210 MB of logits to cuda:0).  Here every rank keeps its own
replica, loads its own shard of the global batch, computes its loss locally and exchanges exactly one thing per
step: the gradient arena, in a few large buckets that are launched from the tail of the arena as the backward pass
retires layers (last layers' gradients are ready first), on a side stream so the collective overlaps the rest of
the backward.  xGMI is point-to-point (7 links x ~153 GB/s per GPU); a ring all-reduce is per-link bound, so buckets
are kept large (default 8 MiB) rather than tuned for an NVSwitch fabric.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torch.distributed.run).  -> (rank, local, world)"""
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
    return rank, local, world


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Rank r takes items [r*n/world, (r+1)*n/world) of every global batch (replaces scatter, vqwae_train.py:702).
    The reference requires batch % n_gpus == 0 (:754); so do we."""
    if n_items % world != 0:
        raise ValueError(f"global batch {n_items} is not divisible by world size {world}")
    per = n_items // world
    return rank * per, (rank + 1) * per


class GradBucketer:
    """Bucketed, overlapped all-reduce (mean) of a flat fp32 gradient arena.

    grads: 1-D tensor (the arena).  Buckets are contiguous slices of at most bucket_bytes; `cuts` (element offsets) are
    forced bucket boundaries, so that a slice of the arena that becomes final early (the gated layers + head, ~95 % of
    it: backward.layer_segment) is a whole number of buckets.  `ready_range(lo, hi)` launches every not-yet-launched
    bucket inside [lo, hi) on the side stream -- adjacent ones as ONE collective: xGMI rings are per-link bound and reach
    their bandwidth on large messages -- `ready(lo)` is ready_range(lo, end); `finish()` launches the rest, waits, and
    scales by 1/world.  With timing=True the collectives are bracketed: `last_comm_ms` (first hand-over -> last collective DONE)
    and `last_wait_ms` (how long the compute stream stood still in finish()).  On a GPU both brackets are events on the side
    stream: ProcessGroupNCCL runs a collective on a stream of its own, which first waits for the stream that was current at the
    call (the side stream: the start event) and which a work handle's wait() makes the THEN-current stream wait for -- so the
    handles are waited for under the side stream before the end event is recorded there, and the compute stream then waits for
    the side stream.  (Round 3 waited on the compute stream and recorded the end event on a side stream the collectives never
    ran on: it measured the gap between hand-overs.)  CPU tensors (gloo, the tests): host wall clock around the same points."""

    def __init__(self, grads: torch.Tensor, bucket_bytes: int = 16 << 20, group=None, cuts=(), timing: bool = False,
                 wire_dtype: str = "fp32"):
        assert grads.dim() == 1 and grads.dtype == torch.float32
        if wire_dtype not in ("fp32", "bf16"):
            raise ValueError(f"wire_dtype={wire_dtype!r}: 'fp32' (exact: the rank mean of fp32 gradients) or 'bf16'")
        # wire_dtype "bf16": every collective carries a bf16 copy of its slice (half the bytes per xGMI link) and the result is widened
        # back into the fp32 arena -- opt-in, for 16-bit engines whose gradients carry bf16 rounding already; the default keeps the
        # all-reduce in fp32.  Each rank's slice is scaled by 1 / world BEFORE the cast (exact for a power-of-two world), so the wire
        # carries the addends of the MEAN: the running sum never leaves the gradients' own range (a small tail of the arena cannot be
        # swamped by a sum W times its size, nothing overflows) and no scale follows the collective.  Error: the collective adds W
        # bf16 addends in bf16 -- at most (W - 1) roundings of 2^-9 relative to the running partial sum on top of the cast's 2^-9, i.e.
        # <= W * 2^-9 of the mean's range (world 8: 1.6e-2; measured 4e-3 at world 2, tests/test_distributed_cpu.py).  The copies live in
        # ONE persistent bf16 staging arena (round 5 allocated a fresh copy per collective and step).
        self.wire_bf16 = wire_dtype == "bf16"
        self._wires = []
        self._wire_arena = None
        self.grads = grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = grads.numel()
        per = max(1, bucket_bytes // 4)
        edges = sorted({0, n, *[int(c) for c in cuts if 0 < int(c) < n]})
        self.bounds: List[Tuple[int, int]] = []
        for a, b in zip(edges[:-1], edges[1:]):
            self.bounds += [(lo, min(b, lo + per)) for lo in range(a, b, per)]
        self.launched = [False] * len(self.bounds)
        self.handles = []
        self.comm_stream = torch.cuda.Stream(grads.device) if grads.is_cuda else None
        self.timing = timing and grads.is_cuda
        self.timing_host = timing and not grads.is_cuda
        self._ev = []
        self._t0 = None
        self.last_comm_ms = self.last_wait_ms = 0.0
        self.n_collectives = 0

    def _launch(self, i: int, j: Optional[int] = None):
        """all-reduce buckets i..j (contiguous) as ONE collective"""
        j = i if j is None else j
        lo, hi = self.bounds[i][0], self.bounds[j][1]
        for k in range(i, j + 1):
            self.launched[k] = True
        if self.world == 1:
            return
        view = self.grads[lo:hi]
        self.n_collectives += 1
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.grads.device))
            with torch.cuda.stream(self.comm_stream):
                if self.timing and not self._ev:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(self.comm_stream)
                    self._ev.append(e)
                self.handles.append(dist.all_reduce(self._wire(view, lo, hi), op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            if self.timing_host and self._t0 is None:
                import time
                self._t0 = time.perf_counter()
            self.handles.append(dist.all_reduce(self._wire(view, lo, hi), op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _wire(self, view: torch.Tensor, lo: int, hi: int) -> torch.Tensor:
        """what the collective carries for slice [lo, hi) of the arena (called under the stream the collective is issued on)"""
        if not self.wire_bf16:
            return view
        if self._wire_arena is None:
            self._wire_arena = torch.empty(self.grads.numel(), dtype=torch.bfloat16, device=self.grads.device)
        w = self._wire_arena[lo:hi]
        torch.mul(view, 1.0 / self.world, out=view)      # the addend of the mean, in place (the arena slice is overwritten by the result)
        w.copy_(view)
        self._wires.append((w, view))
        return w

    def _launch_runs(self, idx):
        """launch the not-yet-launched buckets among idx (ascending), every run of adjacent ones as one collective"""
        run = []
        for i in idx:
            if self.launched[i]:
                continue
            if run and i != run[-1] + 1:
                self._launch(run[0], run[-1])
                run = []
            run.append(i)
        if run:
            self._launch(run[0], run[-1])

    def ready_range(self, lo: int, hi: int):
        """Gradients at element offsets [lo, hi) are final."""
        self._launch_runs([i for i, (a, b) in enumerate(self.bounds) if a >= lo and b <= hi])

    def ready(self, lo: int):
        """Gradients at element offsets >= lo are final."""
        self.ready_range(lo, self.grads.numel())

    def finish(self):
        # what backward did not hand over early goes out in as few collectives as possible: every run of adjacent buckets is
        # one all-reduce (the whole arena when nothing was launched early)
        self._launch_runs(range(len(self.bounds)))
        dev = self.grads.device
        if self.timing and self.world > 1:
            w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            w0.record(torch.cuda.current_stream(dev))
        if self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                for h in self.handles:      # the SIDE stream waits for the collectives' own streams ...
                    h.wait()
                for w, view in self._wires:  # ... widens the bf16 sums back into the arena ...
                    view.copy_(w)
                if self.timing and self._ev:
                    e = torch.cuda.Event(enable_timing=True)
                    e.record(self.comm_stream)   # ... so this event fires when the last collective is done
                    self._ev.append(e)
            torch.cuda.current_stream(dev).wait_stream(self.comm_stream)
        else:
            import time
            tw0 = time.perf_counter()
            for h in self.handles:
                h.wait()
            for w, view in self._wires:
                view.copy_(w)
            if self.timing_host and self._t0 is not None:
                t1 = time.perf_counter()
                self.last_comm_ms, self.last_wait_ms = (t1 - self._t0) * 1e3, (t1 - tw0) * 1e3
            self._t0 = None
        if self.world > 1:
            if not self.wire_bf16:                       # (the bf16 wire carried the addends of the mean already)
                self.grads.mul_(1.0 / self.world)
            if self.timing:
                w1.record(torch.cuda.current_stream(dev))
                self._pending = (self._ev, (w0, w1))
        self._ev = []
        self.handles = []
        self._wires = []
        self.launched = [False] * len(self.bounds)

    def collect_timing(self):
        """(comm_ms, wait_ms) of the last finished step; synchronises the events (call outside the timed region)."""
        p = getattr(self, "_pending", None)
        if p is None:
            return (self.last_comm_ms, self.last_wait_ms) if self.timing_host else (0.0, 0.0)
        ev, (w0, w1) = p
        w1.synchronize()
        self.last_comm_ms = ev[0].elapsed_time(ev[-1]) if len(ev) == 2 else 0.0
        self.last_wait_ms = w0.elapsed_time(w1)
        return self.last_comm_ms, self.last_wait_ms


class GradSync(GradBucketer):
    """The bucketer of a WaeEngine's gradient arena, cut at the boundaries of the slices backward finishes first -- the upper half
    of the gated layers + the head in the middle of the sweep, the lower half at its end, first conv / embedding / upsampling /
    encoder / codebook in finish() (pass as train_step(grad_sync=...))."""

    def __init__(self, eng, bucket_bytes: int = 16 << 20, group=None, timing: bool = False, wire_dtype: Optional[str] = None):
        from . import backward as BW
        BW._prepare_bwd(eng)
        # cut at the slice boundaries backward hands over: [lo, mid) lower half of the gated layers, [mid, hi) upper half + head
        lo, hi = BW.layer_segment(eng)
        super().__init__(eng.grads, bucket_bytes, group, cuts=(lo, BW.layer_segment_mid(eng), hi), timing=timing,
                         wire_dtype=wire_dtype or eng.opt.dp_wire)       # (WAE_DP_WIRE: options.py)


def ragged_ce_scale(lengths: Optional[torch.Tensor], T: int, batch: int, group=None) -> Tuple[float, float]:
    """Data-parallel shards of a ragged global batch: the reference normalises the masked CE by the mask sum of the WHOLE
    gathered batch (vqwae_train.py:374-379 after the gather of :705).  Rank r holds n_r = sum_b max(min(len_b, T) - 1, 0) of
    the N = sum_r n_r loss positions; its engine normalises by n_r, so scaling its CE gradient by n_r * world / N makes the
    rank MEAN of the gradients equal the gradient of the global masked mean.  -> (scale, N).  One 1-element all-reduce, and
    only when lengths is given (equal-length crops -- every batch of the reference's own pipeline -- need none)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if lengths is None:
        return 1.0, float(world * batch * (T - 1))
    n_r = float(torch.clamp(lengths.detach().to("cpu", torch.int64).clamp(max=T) - 1, min=0).sum())
    tot = torch.tensor([n_r], dtype=torch.float64)
    if world > 1:
        if dist.get_backend(group) == "nccl":
            tot = tot.cuda()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM, group=group)
    N = float(tot.item())
    return (n_r * world / N if N > 0 else 1.0), N


def step_ce_scale(lengths: Optional[torch.Tensor], T: int, batch: int, variable_length: bool, group=None) -> Tuple[float, Optional[float]]:
    """The per-step CE scale of a data-parallel rank (what vqwae_train.py calls).  Whether the ranks meet in ragged_ce_scale's
    all-reduce is decided by a GLOBAL fact -- `variable_length`: hparams.max_time_steps is None, i.e. every shard is padded to
    its own longest clip (vqwae_train.py:455-478) -- never by what one rank observes in its own shard: the length-sorted sampler
    can hand one rank a shard of equal-length clips while another rank's is ragged, and a collective entered by some ranks only
    hangs the job (or pairs with the gradient all-reduce).  With variable lengths every rank enters, a full shard contributing
    n_r = batch * (T_r - 1) with its OWN padded length T_r; with fixed-length crops no rank does, and a shard that is ragged
    anyway is an error, not a silent local normalisation."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return 1.0, None
    if not variable_length:
        if lengths is not None and bool((lengths.detach().to("cpu") != T).any()):
            raise ValueError("a fixed-length (max_time_steps) batch holds clips shorter than the crop: the global masked mean "
                             "needs every rank in the mask-sum all-reduce; train with max_time_steps=None or drop short clips")
        return 1.0, float(world * batch * (T - 1))
    if lengths is None:
        lengths = torch.full((batch,), T, dtype=torch.int64)
    return ragged_ce_scale(lengths, T, batch, group)


def all_reduce_scalars(values: torch.Tensor, group=None, average: bool = True) -> torch.Tensor:
    """Logged per-rank scalars (loss, vq_loss, perplexity): the reference averages the per-replica values
    (vqwae_train.py:759 `torch.mean(vq_loss), torch.mean(perp)`), so the mean over ranks reproduces it."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(values, op=dist.ReduceOp.SUM, group=group)
        if average:
            values /= dist.get_world_size(group)
    return values


def masked_loss_global(loss_sum: torch.Tensor, mask_sum: torch.Tensor, group=None) -> torch.Tensor:
    """Ragged batches: the reference normalises the CE by the GLOBAL mask sum of the gathered batch
    (vqwae_train.py:379); with shards that needs sum(loss*mask) and sum(mask) reduced separately."""
    pair = torch.stack([loss_sum.reshape(()), mask_sum.reshape(())]).to(torch.float64)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(pair, op=dist.ReduceOp.SUM, group=group)
    return (pair[0] / pair[1]).to(torch.float32)


def broadcast_params(params: torch.Tensor, src: int = 0, group=None):
    """Once at start-up (persistent replicas); the reference re-broadcasts all parameters EVERY step (:701)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(params, src=src, group=group)
