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

    grads: 1-D tensor (the arena).  Buckets are contiguous slices of ~bucket_bytes, numbered from the FRONT of the
    arena; backward fills the arena from the back, so `ready(lo)` launches every not-yet-launched bucket that lies
    entirely at or above element offset `lo`.  `finish()` launches the rest, waits, and scales by 1/world."""

    def __init__(self, grads: torch.Tensor, bucket_bytes: int = 8 << 20, group=None):
        assert grads.dim() == 1 and grads.dtype == torch.float32
        self.grads = grads
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        n = grads.numel()
        per = max(1, bucket_bytes // 4)
        self.bounds: List[Tuple[int, int]] = [(lo, min(n, lo + per)) for lo in range(0, n, per)]
        self.launched = [False] * len(self.bounds)
        self.handles = []
        self.comm_stream = torch.cuda.Stream(grads.device) if grads.is_cuda else None

    def _launch(self, i: int, j: Optional[int] = None):
        """all-reduce buckets i..j (contiguous) as ONE collective"""
        j = i if j is None else j
        lo, hi = self.bounds[i][0], self.bounds[j][1]
        for k in range(i, j + 1):
            self.launched[k] = True
        if self.world == 1:
            return
        view = self.grads[lo:hi]
        if self.comm_stream is not None:
            self.comm_stream.wait_stream(torch.cuda.current_stream(self.grads.device))
            with torch.cuda.stream(self.comm_stream):
                self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def ready(self, lo: int):
        """Gradients at element offsets >= lo are final."""
        for i in range(len(self.bounds) - 1, -1, -1):
            if self.bounds[i][0] < lo:
                break
            if not self.launched[i]:
                self._launch(i)

    def finish(self):
        # what backward did not hand over early goes out in as few collectives as possible: every run of adjacent buckets is
        # one all-reduce (the whole arena when nothing was launched early -- a ring all-reduce over xGMI is per-link bound and
        # reaches its bandwidth only on large messages; six 8-MiB calls pay six launch latencies for nothing)
        i = len(self.bounds) - 1
        while i >= 0:
            if self.launched[i]:
                i -= 1
                continue
            j = i
            while i - 1 >= 0 and not self.launched[i - 1]:
                i -= 1
            self._launch(i, j)
            i -= 1
        for h in self.handles:
            h.wait()
        if self.comm_stream is not None:
            torch.cuda.current_stream(self.grads.device).wait_stream(self.comm_stream)
        if self.world > 1:
            self.grads.mul_(1.0 / self.world)
        self.handles = []
        self.launched = [False] * len(self.bounds)


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
