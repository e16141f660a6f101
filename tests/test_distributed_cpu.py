"""N>1 path on CPU: world_size-2 gloo processes exercise the bucketed gradient all-reduce, the shard ranges and
the global masked-loss normalisation."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from wavenet_autoencoders_amd import distributed as D
    r, l, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    n = 1_000_003
    g = torch.arange(n, dtype=torch.float32) * (rank + 1)
    b = D.GradBucketer(g, bucket_bytes=1 << 20)
    assert len(b.bounds) == 4 and b.bounds[-1][1] == n
    # backward retires the arena from the back in three pieces
    b.ready(700_000)
    assert b.launched == [False, False, False, True]
    b.ready(262_144)
    assert b.launched == [False, True, True, True]
    b.finish()
    want = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok_grad = torch.allclose(g, want, rtol=1e-6)
    # second step reuses the bucketer
    g.copy_(torch.full((n,), float(rank)))
    b.finish()
    ok_grad2 = torch.allclose(g, torch.full((n,), (world - 1) / 2.0))
    # the opt-in bf16 wire: the same mean to bf16 rounding, the arena stays fp32
    torch.manual_seed(5 + rank)
    g16 = torch.randn(300_001)
    ref16 = g16.clone()
    dist.all_reduce(ref16)
    ref16 /= world
    b16 = D.GradBucketer(g16, bucket_bytes=1 << 19, wire_dtype="bf16")
    b16.ready(150_000)
    b16.finish()
    err16 = float((g16 - ref16).abs().max() / ref16.abs().max())
    ok_grad2 = ok_grad2 and g16.dtype == torch.float32 and 0 < err16 < 8e-3 and not b16._wires
    lo, hi = D.shard_range(8, rank, world)
    sc = D.all_reduce_scalars(torch.tensor([float(rank), 2.0 * rank]))
    gl = D.masked_loss_global(torch.tensor(3.0 * (rank + 1)), torch.tensor(float(rank + 1)))
    p = torch.full((5,), float(rank + 7))
    D.broadcast_params(p)
    # variable-length presets (max_time_steps None): rank 0's shard happens to be all full-length (padded to ITS longest clip,
    # 100 steps), rank 1's is ragged at 80 -- both must enter the mask-sum all-reduce (a per-rank decision would hang here)
    Tr, ln = (100, torch.tensor([100, 100])) if rank == 0 else (80, torch.tensor([80, 50]))
    scale, N = D.step_ce_scale(ln, Tr, 2, variable_length=True)
    fixed = D.step_ce_scale(torch.tensor([64, 64]), 64, 2, variable_length=False)     # fixed crops: no collective on any rank
    try:
        D.step_ce_scale(torch.tensor([64, 60]), 64, 2, variable_length=False)
        raised = False
    except ValueError:
        raised = True
    q.put((rank, ok_grad, ok_grad2, (lo, hi), sc.tolist(), float(gl), p.tolist(), (scale, N, fixed, raised)))
    dist.destroy_process_group()


def test_gloo_world2_bucketed_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, rng, sc, gl, pv, (scale, N, fixed, raised) in res:
        assert ok1 and ok2
        n_r = (2 * 99, 79 + 49)[rank]
        assert N == 2 * 99 + 79 + 49 and abs(scale - n_r * 2 / N) < 1e-12
        assert fixed == (1.0, float(2 * 2 * 63)) and raised
        assert rng == (rank * 4, rank * 4 + 4)
        assert sc == [0.5, 1.0]
        assert abs(gl - 3.0) < 1e-6          # (3*1 + 3*2) / (1 + 2)
        assert pv == [7.0] * 5


def test_shard_range_requires_divisible_batch():
    from wavenet_autoencoders_amd import distributed as D
    assert D.shard_range(64, 3, 8) == (24, 32)
    with pytest.raises(ValueError):
        D.shard_range(10, 0, 4)


def _ddp_worker(rank, world, port, q):
    """One data-parallel rank of a RAGGED global batch: local loss normalised by the local mask count (what the engine does),
    CE gradient scaled by distributed.ragged_ce_scale, arena all-reduced through the bucketer with the layer-segment cuts."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from helpers import golden_model
    from oracle import wae_oracle as O
    from wavenet_autoencoders_amd import distributed as D
    D.init_from_env("gloo")
    cfg, sd, ins, z, ocfg = golden_model("A")
    T = ins["x"].shape[1]
    # global batch of 4 clips = the two golden clips and their swapped copies; ragged lengths; rank r takes clips [2r, 2r+2)
    xin = torch.cat([ins["xin"], ins["xin"].flip(0)]); x = torch.cat([ins["x"], ins["x"].flip(0)])
    c = torch.cat([ins["c"], ins["c"].flip(0) * 0.5]); g = torch.cat([ins["g"], ins["g"].flip(0)])
    lengths = torch.tensor([T, T - 300, T - 901, T - 77])
    lo, hi = D.shard_range(4, rank, world)
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y, vq, perp, _ = O.vqvae_forward(psd, ocfg, xin[lo:hi], c[lo:hi], g[lo:hi])
    ce_local = O.masked_ce_loss(y, x[lo:hi].unsqueeze(-1), lengths[lo:hi])           # normalised by the LOCAL mask sum
    scale, N = D.ragged_ce_scale(lengths[lo:hi], T, hi - lo)
    (ce_local * scale + vq).backward()
    keys = sorted(psd)
    flat = torch.cat([(psd[k].grad if psd[k].grad is not None else torch.zeros_like(psd[k])).reshape(-1) for k in keys])
    n = flat.numel()
    b = D.GradBucketer(flat, bucket_bytes=64 << 10, cuts=(n // 5, n // 2))
    assert any(bb[0] == n // 5 for bb in b.bounds) and any(bb[1] == n // 2 for bb in b.bounds)
    b.ready_range(n // 5, n // 2)                                                     # the slice backward finishes first
    assert all(l == (n // 5 <= a and e <= n // 2) for l, (a, e) in zip(b.launched, b.bounds))
    b.finish()
    n_local = float(torch.clamp(lengths[lo:hi] - 1, min=0).sum())
    gl = D.masked_loss_global(ce_local.detach() * n_local, torch.tensor(n_local))
    q.put((rank, flat.numpy(), float(gl), float(N), float(vq)))
    dist.destroy_process_group()


def test_gloo_world2_ragged_step_equals_the_global_batch():
    """vqwae_train.py:374-379,698-706,759: the reference gathers the replicas' logits and takes ONE masked mean over the global
    batch, plus the mean of the replicas' vq_loss.  Two ranks with local normalisation + ragged_ce_scale + the averaged
    all-reduce must produce exactly that loss and gradient."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ddp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import golden_model
    from oracle import wae_oracle as O
    cfg, sd, ins, z, ocfg = golden_model("A")
    T = ins["x"].shape[1]
    xin = torch.cat([ins["xin"], ins["xin"].flip(0)]); x = torch.cat([ins["x"], ins["x"].flip(0)])
    c = torch.cat([ins["c"], ins["c"].flip(0) * 0.5]); g = torch.cat([ins["g"], ins["g"].flip(0)])
    lengths = torch.tensor([T, T - 300, T - 901, T - 77])
    psd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ys, vqs = [], []
    for r in range(world):                          # the replicas of data_parallel_workaround, then the gather
        y, vq, perp, _ = O.vqvae_forward(psd, ocfg, xin[2 * r:2 * r + 2], c[2 * r:2 * r + 2], g[2 * r:2 * r + 2])
        ys.append(y)
        vqs.append(vq)
    ce = O.masked_ce_loss(torch.cat(ys), x.unsqueeze(-1), lengths)
    (ce + torch.stack(vqs).mean()).backward()
    keys = sorted(psd)
    want = torch.cat([(psd[k].grad if psd[k].grad is not None else torch.zeros_like(psd[k])).reshape(-1) for k in keys]).numpy()
    import numpy as np
    for rank, flat, gl, N, vq in res:
        assert abs(gl - float(ce)) < 1e-5 * float(ce)
        assert N == float(torch.clamp(lengths - 1, min=0).sum())
        assert np.abs(flat - want).max() < 2e-5 * np.abs(want).max()
    assert np.array_equal(res[0][1], res[1][1])     # both ranks hold the same reduced arena


def _timing_worker(rank, world, port, q):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from wavenet_autoencoders_amd import distributed as D
    D.init_from_env("gloo")
    g = torch.ones(200_000, dtype=torch.float32) * (rank + 1)
    b = D.GradBucketer(g, bucket_bytes=1 << 20, timing=True)
    dist.barrier()
    if rank == 1:
        time.sleep(0.3)          # a slow peer: rank 0's collective cannot complete before rank 1 enters it
    t0 = time.perf_counter()
    b.ready(0)                   # hand the whole arena over
    b.finish()
    wall_ms = (time.perf_counter() - t0) * 1e3
    comm_ms, wait_ms = b.collect_timing()
    q.put((rank, comm_ms, wait_ms, wall_ms, float(g[0])))
    dist.destroy_process_group()


def test_reported_allreduce_time_covers_the_collective():
    """The `allreduce.ms` bench.py reports must be first hand-over -> last collective DONE, not the gap between hand-over points:
    with a peer that enters 0.3 s late, the fast rank's figure has to include the wait for it."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, comm0, wait0, wall0, v0), (r1, comm1, wait1, wall1, v1) = res
    assert v0 == v1 == 1.5
    assert comm0 >= 250.0 and comm0 <= wall0 + 1.0, (comm0, wall0)     # rank 0 waited for the slow peer, and says so
    assert wait0 >= 250.0
