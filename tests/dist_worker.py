"""One rank of tests/test_gpu_distributed.py: a FRESH process (started by the test with subprocess, never an exec from a process that
holds the GPU) that joins a gloo group, runs ONE data-parallel WaeEngine.train_step on its shard of golden model B -- every rank on
cuda:0, the box has one GPU -- and writes what it ended with.   python tests/dist_worker.py <rank> <world> <port> <dtype> <out.pt>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, dtype, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    from helpers import golden_model
    from wavenet_autoencoders_amd import Geometry, distributed as D
    from wavenet_autoencoders_amd.engine import WaeEngine
    r, _, w = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    cfg, sd, ins, z, ocfg = golden_model("B")
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype, device="cuda:0")
    eng.load_state_dict(sd)
    eng.init_optimizer()
    D.broadcast_params(eng.params)
    B, T = ins["x"].shape
    lo, hi = D.shard_range(B, rank, world)
    lengths = torch.tensor([T, T - 137])[lo:hi]
    gs = D.GradSync(eng, bucket_bytes=1 << 16)          # small buckets: several per hand-over, merged into one collective each
    scale, n_glob = D.ragged_ce_scale(lengths, T, hi - lo)
    res = eng.train_step(ins["x"][lo:hi].cuda(), ins["c"][lo:hi].cuda(), ins["g"][lo:hi].cuda(), lengths=lengths.cuda(), lr=4e-4,
                         grad_sync=gs, ce_scale=scale)
    torch.cuda.synchronize()
    torch.save(dict(params=eng.params.cpu(), grads=eng.grads.cpu(), n_collectives=gs.n_collectives, nbuckets=len(gs.bounds),
                    ce=float(res["ce"]), scale=scale, n_glob=n_glob, lo=lo, hi=hi), out)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
