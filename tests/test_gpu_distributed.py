"""Two ranks of the ENGINE's data-parallel path on the GPU box (round-5 finding: the world-2 gloo tests run CPU tensors, and
test_grad_sync_path_equals_plain_backward is one process).  The box has one GPU, so both ranks run on cuda:0 and meet over gloo: what
is exercised is everything but RCCL itself -- per-rank shards of a ragged global batch, the CE scale from the global mask sum
(vqwae_train.py:374-379 after the gather of :705), GradSync's three hand-overs from inside backward on the side stream, the fused
clip + Adam + EMA on the averaged arena.  The reference: `data_parallel_workaround` (vqwae_train.py:698-706) -- a step on N replicas
equals the step on the gathered batch."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from helpers import golden_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("dtype,gtol", [("fp32", 2e-5), ("bf16", 3e-2)])
def test_two_ranks_of_the_engine_equal_the_global_batch_step(dtype, gtol, tmp_path):
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    world, port = 2, str(_free_port())
    outs = [str(tmp_path / f"rank{r}.pt") for r in range(world)]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), port, dtype, outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    got = [torch.load(o) for o in outs]
    # the single-process step on the gathered (global) batch
    cfg, sd, ins, z, ocfg = golden_model("B")
    B, T = ins["x"].shape
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    eng.load_state_dict(sd)
    eng.init_optimizer()
    lengths = torch.tensor([T, T - 137])
    res = eng.train_step(ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda(), lengths=lengths.cuda(), lr=4e-4)
    torch.cuda.synchronize()
    g_ref, p_ref = eng.grads.cpu(), eng.params.cpu()
    # every rank ends with the same averaged gradients and the same parameters ...
    assert torch.equal(got[0]["grads"], got[1]["grads"]) and torch.equal(got[0]["params"], got[1]["params"])
    # ... handed over from inside backward -- 16-bit engines: the upper half of the layers + the head in the middle of the sweep, the
    # lower half at its end; fp32 (per-layer tile launches, no split): the layers + head once -- and the rest in finish(), whose two
    # ends of the arena (first conv in front; embedding, upsampling, encoder, codebook behind) are not adjacent: two collectives.
    # Every hand-over is ONE collective however many buckets it spans.
    assert got[0]["n_collectives"] == (4 if dtype == "bf16" else 3) and got[0]["nbuckets"] > got[0]["n_collectives"], \
        (got[0]["n_collectives"], got[0]["nbuckets"])
    assert got[0]["n_glob"] == float((lengths - 1).sum()) and abs(got[0]["scale"] + got[1]["scale"] - 2.0) < 1e-6
    # ... and they are the global batch's: gradients tensor by tensor (norm-wise), the masked CE as the count-weighted rank mean
    lay = eng.lay
    bad = {}
    for k in lay.offsets:
        a = got[0]["grads"][lay.off(k):lay.off(k) + lay.numel(k)]
        b = g_ref[lay.off(k):lay.off(k) + lay.numel(k)]
        err, ref = float((a - b).abs().max()), float(b.abs().max())
        if err > gtol * max(ref, 1e-8) + 1e-9:
            bad[k] = (err, ref)
    assert not bad, bad
    n = [float(T - 1), float(T - 138)]
    ce_glob = (got[0]["ce"] * n[0] + got[1]["ce"] * n[1]) / (n[0] + n[1])
    assert abs(ce_glob - float(res["ce"])) < (1e-5 if dtype == "fp32" else 2e-2) * abs(float(res["ce"]))
    # post-Adam parameters: the first Adam step moves every weight by lr * g / (|g| + eps) -- where a gradient is ~0 its sign is rounding,
    # so the comparison is on the weights whose gradient is clearly non-zero
    big = g_ref.abs() > 1e-6 * g_ref.abs().max()
    if dtype == "fp32":
        assert float((got[0]["params"] - p_ref)[big].abs().max()) < 5e-6
    assert float((got[0]["params"] - p_ref).abs().max()) <= 2 * 4e-4 + 1e-6
