"""GPU parity: DMoL loss/gradient/sampler (a10, a11) and the fused clip+Adam+EMA step (a15)."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import load_npz, rel_err
from oracle import wae_oracle as O

pytestmark = pytest.mark.gpu


def _lib():
    from wavenet_autoencoders_amd import _lib as L
    return L, L.lib()


@pytest.mark.parametrize("lsm", [7, 9])
def test_dmol_loss_and_grad_against_golden(lsm):
    L, lib = _lib()
    z = load_npz("dmol")
    y_hat = torch.from_numpy(z["y_hat"]).cuda().contiguous()          # (B, 30, T)
    y = torch.from_numpy(z["y"]).cuda().squeeze(-1).contiguous()      # (B, T) incl. the edge cases
    B, C, T = y_hat.shape
    nll = torch.empty(B, T, device="cuda")
    dy = torch.empty_like(y_hat)
    L.check(lib.wae_dmol_loss_fwd(L.ptr(y_hat), L.ptr(y), L.ptr(nll), L.ptr(dy), B, C // 3, T, 256, -float(lsm), 0, None))
    torch.cuda.synchronize()
    assert rel_err(nll.cpu().unsqueeze(-1), z[f"loss_el_{lsm}"]) < 1e-4
    assert abs(float(nll.sum()) - float(z[f"loss_sum_{lsm}"])) < 1e-3 * abs(float(z[f"loss_sum_{lsm}"]))
    assert rel_err(dy.cpu(), z[f"grad_{lsm}"]) < 1e-3               # reduce=True sums, so the grads are per element


def test_dmol_loss_65536_classes_and_shift():
    L, lib = _lib()
    z = load_npz("dmol")
    y_hat = torch.from_numpy(z["y_hat"]).cuda().contiguous()
    y = torch.from_numpy(z["y"]).cuda().squeeze(-1).contiguous()
    B, C, T = y_hat.shape
    nll = torch.empty(B, T, device="cuda")
    L.check(lib.wae_dmol_loss_fwd(L.ptr(y_hat), L.ptr(y), L.ptr(nll), None, B, C // 3, T, 65536, -16.0, 0, None))
    assert rel_err(nll.cpu().unsqueeze(-1), z["loss_el_65536"]) < 1e-4
    # the training shift (vqwae_train.py:766): position t is scored against y[t+1]; last position is 0
    L.check(lib.wae_dmol_loss_fwd(L.ptr(y_hat), L.ptr(y), L.ptr(nll), None, B, C // 3, T, 256, -7.0, 1, None))
    ref = O.dmol_loss(torch.from_numpy(z["y_hat"])[:, :, :-1], torch.from_numpy(z["y"])[:, 1:, :], 256, -7.0, reduce=False)
    assert rel_err(nll.cpu()[:, :-1].unsqueeze(-1), ref) < 1e-4
    assert float(nll[:, -1].abs().max()) == 0.0
    lengths = torch.tensor([T, T - 11])
    out = torch.zeros(2, device="cuda")
    dl = lengths.to(torch.int32).cuda()
    L.check(lib.wae_masked_mean(L.ptr(nll), L.ptr(dl), L.ptr(out), B, T, None))
    want = O.masked_dmol_loss(torch.from_numpy(z["y_hat"]), torch.from_numpy(z["y"]), lengths, 256, -7.0)
    assert abs(float(out[0]) - float(want)) < 1e-4 * abs(float(want))


def test_dmol_sampler_against_golden():
    L, lib = _lib()
    z = load_npz("dmol")
    y_hat = torch.from_numpy(z["y_hat"]).cuda().contiguous()
    B, C, T = y_hat.shape
    out = torch.empty(B, T, device="cuda")
    um = torch.from_numpy(z["u_mix"]).cuda().contiguous()      # keep the tensors alive across the async launch
    ul = torch.from_numpy(z["u_log"]).cuda().contiguous()
    L.check(lib.wae_dmol_sample(L.ptr(y_hat), L.ptr(um), L.ptr(ul), L.ptr(out), B, C // 3, T, -7.0, 0, None))
    torch.cuda.synchronize()
    assert rel_err(out.cpu(), z["sample"]) < 1e-5


@pytest.mark.parametrize("clip", [100.0, 0.5, -1.0])
def test_clip_adam_ema_matches_oracle(clip):
    L, lib = _lib()
    n = 100003
    p0 = O.hash_fill((n,), 31, 0.5)
    g0 = O.hash_fill((n,), 32, 0.02)
    params = {"w": p0.clone()}
    m, v, sh = {"w": torch.zeros(n)}, {"w": torch.zeros(n)}, {"w": p0.clone()}
    dp, dm, dv, dsh = p0.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda(), p0.clone().cuda()
    scratch = torch.zeros(1, dtype=torch.float64, device="cuda")
    gn = torch.zeros(1, device="cuda")
    for step in (1, 2, 3):
        g = g0 * step
        want_norm = O.clip_adam_ema_step(params, {"w": g}, m, v, sh, step, 4e-4, clip_thresh=clip)
        dg = g.cuda()
        L.check(lib.wae_clip_adam_ema(L.ptr(dp), L.ptr(dg), L.ptr(dm), L.ptr(dv), L.ptr(dsh), n, L.ptr(scratch), L.ptr(gn),
                                      step, 4e-4, 0.9, 0.999, 1e-8, 0.0, clip, 0.9999, None))
        torch.cuda.synchronize()
        assert abs(float(gn) - float(want_norm)) < 1e-5 * float(want_norm)
        assert rel_err(dp.cpu(), params["w"]) < 1e-6
        assert rel_err(dsh.cpu(), sh["w"]) < 1e-6
        assert rel_err(dm.cpu(), m["w"]) < 1e-5
        assert rel_err(dv.cpu(), v["w"]) < 1e-5


def test_resume_from_a_reference_optimizer_state():
    """SURVEY 8f rank 2 (vqwae_train.py:878-892,959-976): a checkpoint written by the reference after its first step -- weights
    + torch.optim.Adam.state_dict() -- is loaded, the engine takes the second step on the same batch and must land on the
    reference's second-step weights and moments; its own optimizer dict then equals torch's."""
    import json
    from helpers import golden_model, load_npz
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd import checkpoint as CK
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, sd, ins, zm, ocfg = golden_model("A")
    z = load_npz("optim_A")
    names = json.loads(str(z["names"]))
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype="fp32")
    assert names == list(eng.lay.offsets)                           # optimizer index i = i-th arena tensor
    shapes = [eng.lay.shapes[k] for k in names]
    sizes = [int(np.prod(s)) for s in shapes]

    def split(flat):
        return [t.view(s) for t, s in zip(torch.from_numpy(flat).split(sizes), shapes)]

    eng.load_state_dict(dict(zip(names, split(z["params1"]))))
    group = dict(json.loads(str(z["param_group"])), params=list(range(len(names))))
    ref_opt = {"state": {i: {"step": torch.tensor(float(z["step1"][i])), "exp_avg": m, "exp_avg_sq": v}
                         for i, (m, v) in enumerate(zip(split(z["exp_avg1"]), split(z["exp_avg_sq1"])))},
               "param_groups": [group]}
    eng.init_optimizer()
    got_group = CK.load_adam_state_dict(eng, ref_opt)
    assert eng.opt_step == 1 and got_group["lr"] == 4e-4
    T = ins["x"].shape[1]
    res = eng.train_step(ins["x"].cuda(), ins["c"].cuda(), ins["g"].cuda(), lengths=None, lr=4e-4, clip_thresh=100.0)
    torch.cuda.synchronize()
    assert abs(float(res["loss"]) - float(z["loss2"])) < 1e-4 * abs(float(z["loss2"]))
    mine = CK.adam_state_dict(eng, 4e-4)
    assert float(mine["state"][0]["step"]) == 2.0
    for i, (k, p2, m2, v2) in enumerate(zip(names, split(z["params2"]), split(z["exp_avg2"]), split(z["exp_avg_sq2"]))):
        off, n = eng.lay.off(k), eng.lay.numel(k)
        # second step: lr * m_hat / (sqrt(v_hat) + eps); gradients agree to ~1e-4 of their range, so do the moments
        # (where a gradient cancels to ~0 -- some weight_g rows -- the step lr * m_hat / (sqrt(v_hat) + eps) inherits the
        # moment's relative error, at most 2 lr)
        gm = float(m2.abs().max()) + 1e-12
        tol = 8e-4 * torch.clamp(4 * (mine["state"][i]["exp_avg"] - m2).abs() / (m2.abs() + 1e-9), max=1.0) + 2e-6
        assert bool(((eng.params[off:off + n].view(p2.shape).cpu() - p2).abs() <= tol).all()), k
        assert float((mine["state"][i]["exp_avg"] - m2).abs().max()) < 1e-3 * gm + 1e-9, k
        assert float((mine["state"][i]["exp_avg_sq"] - v2).abs().max()) < 2e-3 * float(v2.abs().max()) + 1e-14, k


def test_multi_job_gather_and_scatter_equal_the_single_job_entries():
    """wae_pack_gather_multi / wae_unpack_scatter_add_multi (one launch for the weight families / gradient blocks of a step)
    against the single-job entries: same destinations bit for bit (plain, row-sum and atomic modes; fp32, bf16, fp16)."""
    import ctypes
    from wavenet_autoencoders_amd import _lib as L
    lib = L.lib()
    dev = "cuda"
    gen = torch.Generator().manual_seed(3)
    src = torch.randn(50000, generator=gen).to(dev)
    jobs, single = [], []
    keep = []
    for n, nb, ss, ds, dt in ((4097, 3, 9000, 5000, L.WAE_BF16), (300, 1, 0, 0, L.WAE_F32), (70001 // 7, 2, 12000, 10001, L.WAE_F16),
                              (1, 1, 0, 0, L.WAE_F32)):
        mp = torch.randint(-1, 20000, (n,), generator=gen, dtype=torch.int32).to(dev)
        tdt = {L.WAE_BF16: torch.bfloat16, L.WAE_F16: torch.float16, L.WAE_F32: torch.float32}[dt]
        d1 = torch.zeros(nb * max(ds, n), dtype=tdt, device=dev)
        d2 = torch.zeros_like(d1)
        keep += [mp, d1, d2]
        L.check(lib.wae_pack_gather(L.ptr(src), L.ptr(mp), L.ptr(d1), n, nb, ss, ds, dt, None), "single")
        jobs.append(L.GatherJob(src.data_ptr(), mp.data_ptr(), d2.data_ptr(), n, ss, ds, nb, dt))
        single.append((d1, d2))
    arr = (L.GatherJob * len(jobs))(*jobs)
    L.check(lib.wae_pack_gather_multi(arr, len(jobs), None), "multi")
    torch.cuda.synchronize()
    for d1, d2 in single:
        assert torch.equal(d1, d2)
    # scatter: unique plain adds, row sums (unique == 2) and atomics (unique == 0; integer-valued data: order-independent sums)
    sj, pairs = [], []
    for rows, cols, ld, nb, unique in ((37, 24, 32, 2, 1), (19, 128, 160, 3, 2), (64, 16, 16, 1, 0)):
        n = rows * cols
        srcm = torch.randint(-8, 9, (nb * rows * ld + 7,), generator=gen).float().to(dev)
        if unique == 1:
            mp = torch.randperm(6000, generator=gen)[:n].to(torch.int32)
            mp[::5] = -1
        elif unique == 2:
            mp = torch.randperm(6000, generator=gen)[:rows].to(torch.int32).repeat_interleave(cols)
        else:
            mp = torch.randint(0, 50, (n,), generator=gen, dtype=torch.int32)
        mp = mp.to(dev)
        d1 = torch.zeros(nb * 7000, device=dev)
        d2 = torch.zeros_like(d1)
        keep += [srcm, mp, d1, d2]
        L.check(lib.wae_unpack_scatter_add(L.ptr(srcm), L.ptr(mp), L.ptr(d1), n, nb, rows * ld, 7000, cols, ld, unique, None), "single")
        sj.append(L.ScatterJob(srcm.data_ptr(), mp.data_ptr(), d2.data_ptr(), n, rows * ld, 7000, ld, nb, cols, unique, 0))
        pairs.append((d1, d2))
    arr2 = (L.ScatterJob * len(sj))(*sj)
    L.check(lib.wae_unpack_scatter_add_multi(arr2, len(sj), None), "multi scatter")
    torch.cuda.synchronize()
    for d1, d2 in pairs:
        assert torch.equal(d1, d2) and float(d1.abs().sum()) > 0
    with pytest.raises(L.WaeError):
        L.check(lib.wae_pack_gather_multi(arr, 0, None), "no jobs")
