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
