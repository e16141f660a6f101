"""GPU tests of the small kernels round 4 rewrote, straight through the C ABI against plain torch: weight norm on arenas with rows of
every kind (short, long, misaligned, longer than the register form) and gaps of every size (biases, un-normed tensors of hundreds of
thousands of floats), the multi-job pack gather, the one-hot operand, gproj with more than 32 clips, the head from a stored h0."""
import ctypes
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from wavenet_autoencoders_amd import _lib as L
    return L, L.lib()


def _arena(seed, rows_spec, gaps_spec):
    """rows_spec: list of (nrows, cols) blocks; gaps_spec: gap (floats) in front of every block (the g scalars of a block sit at the end
    of its gap, as torch's weight_norm registers them) and one trailing gap."""
    rng = np.random.RandomState(seed)
    v_off, g_off, cols, off = [], [], [], 0
    for (nr, nc), gap in zip(rows_spec, gaps_spec):
        off += gap
        g0 = off
        off += nr                       # weight_g
        for r in range(nr):
            v_off.append(off); g_off.append(g0 + r); cols.append(nc)
            off += nc
    total = off + gaps_spec[-1]
    params = rng.randn(total).astype(np.float32)
    return params, np.asarray(v_off, np.int64), np.asarray(g_off, np.int64), np.asarray(cols, np.int32)


@pytest.mark.parametrize("case", ["mixed", "big_gaps", "no_rows"])
def test_weight_norm_forward_and_backward_on_synthetic_arenas(case):
    L, lib = _lib()
    if case == "mixed":      # aligned short / long rows, a misaligning 9-float block, rows longer than the register form (1024)
        params, v, g, c = _arena(1, [(64, 64), (3, 9), (48, 184), (5, 11), (32, 768), (7, 1500), (16, 1024), (40, 256)],
                                 [5, 17, 0, 3, 129, 1, 64, 2, 33])
    elif case == "big_gaps":  # un-normed tensors between the rows (the encoder's convolutions, a codebook)
        params, v, g, c = _arena(2, [(32, 64), (32, 768), (8, 64)], [300001, 14000, 70000, 123457])
    else:
        params, v, g, c = _arena(3, [], [5000])
    n = len(c)
    dev = "cuda"
    P = torch.from_numpy(params).to(dev)
    eff = torch.full_like(P, float("nan"))
    tv, tg, tc = (torch.from_numpy(a).to(dev) for a in (v, g, c))
    L.check(lib.wae_weight_norm_fwd(L.ptr(P), L.ptr(eff), P.numel(), L.ptr(tv) if n else None, L.ptr(tg) if n else None,
                                    L.ptr(tc) if n else None, n, None))
    want = params.astype(np.float64).copy()
    for r in range(n):
        row = params[v[r]:v[r] + c[r]].astype(np.float64)
        want[v[r]:v[r] + c[r]] = params[g[r]] * row / np.sqrt((row * row).sum())
    got = eff.cpu().numpy()
    assert not np.isnan(got).any()
    assert np.abs(got - want).max() < 2e-6 * max(1.0, np.abs(want).max())
    # backward on the whole arena and on a sub-range cut at row boundaries
    d_eff = torch.from_numpy(np.random.RandomState(9).randn(P.numel()).astype(np.float32)).to(dev)
    wantg = d_eff.cpu().numpy().astype(np.float64).copy()
    for r in range(n):
        row = params[v[r]:v[r] + c[r]].astype(np.float64)
        dw = wantg[v[r]:v[r] + c[r]].copy()
        nrm = np.sqrt((row * row).sum())
        dv = (row * dw).sum()
        wantg[v[r]:v[r] + c[r]] = params[g[r]] / nrm * (dw - row * dv / nrm ** 2)
        wantg[g[r]] = dv / nrm
    grads = torch.full_like(P, float("nan"))
    L.check(lib.wae_weight_norm_bwd(L.ptr(P), L.ptr(d_eff), L.ptr(grads), P.numel(), L.ptr(tv) if n else None, L.ptr(tg) if n else None,
                                    L.ptr(tc) if n else None, n, None))
    gg = grads.cpu().numpy()
    assert not np.isnan(gg).any()
    assert np.abs(gg - wantg).max() < 5e-6 * max(1.0, np.abs(wantg).max())
    if n >= 4:
        # a sub-range cut where backward.py cuts it: [start of a block's g scalars, start of a later block's g scalars)
        starts = [r for r in range(n) if r == 0 or g[r] != g[r - 1] + 1]
        r0, r1 = starts[1], starts[-1]
        lo, hi = int(g[r0]), int(g[r1])
        grads.fill_(float("nan"))
        L.check(lib.wae_weight_norm_bwd_range(L.ptr(P), L.ptr(d_eff), L.ptr(grads), lo, hi, L.ptr(tv), L.ptr(tg), L.ptr(tc), r0, r1, None))
        gg = grads.cpu().numpy()
        assert not np.isnan(gg[lo:hi]).any() and np.isnan(gg[:lo]).all() and np.isnan(gg[hi:]).all()
        assert np.abs(gg[lo:hi] - wantg[lo:hi]).max() < 5e-6 * max(1.0, np.abs(wantg).max())


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_pack_gather_multi_against_indexing(dtype):
    L, lib = _lib()
    td = dict(bf16=torch.bfloat16, fp16=torch.float16, fp32=torch.float32)[dtype]
    dt = dict(bf16=L.WAE_BF16, fp16=L.WAE_F16, fp32=L.WAE_F32)[dtype]
    rng = np.random.RandomState(4)
    src = torch.from_numpy(rng.randn(50000).astype(np.float32)).cuda()
    jobs, wants, keep = [], [], []
    for n, nb, ss, ds in ((1001, 5, 7000, 1001), (4096, 1, 0, 0), (37, 9, 3001, 64), (2, 3, 11, 5)):   # odd n, odd strides, one batch
        mp = rng.randint(-1, 6000, size=n).astype(np.int32)
        m = torch.from_numpy(mp).cuda()
        dst = torch.full((max(ds, n) * nb + 8,), 7.0, dtype=td, device="cuda")
        jobs.append(L.GatherJob(src.data_ptr(), m.data_ptr(), dst.data_ptr(), n, ss, ds, nb, dt))
        keep += [m, dst]
        want = torch.full_like(dst, 7.0)
        for b in range(nb):
            vals = torch.where(m >= 0, src[(m.clamp(min=0) + b * ss).long()], torch.zeros((), device="cuda"))
            want[b * ds:b * ds + n] = vals.to(td)
        wants.append((dst, want))
    arr = (L.GatherJob * len(jobs))(*jobs)
    L.check(lib.wae_pack_gather_multi(arr, len(jobs), None))
    torch.cuda.synchronize()
    for dst, want in wants:
        assert torch.equal(dst, want)


def test_onehot_rows():
    L, lib = _lib()
    ids = torch.tensor([0, 5, 255, 256, -1, 130, 7], dtype=torch.int32, device="cuda")
    for td, dt in ((torch.bfloat16, L.WAE_BF16), (torch.float16, L.WAE_F16), (torch.float32, L.WAE_F32)):
        out = torch.full((ids.numel(), 256), 3.0, dtype=td, device="cuda")
        L.check(lib.wae_onehot_rows(L.ptr(ids), L.ptr(out), ids.numel(), 256, dt, None))
        want = torch.zeros(ids.numel(), 256, device="cuda")
        for i, v in enumerate(ids.tolist()):
            if 0 <= v < 256:
                want[i, v] = 1.0
        assert torch.equal(out.float(), want)


def test_softmax_bct_forward_and_backward_against_torch():
    from wavenet_autoencoders_amd.losses import softmax_bct
    torch.manual_seed(3)
    x = (torch.randn(3, 256, 301, device="cuda") * 4).requires_grad_(True)
    w = torch.randn(3, 256, 301, device="cuda")
    p = softmax_bct(x)
    (p * w).sum().backward()
    gx = x.grad.clone()
    x.grad = None
    pr = torch.softmax(x, dim=1)
    (pr * w).sum().backward()
    assert (p - pr).abs().max() < 2e-6 and (gx - x.grad).abs().max() < 2e-6 * max(1.0, float(x.grad.abs().max()))


def test_bmm_f32_against_torch():
    L, lib = _lib()
    torch.manual_seed(4)
    a = torch.randn(5, 37, 50, device="cuda")
    b = torch.randn(5, 50, 21, device="cuda")
    c = torch.full((5, 37, 21), 9.0, device="cuda")
    L.check(lib.wae_bmm_f32(L.ptr(a), L.ptr(b), L.ptr(c), 5, 37, 50, 21, 50, 21, 21, 37 * 50, 50 * 21, 37 * 21, 0.5, None))
    want = 0.5 * torch.bmm(a.double(), b.double())
    assert (c.double() - want).abs().max() < 1e-5
