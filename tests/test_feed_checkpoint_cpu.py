"""Host logic either side of the hot path, on CPU: the dump reader / sampler / crop batcher / prefetcher
(vqwae_train.py:163-321,438-552,1003-1060) and the optimizer-state mapping of checkpoints (:878-892,959-976)."""
import os
import types

import numpy as np
import pytest
import torch

from wavenet_autoencoders_amd import data as DT
from wavenet_autoencoders_amd import packing as P

HOP = 160


def _make_dump(root, phase, n_utts, rng, min_frames=20, max_frames=90, speakers=3):
    os.makedirs(os.path.join(root, phase), exist_ok=True)
    lines = []
    for i in range(n_utts):
        n = int(rng.integers(min_frames, max_frames))
        d = os.path.join(root, phase, f"utt{i:03d}") + os.sep
        os.makedirs(d, exist_ok=True)
        wave = rng.integers(0, 256, size=n * HOP).astype(np.int16)
        wave[0] = i % 256                                               # tags the utterance
        np.save(os.path.join(d, "wave.npy"), wave)
        np.save(os.path.join(d, "mfcc.norm.npy"), rng.standard_normal((n, 39)).astype(np.float32))
        lines.append(f"{d}|{n}|{i % speakers}|dummy text")
    with open(os.path.join(root, phase, "train.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


@pytest.fixture()
def dump(tmp_path):
    rng = np.random.default_rng(3)
    _make_dump(str(tmp_path), "train_no_dev", 53, rng)
    _make_dump(str(tmp_path), "dev", 9, rng)
    return str(tmp_path)


def test_sampler_is_a_length_binned_permutation():
    rng = np.random.default_rng(0)
    lengths = rng.integers(10, 500, size=203)
    s = DT.SimilarLengthSampler(lengths, batch_size=8, seed=5)
    assert s.batch_group_size == 64
    e1, e2 = list(iter(s)), list(iter(s))
    assert sorted(e1) == list(range(203)) and sorted(e2) == list(range(203))        # vqwae_train.py:1036-1038's sanity check
    assert e1 != e2
    # every full group of 64 consecutive draws holds one contiguous slice of the length-sorted list: similar lengths together
    order = np.argsort(lengths, kind="stable")
    rank_of = np.empty(203, dtype=np.int64)
    rank_of[order] = np.arange(203)
    for k in range(3):
        r = np.sort(rank_of[e1[64 * k:64 * (k + 1)]])
        assert r[-1] - r[0] == 63 and r[0] % 64 == 0
    assert sorted(rank_of[e1[192:]]) == list(range(192, 203))                         # the remainder comes last, shuffled
    # same seed -> same epochs (every rank builds the same global order)
    s2 = DT.SimilarLengthSampler(lengths, batch_size=8, seed=5)
    assert list(iter(s2)) == e1 and list(iter(s2)) == e2


def test_sampler_reproduces_the_reference_class():
    """Orders produced by the reference's own PartialyRandomizedSimilarTimeLengthSampler (tests/golden/make_golden.py cuts the class
    out of vqwae_train.py:249-295 and runs it after random.seed(1234)): identical index for index over three epochs when lengths
    are unique; with tied lengths torch.sort is not stable, so the LENGTH at every position is what is pinned."""
    import json
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sampler_ref.npz"))
    for i, case in enumerate(json.loads(str(z["cases"]))):
        lengths, want = z[f"lengths{i}"], z[f"orders{i}"]
        s = DT.SimilarLengthSampler(lengths, batch_size=case["batch_size"], seed=1234)
        for ep in range(3):
            got = np.array(list(iter(s)))
            if case["tied"]:
                assert np.array_equal(lengths[got], lengths[want[ep]]), (i, ep)
                assert sorted(got.tolist()) == list(range(case["N"]))
            else:
                assert np.array_equal(got, want[ep]), (i, ep)


def test_reader_drops_short_clips_and_checks_speaker_ids(dump):
    items = DT.read_index(dump, "train_no_dev", 32, n_speakers=3)
    assert all(n > 32 for _, n, _ in items) and 0 < len(items) < 53
    with pytest.raises(IndexError):
        DT.read_index(dump, "train_no_dev", 32, n_speakers=2)                          # nn.Embedding would raise (wavenet.py:185-190)
    with pytest.raises(Exception):
        DT.read_index(dump, "nonexistent", 32)


def test_ranks_walk_the_same_global_batches(dump):
    items = DT.read_index(dump, "train_no_dev", 32, n_speakers=3)
    world, bs = 2, 8
    loaders = [DT.CropBatcher(items, bs, HOP, 32 * HOP, rank=r, world=world, seed=11) for r in range(world)]
    assert len(loaders[0]) == len(loaders[1])
    for epoch in range(2):
        per_rank = [list(iter(ld)) for ld in loaders]
        assert len(per_rank[0]) == len(per_rank[1]) == len(loaders[0])                 # equal step counts: no unmatched collective
        seen = []
        for b0, b1 in zip(*per_rank):
            assert b0[0].shape == b1[0].shape                                          # equal clips per step
            assert b0[0].shape[1] == 32 * HOP and b0[1].shape[1:] == (39, 32)
            assert b0[0].dtype == torch.int32 and int(b0[0].min()) >= 0 and int(b0[0].max()) < 256
            assert bool((b0[3] == 32 * HOP).all())
            seen += [b0[0].shape[0], b1[0].shape[0]]
        # every kept utterance once per epoch, but for the < world items cut from the short last batch
        assert len(items) - sum(seen) < world
    # the crop is a window of the file: x[s*hop:(s+frames)*hop] with c[s:s+frames] (vqwae_train.py:472-476)
    ld = DT.CropBatcher(items[:8], 8, HOP, 32 * HOP, seed=1)
    x, c, g, ln = next(iter(ld))
    order = ld.sampler.sorted_indices                                                  # not needed: match rows by content
    for i in range(8):
        hit = False
        for d, n, spk in items[:8]:
            w = np.load(os.path.join(d, "wave.npy"))
            f = np.load(os.path.join(d, "mfcc.norm.npy"))
            for s in range(0, n - 32):
                if np.array_equal(w[s * HOP:(s + 32) * HOP], x[i].numpy()):
                    assert np.allclose(f[s:s + 32].T, c[i].numpy()) and int(g[i]) == spk
                    hit = True
        assert hit


def test_bad_class_ids_raise_on_the_host(dump):
    items = DT.read_index(dump, "train_no_dev", 32)
    d = items[0][0]
    w = np.load(os.path.join(d, "wave.npy"))
    w[5] = 300
    np.save(os.path.join(d, "wave.npy"), w)
    ld = DT.CropBatcher(items[:1], 1, HOP, None, seed=1)
    with pytest.raises(IndexError):
        next(iter(ld))


def test_dev_phase_and_ragged_collate(dump):
    items = DT.read_index(dump, "dev", 0)
    ld = DT.CropBatcher(items, 4, HOP, None, train=False, seed=2)                      # max_time_steps None: whole clips, padded
    batches = list(iter(ld))
    assert sum(b[0].shape[0] for b in batches) == len(items)
    x, c, g, ln = batches[0]
    assert x.shape[1] == int(ln.max()) and c.shape[2] * HOP == x.shape[1]
    i = int(ln.argmin())
    assert bool((x[i, int(ln[i]):] == 127).all())                                     # mulaw_quantize(0) pads (vqwae_train.py:509)
    assert bool((c[i, :, int(ln[i]) // HOP:] == 0).all())


def test_prefetcher_yields_the_loader_in_order_and_forwards_errors(dump):
    items = DT.read_index(dump, "train_no_dev", 32)
    a = DT.CropBatcher(items, 8, HOP, 32 * HOP, seed=4)
    b = DT.CropBatcher(items, 8, HOP, 32 * HOP, seed=4)
    for (x0, c0, g0, l0), (x1, c1, g1, l1) in zip(iter(a), DT.Prefetcher(b, None, depth=2)):
        assert torch.equal(x0, x1) and torch.equal(c0, c1) and torch.equal(g0, g1)

    class Boom:
        def __len__(self):
            return 2

        def __iter__(self):
            yield 1, 2, 3, 4
            raise RuntimeError("disk error")

    with pytest.raises(RuntimeError, match="disk error"):
        list(DT.Prefetcher(Boom(), None))
    # leaving the loop early stops the thread
    it = iter(DT.Prefetcher(a, None, depth=1))
    next(it)
    it.close()


# ---------------------------------------------------------------------------------------------------------------- checkpoints
CFG = dict(layers=4, stacks=2, R=32, G=48, S=32, O=64, Cc=16, Cg=8, k=3, n_speakers=5, upsample_scales=[4, 4, 8, 5],
           encoder_hid=32, c_in=39, K=32, cin_pad=0)


def _fake_engine():
    """the attributes checkpoint.py touches, on CPU tensors"""
    lay = P.ParamLayout(P.Geometry.from_cfg(CFG))
    eng = types.SimpleNamespace(lay=lay)
    eng.exp_avg = torch.zeros(lay.total)
    eng.exp_avg_sq = torch.zeros(lay.total)
    eng.opt_step = 0
    return eng


def test_adam_state_round_trips_through_torch_optim():
    """The engine's optimizer state is written as torch.optim.Adam.state_dict() (what the reference saves, vqwae_train.py:881)
    and a real torch Adam over the reference's parameter list accepts it; torch's own state loads back bit for bit."""
    from wavenet_autoencoders_amd import checkpoint as CK
    eng = _fake_engine()
    lay = eng.lay
    gen = torch.Generator().manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(lay.shapes[k], generator=gen)) for k in lay.offsets]
    opt = torch.optim.Adam(params, lr=4e-4, eps=1e-8)
    assert CK.adam_state_dict(eng, 4e-4)["state"] == {}                                # before the first step: empty, like torch's
    for _ in range(3):
        for p in params:
            p.grad = torch.randn(p.shape, generator=gen)
        opt.step()
    sd = opt.state_dict()
    group = CK.load_adam_state_dict(eng, sd)
    assert eng.opt_step == 3 and group["lr"] == 4e-4
    for i, k in enumerate(lay.offsets):
        off, n = lay.off(k), lay.numel(k)
        assert torch.equal(eng.exp_avg[off:off + n].view(lay.shapes[k]), sd["state"][i]["exp_avg"])
        assert torch.equal(eng.exp_avg_sq[off:off + n].view(lay.shapes[k]), sd["state"][i]["exp_avg_sq"])
    out = CK.adam_state_dict(eng, 4e-4)
    opt2 = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in params], lr=1e-3)
    opt2.load_state_dict(out)                                                          # torch accepts the engine's dict
    st2 = opt2.state_dict()["state"]
    assert all(torch.equal(st2[i]["exp_avg"], sd["state"][i]["exp_avg"]) and float(st2[i]["step"]) == 3.0 for i in st2)
    assert opt2.state_dict()["param_groups"][0]["lr"] == 4e-4
    # torch 0.4-era states carry python-int steps (README.md:20)
    old = {"state": {i: dict(s, step=3) for i, s in sd["state"].items()}, "param_groups": sd["param_groups"]}
    eng2 = _fake_engine()
    CK.load_adam_state_dict(eng2, old)
    assert eng2.opt_step == 3 and torch.equal(eng2.exp_avg, eng.exp_avg)


def test_unusable_optimizer_state_raises_instead_of_resetting():
    from wavenet_autoencoders_amd import checkpoint as CK
    eng = _fake_engine()
    with pytest.raises(ValueError):
        CK.load_adam_state_dict(eng, {"state": {}, "param_groups": [{"params": [0, 1, 2]}]})
    with pytest.raises(ValueError):
        CK.load_adam_state_dict(eng, {"moments": 1})
    n = len(eng.lay.offsets)
    with pytest.raises(NotImplementedError):
        CK.load_adam_state_dict(eng, {"state": {}, "param_groups": [{"params": list(range(n)), "amsgrad": True}]})
    # round-1 checkpoints of this repository still load
    CK.load_adam_state_dict(eng, dict(layout="flat-arena", exp_avg=torch.ones(eng.lay.total), exp_avg_sq=torch.ones(eng.lay.total), step=7))
    assert eng.opt_step == 7 and float(eng.exp_avg.sum()) == eng.lay.total
