"""Data feed of the training entry point: the reference's dump format, cropping and batching, emitting class ids
instead of one-hot tensors and sharding utterances per rank.

Format written by the reference's preprocess_2019.py (:131-147, :33-36): ``<dump>/<phase>/train.txt`` with lines
``out_dir|N_frames|speaker_idx|text``; each ``out_dir`` holds ``wave.npy`` (int16 mu-law ids, N*hop samples) and
``mfcc.npy`` / ``mfcc.norm.npy`` ((N, 39) float32).  Cropping follows collate_fn (vqwae_train.py:455-478):
``s ~ U[cin_pad, N - frames - cin_pad)``, ``x[s*hop : (s+frames)*hop]``, ``c[s-cin_pad : s+frames+cin_pad]``.
"""
import os

import numpy as np
import torch


def mulaw_quantize(x, mu=255):
    """nnmnkwii.preprocessing.mulaw_quantize restated: sign(x) log(1+mu|x|)/log(1+mu) -> [0, mu] ints."""
    x = np.asarray(x, dtype=np.float64)
    y = np.sign(x) * np.log1p(mu * np.abs(x)) / np.log1p(mu)
    return ((y + 1) / 2 * mu).astype(np.int64)


def inv_mulaw_quantize(y, mu=255):
    y = 2 * np.asarray(y, dtype=np.float64) / mu - 1
    return np.sign(y) * (1.0 / mu) * ((1.0 + mu) ** np.abs(y) - 1.0)


def read_index(dump_root, phase, min_frames):
    """-> list of (dir, n_frames, speaker) ; utterances shorter than the crop are dropped (vqwae_train.py:207-212)."""
    path = os.path.join(dump_root, phase, "train.txt")
    items = []
    with open(path, "rb") as f:
        for line in f:
            parts = line.decode("utf-8").strip().split("|")
            if len(parts) < 3:
                continue
            n = int(parts[1])
            if n > min_frames:
                d = parts[0] if os.path.isabs(parts[0]) else os.path.join(dump_root, phase, parts[0])
                items.append((d, n, int(parts[2])))
    return items


class CropBatcher:
    """Yields (x ids (B,T) int32, c (B, c_in, frames) float32, g (B,) int64, lengths (B,)) for ONE rank's shard."""

    def __init__(self, items, batch_size, hop, max_time_steps, feat="mfcc.norm.npy", cin_pad=0, rank=0, world=1, seed=1234):
        self.items = items[rank::world]
        self.bs, self.hop, self.cin_pad, self.feat = batch_size, hop, cin_pad, feat
        self.frames = (max_time_steps - max_time_steps % hop) // hop
        self.rng = np.random.default_rng(seed + rank)

    def __len__(self):
        return len(self.items) // self.bs

    def __iter__(self):
        order = self.rng.permutation(len(self.items))
        for i in range(0, len(order) - self.bs + 1, self.bs):
            xs, cs, gs = [], [], []
            for j in order[i:i + self.bs]:
                d, n, spk = self.items[j]
                x = np.load(os.path.join(d, "wave.npy"))
                c = np.load(os.path.join(d, self.feat))
                assert len(x) == (len(c) - 2 * 0) * self.hop           # assert_ready_for_upsampling (:434-435)
                s = int(self.rng.integers(self.cin_pad, len(c) - self.frames - self.cin_pad))
                xs.append(x[s * self.hop:(s + self.frames) * self.hop].astype(np.int32))
                cs.append(c[s - self.cin_pad:s + self.frames + self.cin_pad].T.astype(np.float32))
                gs.append(spk)
            x = torch.from_numpy(np.stack(xs))
            yield x, torch.from_numpy(np.stack(cs)), torch.tensor(gs, dtype=torch.int64), torch.full((len(xs),), x.shape[1])


class SyntheticBatcher:
    """Dataset-free stand-in with the statistics of SURVEY 8d (ids U{0..255}, MFCC-like N(0,1), speakers U{0..n})."""

    def __init__(self, batch_size, hop, max_time_steps, c_in=39, n_speakers=153, steps=100, rank=0, seed=1234):
        self.bs, self.T, self.F, self.c_in, self.n, self.steps = batch_size, max_time_steps, max_time_steps // hop, c_in, n_speakers, steps
        self.rng = np.random.default_rng(seed + rank)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            x = torch.from_numpy(self.rng.integers(0, 256, size=(self.bs, self.T), dtype=np.int32))
            c = torch.from_numpy(self.rng.standard_normal((self.bs, self.c_in, self.F)).astype(np.float32))
            g = torch.from_numpy(self.rng.integers(0, self.n, size=(self.bs,), dtype=np.int64))
            yield x, c, g, torch.full((self.bs,), self.T)
