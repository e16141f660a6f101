"""Data feed of the training entry point: the reference's dump format, length-binned sampling, cropping and batching --
emitting class ids instead of one-hot tensors, one shard of every global batch per rank, loaded one batch ahead by a
background thread into pinned memory.

Format written by the reference's preprocess_2019.py (:131-147, :33-36): ``<dump>/<phase>/train.txt`` with lines
``out_dir|N_frames|speaker_idx|text``; each ``out_dir`` holds ``wave.npy`` (int16 mu-law ids, N*hop samples) and
``mfcc.npy`` / ``mfcc.norm.npy`` ((N, 39) float32).  Reference code restated here:
  * ``_NPYDataSource.collect_files`` (vqwae_train.py:177-232): utterances with N*hop <= max_steps + 2*cin_pad*hop are dropped;
  * ``PartialyRandomizedSimilarTimeLengthSampler`` (:249-295): sort by length, shuffle inside groups of 8 batches, shuffle
    the groups, append the shuffled remainder;
  * ``collate_fn`` (:438-552): ``s ~ U[cin_pad, N - frames - cin_pad)``, ``x[s*hop : (s+frames)*hop]``,
    ``c[s-cin_pad : s+frames+cin_pad]``, shorter clips padded with the silence class ``mulaw_quantize(0)`` (:509);
  * ``get_data_loaders`` (:1003-1060): phases ``train_no_dev`` (sampler) and ``dev`` (plain shuffle), the last batch of an epoch
    may be short (``DataLoader`` default ``drop_last=False``).
Data parallel (replaces ``scatter``, :702): every rank walks the SAME global batch list (same seed, same epoch) and takes
items [r*B/N, (r+1)*B/N) of each batch, so all ranks run the same number of steps; a short last batch is cut to a multiple of
the world size.
"""
import os
import queue
import random
import threading

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


def read_index(dump_root, phase, min_frames, n_speakers=None):
    """-> list of (dir, n_frames, speaker); utterances no longer than the crop are dropped (vqwae_train.py:207-212).
    A speaker id outside [0, n_speakers) raises here, on the host, like the reference's nn.Embedding would (IndexError)."""
    path = os.path.join(dump_root, phase, "train.txt")
    if not os.path.exists(path):
        raise Exception(f"{path} does not exist")                      # vqwae_train.py:180-183
    items = []
    with open(path, "rb") as f:
        for line in f:
            parts = line.decode("utf-8").strip().split("|")
            if len(parts) < 3:
                continue
            n = int(parts[1])
            spk = int(parts[2])
            if n_speakers is not None and spk != -1 and not 0 <= spk < n_speakers:
                raise IndexError(f"{path}: speaker id {spk} of {parts[0]} is outside [0, {n_speakers}) (hparams n_speakers)")
            if n > min_frames:
                d = parts[0] if os.path.isabs(parts[0]) else os.path.join(dump_root, phase, parts[0])
                items.append((d, n, spk))
    return items


class SimilarLengthSampler:
    """PartialyRandomizedSimilarTimeLengthSampler (vqwae_train.py:249-295): indices sorted by length; every group of
    ``batch_group_size`` (default 8 batches) is shuffled in place, the groups are permuted, the remainder is shuffled and
    appended.  The draws go through one ``random.Random`` in the reference's order (each group, the group list, the tail) --
    ``random.shuffle`` is the same Fisher-Yates walk on an array view as on a list -- and, as there, the in-place shuffles
    accumulate over epochs (the reference shuffles views of its own ``sorted_indices``)."""

    def __init__(self, lengths, batch_size=8, batch_group_size=None, seed=1234):
        lengths = np.asarray(lengths, dtype=np.int64)
        self.sorted_indices = np.argsort(lengths, kind="stable")
        self.batch_size = batch_size
        if batch_group_size is None:
            batch_group_size = min(batch_size * 8, len(lengths))
            if batch_group_size % batch_size != 0:
                batch_group_size -= batch_group_size % batch_size
        assert batch_group_size % batch_size == 0
        self.batch_group_size = batch_group_size
        self.rng = random.Random(seed)

    def __len__(self):
        return len(self.sorted_indices)

    def __iter__(self):
        idx, gsz = self.sorted_indices, self.batch_group_size
        bins = []
        ngroups = len(idx) // gsz if gsz > 0 else 0
        for i in range(ngroups):
            group = idx[i * gsz:(i + 1) * gsz]
            self.rng.shuffle(group)                                    # in place, on the view (vqwae_train.py:278)
            bins.append(group)
        self.rng.shuffle(bins)
        binned = np.concatenate(bins) if bins else np.zeros(0, dtype=np.int64)
        if len(binned) < len(idx):
            last = idx[len(binned):]
            self.rng.shuffle(last)
            binned = np.concatenate([binned, last])
        return iter(binned.tolist())


def global_batches(order, batch_size, world):
    """Cut an epoch's index order into global batches (DataLoader, drop_last=False); a short last batch keeps the largest
    multiple of `world` items so every rank gets the same number of steps and of clips per step."""
    out = []
    for i in range(0, len(order), batch_size):
        b = order[i:i + batch_size]
        if len(b) < batch_size:
            b = b[:len(b) - len(b) % world]
        if b:
            out.append(b)
    return out


class CropBatcher:
    """One rank's view of a phase: yields (x ids (B,T) int32, c (B, c_in, frames) float32, g (B,) int64, lengths (B,))
    for its slice of every global batch.  train=True: SimilarLengthSampler order; else a plain shuffle (the reference's dev
    loader, shuffle=True).  Class ids are checked against [0, n_classes) here, on the host (the kernels index tables with them)."""

    def __init__(self, items, batch_size, hop, max_time_steps, feat="mfcc.norm.npy", cin_pad=0, rank=0, world=1, seed=1234,
                 train=True, n_classes=256, pad_class=None):
        if batch_size % world != 0:
            raise ValueError("batch size % num gpu must be 0 (vqwae_train.py:754)")
        self.items = items
        self.bs, self.per, self.rank, self.world = batch_size, batch_size // world, rank, world
        self.hop, self.cin_pad, self.feat = hop, cin_pad, feat
        self.frames = (max_time_steps - max_time_steps % hop) // hop if max_time_steps is not None else None
        self.train, self.n_classes = train, n_classes
        self.pad_class = int(mulaw_quantize(0, n_classes - 1)) if pad_class is None else pad_class     # vqwae_train.py:509
        lengths = [n for _, n, _ in items]
        if world > 1 and self.frames is not None and any(n < self.frames for n in lengths):
            # Fixed-length data-parallel steps normalise the CE by world * batch * (T - 1) without a collective
            # (distributed.step_ce_scale); a clip shorter than the crop would make ONE rank's shard ragged.  Every rank sees the
            # whole index, so every rank fails here, together and before the first step -- not one rank mid-run with the others
            # parked in the gradient all-reduce.  (The reference's collate keeps such clips; train with max_time_steps=None.)
            short = sum(1 for n in lengths if n < self.frames)
            raise ValueError(f"{short} of {len(lengths)} clips are shorter than max_time_steps ({self.frames} frames): with "
                             "world > 1 drop them from the index or train with max_time_steps=None")
        self.sampler = SimilarLengthSampler(lengths, batch_size, seed=seed) if train else None
        self.order_rng = random.Random(seed + 1)        # dev shuffle: identical on every rank
        self.crop_rng = np.random.default_rng(seed + 7919 * (rank + 1))   # crops: a rank's own stream

    def __len__(self):
        n = len(self.items)
        full, rest = divmod(n, self.bs)
        return full + (1 if rest - rest % self.world > 0 else 0)

    def epoch_batches(self):
        if self.train:
            order = list(iter(self.sampler))
        else:
            order = list(range(len(self.items)))
            self.order_rng.shuffle(order)
        return global_batches(order, self.bs, self.world)

    def load(self, j):
        d, n, spk = self.items[j]
        x = np.load(os.path.join(d, "wave.npy"))
        c = np.load(os.path.join(d, self.feat))
        assert len(x) == len(c) * self.hop, f"{d}: {len(x)} samples for {len(c)} frames"   # assert_ready_for_upsampling (:434-435)
        if self.frames is not None and len(x) > self.frames * self.hop:
            s = int(self.crop_rng.integers(self.cin_pad, len(c) - self.frames - self.cin_pad))
            x = x[s * self.hop:(s + self.frames) * self.hop]
            c = c[s - self.cin_pad:s + self.frames + self.cin_pad]
        if x.size and (int(x.min()) < 0 or int(x.max()) >= self.n_classes):
            raise IndexError(f"{d}/wave.npy holds class ids outside [0, {self.n_classes})")
        return x.astype(np.int32), np.ascontiguousarray(c.T, dtype=np.float32), spk

    def collate(self, idx):
        """collate_fn (vqwae_train.py:438-552) on class ids: pad x with the silence class, c with zeros, report the lengths."""
        rows = [self.load(j) for j in idx]
        T = max(len(r[0]) for r in rows)
        F = max(r[1].shape[1] for r in rows)
        x = np.full((len(rows), T), self.pad_class, dtype=np.int32)
        c = np.zeros((len(rows), rows[0][1].shape[0], F), dtype=np.float32)
        for i, (xi, ci, _) in enumerate(rows):
            x[i, :len(xi)] = xi
            c[i, :, :ci.shape[1]] = ci
        g = torch.tensor([r[2] for r in rows], dtype=torch.int64)
        return torch.from_numpy(x), torch.from_numpy(c), g, torch.tensor([len(r[0]) for r in rows], dtype=torch.int64)

    def __iter__(self):
        for b in self.epoch_batches():
            per = len(b) // self.world
            yield self.collate(b[self.rank * per:(self.rank + 1) * per])


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


class Prefetcher:
    """Iterate `loader` one or more batches ahead: a background thread reads and collates (np.load, crops), stages the
    tensors in pinned host memory and -- when `device` is a GPU -- copies them on a side stream; the training loop takes a
    batch whose copy is already under way and makes its stream wait for that copy only.  (The reference gets the same overlap
    from DataLoader(num_workers, pin_memory), vqwae_train.py:1046-1049.)  lengths stay on the host: the step reads them there."""

    def __init__(self, loader, device=None, depth=2):
        self.loader, self.depth = loader, depth
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == "cuda"
        self.stream = torch.cuda.Stream(self.device) if self.cuda else None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        x, c, g, lengths = batch
        if not self.cuda:
            return (x, c, g, lengths), None
        host = [t.pin_memory() for t in (x, c, g)]
        with torch.cuda.stream(self.stream):
            dev = [t.to(self.device, non_blocking=True) for t in host]
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return (dev[0], dev[1], dev[2], lengths), (ev, host)

    def __iter__(self):
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()

        def work():
            try:
                if self.cuda:
                    torch.cuda.set_device(self.device)
                for batch in self.loader:
                    if stop.is_set():
                        return
                    q.put(self._stage(batch))
                q.put(None)
            except BaseException as e:      # noqa: BLE001 -- handed to the consumer
                q.put(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, sync = item
                if sync is not None:
                    torch.cuda.current_stream(self.device).wait_event(sync[0])
                    for t in batch[:3]:
                        t.record_stream(torch.cuda.current_stream(self.device))
                yield batch
        finally:
            stop.set()
            while th.is_alive():
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)
