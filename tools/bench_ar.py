#!/usr/bin/env python
"""AR synthesis speed (BASELINE config C4): hps/vqwae.json decoder, B utterances, T samples."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wae_oracle as O  # closed-form weights only
from wavenet_autoencoders_amd import Geometry  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

CFG = dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153,
           upsample_scales=[4, 4, 8, 5], cin_pad=0)
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
T = int(sys.argv[3]) if len(sys.argv) > 3 else 6400
eng = WaeEngine(Geometry.from_cfg(CFG), dtype=dtype)
eng.load_state_dict(O.make_state_dict(dict(CFG), salt=7, with_encoder=False))
lat = torch.randn(B, 64, T // 640, device="cuda")
gid = torch.zeros(B, dtype=torch.int64, device="cuda")
for _ in range(2):
    eng.incremental_forward(lat, gid, T, mode="sample")
torch.cuda.synchronize()
t0 = time.perf_counter()
out = eng.incremental_forward(lat, gid, T, mode="sample")
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"AR {dtype} B={B} T={T}: {dt:.3f} s -> {T / dt / 1e3:.2f} kHz per utterance, {B * T / dt / 1e3:.1f} kHz aggregate, "
      f"{dt / T * 1e6:.1f} us/sample")
prof = getattr(eng, "_ar_profile", None)
if prof is not None:
    f = int(prof[1])
    print("  XCC ids of the members:", prof[20:52].tolist())
    print(f"  cooperative path: same-XCD exchange = {f & 1}, XCC id of member 0 / last member = {(f >> 4) & 15} / {(f >> 8) & 15}")
if prof is not None and int(prof[2:].abs().sum()) != 0:
    import numpy as np
    pc = prof.cpu().numpy()[2:30].view(np.uint64)
    names = ["barrier after the GEMV", "gate + x' and skip shares", "-", "residual, next taps + barrier", "skip exchange", "head", "draw",
             "gate rows GEMV", "fused: x round-2 poll", "fused: vbuf writes + barrier", "exchange: stores + requests (weights, scalars, history)", "exchange: polling passes + sum"]
    print("  s_memtime ticks per sample (member 0):", {n: int(v // T) for n, v in zip(names, pc)})
