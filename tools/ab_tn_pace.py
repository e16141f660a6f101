#!/usr/bin/env python
"""A/B of the team pacing window of wae_gemm_tn_stream at C2 (argv: window values): stand-alone launch time of the one weight-
gradient launch of a train step, and the gradients against the unpaced launch (pacing is timing only: the sums differ only by the
arrival order of the fp32 atomics)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import wae_oracle as O  # noqa: E402
from wavenet_autoencoders_amd import Geometry  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

dev = torch.device("cuda:0")
sd = O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False)
x, lat, g = bench.synth_inputs(0, dev)
xi = x.to(torch.int32)
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16", device="cuda:0")
eng.load_state_dict(sd)
eng.init_optimizer()
eng.train_step(xi, lat, g, lengths=None)
torch.cuda.synchronize()
st = eng._ws[("bwd",) + tuple(xi.shape)]["stream"]
ref = None
windows = [int(v) for v in (sys.argv[1:] or ["0", "2", "4", "8", "16"])]
res = {w: [] for w in windows}
for rnd in range(4):                      # interleaved rounds: the launch time drifts by 10 % over a process's life
    for w in windows:
        st.window = w
        if rnd == 0:
            eng.cbuf.zero_()
            st.launch()
            torch.cuda.synchronize()
            c = eng.cbuf.clone()
            if ref is None:
                ref = c
            print(f"window {w:3d}: max |dW - dW(unpaced)| / max|dW| = {float((c - ref).abs().max() / ref.abs().max()):.2e}", flush=True)
        for i in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st.launch()
            e1.record()
            torch.cuda.synchronize()
            if i:
                res[w].append(e0.elapsed_time(e1))
for w in windows:
    v = sorted(res[w])
    print(f"window {w:3d}: launch ms min {v[0]:.3f} median {v[len(v) // 2]:.3f} max {v[-1]:.3f}", flush=True)
# inside a train step (the launch follows the backward sweep: other cache contents, other clocks)
for w in windows:
    st.window = w
    for _ in range(3):
        eng.train_step(xi, lat, g, lengths=None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.train_step(xi, lat, g, lengths=None)
    e1.record()
    torch.cuda.synchronize()
    print(f"window {w:3d}: train step {e0.elapsed_time(e1) / 20:.3f} ms", flush=True)
