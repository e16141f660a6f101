#!/usr/bin/env python
"""C2-size check of wae_gemm_tn_stream: gradients against the per-layer tile path, and stand-alone launch time."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import wae_oracle as O  # noqa: E402
from wavenet_autoencoders_amd import Geometry, backward as BW  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

dev = torch.device("cuda:0")
sd = O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False)
x, lat, g = bench.synth_inputs(0, dev)
xi = x.to(torch.int32)
grads = {}
for mode in ("0", "1"):
    os.environ["WAE_TN_STREAM"] = mode
    eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16", device="cuda:0")
    eng.load_state_dict(sd)
    eng.init_optimizer()
    captured = {}
    r = eng.train_step(xi, lat, g, lengths=None, grad_hook=lambda gr: captured.setdefault("g", gr.clone()))
    torch.cuda.synchronize()
    grads[mode] = captured["g"]
    for i in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.train_step(xi, lat, g, lengths=None)
        e1.record()
        torch.cuda.synchronize()
        print("mode", mode, "train_step ms", round(e0.elapsed_time(e1), 3))
    if mode == "1":
        st = eng._ws[("bwd",) + tuple(xi.shape)]["stream"]
        for i in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            st.launch()
            e1.record()
            torch.cuda.synchronize()
            print("stand-alone stream launch ms", round(e0.elapsed_time(e1), 3))
a, b = grads["0"], grads["1"]
print("grad max |diff|", float((a - b).abs().max()), "max |grad|", float(a.abs().max()), "rel", float((a - b).abs().max() / a.abs().max()))
lay = eng.lay
worst = []
for k in lay.offsets:
    sl = slice(lay.off(k), lay.off(k) + lay.numel(k))
    ref = float(a[sl].abs().max())
    worst.append((float((a[sl] - b[sl]).abs().max()) / max(ref, 1e-12), k, ref))
worst.sort(reverse=True)
for w in worst[:5]:
    print("  %.3e  %-55s  max|g| %.3e" % w)
