#!/usr/bin/env python3
"""Is a step bound by the host's launch rate or by the GPU?  For a bench configuration: the host time to ENQUEUE one step (the call
returns without a sync; measured with the queue kept at most one step deep, so that back-pressure of a full queue is not counted) next to
the GPU time per step (many steps, one sync), and the number of launches per step the library reports.
Usage: python tools/host_vs_gpu.py [c2|c3|c5] [train|forward] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
conf = bench.CONFIGS[name]
dev = torch.device("cuda:0")
geom = Geometry.from_cfg(conf["cfg"])
sd = O.make_state_dict(dict(conf["cfg"]), salt=conf["salt"], with_encoder=conf["encoder"])
eng = WaeEngine(geom, dtype=conf["dtype"], device=str(dev))
eng.load_state_dict(sd)
x, lat, g = bench.synth_inputs(0, dev, conf)
xi = x.to(torch.int32)
lengths = torch.full((conf["B"],), conf["T"], dtype=torch.int32, device=dev)
fwd = eng.forward if conf["encoder"] else eng.decoder_forward
if mode == "train":
    eng.init_optimizer()


def step():
    if mode == "train":
        return eng.train_step(xi, lat, g, lengths=None)["loss"]
    return fwd(xi, lat, g, targets=xi, lengths=lengths, want_logits=False)["loss"]


for _ in range(5):
    step()
torch.cuda.synchronize()
# GPU-bound rate: the queue runs as deep as the host gets ahead
t0 = time.perf_counter()
for _ in range(steps):
    step()
t_enq_deep = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# host enqueue time with an empty queue in front of every step
enq = []
for _ in range(steps):
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    enq.append(time.perf_counter() - t1)
torch.cuda.synchronize()
enq.sort()
print(f"{name} {mode} {conf['dtype']}: {t_all / steps * 1e3:.3f} ms per step end to end; host enqueue {enq[len(enq) // 2] * 1e3:.3f} ms per step "
      f"(median, empty queue; min {enq[0] * 1e3:.3f}); host loop returned after {t_enq_deep / steps * 1e3:.3f} ms per step with the queue running")
