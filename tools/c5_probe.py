"""Probe: does a C5-like wide configuration (R = G = S = 512) run?  (residual/skip widths above 256 are
not supported yet -- this prints the error each dtype raises.)"""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
cfg = dict(layers=4, stacks=2, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=8, upsample_scales=[4, 4, 4, 5], cin_pad=0)
sd = O.make_state_dict(dict(cfg), salt=3, with_encoder=False)
for dt in ("bf16", "fp32"):
    try:
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dt)
        eng.load_state_dict(sd)
        eng.init_optimizer()
        B, T = 2, 1280
        x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
        lat = torch.randn(B, 64, T // 320, device="cuda")
        g = torch.randint(0, 8, (B,), device="cuda")
        r = eng.train_step(x, lat, g)
        torch.cuda.synchronize()
        print(dt, "train_step ok, loss", float(r["loss"]))
    except Exception as e:
        print(dt, "FAILED:", type(e).__name__, str(e)[:300])
