"""BASELINE config C3 on one GPU: hps/vqwae.json in full (encoder + VQ + 20-layer decoder, R = G = S = 256), the per-GPU shard of
the global batch 64 over 8 GPUs (8 clips x 5120 samples).  Times the full train step (encoder, VQ, upsampling, decoder, backward,
clip + Adam + EMA) and the inference forward; one JSON line (44 545 B and 11.40 MFLOP per sample forward, bf16)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
B, T = 8, 5120
cfg = dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5],
           encoder_hid=256, c_in=39, K=256, cin_pad=0)
sd = O.make_state_dict(dict(cfg), 7)
out = {}
for dt in sys.argv[1:] or ["bf16"]:
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dt)
    eng.load_state_dict(sd)
    eng.init_optimizer()
    x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
    c = torch.randn(B, 39, T // 160, device="cuda")
    g = torch.randint(0, 153, (B,), device="cuda")
    def timed(fn, n):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, r
    tf, r = timed(lambda: eng.forward(x, c, g, targets=x, want_logits=False), 10)
    tt, r2 = timed(lambda: eng.train_step(x, c, g), 10)
    ns = B * T
    es = 4 if dt == "fp32" else 2
    out[dt] = dict(forward_ms=tf * 1e3, forward_samples_per_s=ns / tf, forward_hbm_frac=ns * (20 * (1088 * es) + 256 * es + 5) / tf / 8e12,
                   train_ms=tt * 1e3, train_samples_per_s=ns / tt, train_mfma_frac=3 * ns * 11.40e6 / tt / (2.5e15 if es == 2 else 157.3e12),
                   loss=float(r2["loss"]), vq_loss=float(r2["vq_loss"]), mem_GB=torch.cuda.max_memory_allocated() / 1e9)
print(json.dumps({"workload": f"C3 shard: hps/vqwae.json, {B}x{T}", **out}))
