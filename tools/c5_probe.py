"""BASELINE config C5 on one GPU: 48 layers / 4 stacks, R = G = S = 512, Cc = 64, Cg = 32, the per-GPU shard of the global
batch 128 over 8 GPUs (16 clips x 5120 samples), bf16.  Times the inference forward and the full train step and prints one
JSON line (204 289 B and 104.6 MFLOP per sample forward)."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
L = int(os.environ.get("C5_LAYERS", "48"))
B, T = int(os.environ.get("C5_B", "16")), 5120
cfg = dict(layers=L, stacks=4, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=8, upsample_scales=[4, 4, 4, 5], cin_pad=0)
sd = O.make_state_dict(dict(cfg), salt=3, with_encoder=False)
out = {}
for dt in sys.argv[1:] or ["bf16"]:
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dt)
    eng.load_state_dict(sd)
    eng.init_optimizer()
    x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
    lat = torch.randn(B, 64, T // 320, device="cuda")
    g = torch.randint(0, 8, (B,), device="cuda")
    def timed(fn, n):
        for _ in range(2): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n, r
    tf, r = timed(lambda: eng.decoder_forward(x, lat, g, targets=x, want_logits=False), 5)
    tt, r2 = timed(lambda: eng.train_step(x, lat, g), 5)
    ns = B * T
    out[dt] = dict(forward_ms=tf * 1e3, forward_samples_per_s=ns / tf, forward_hbm_frac=ns * (L * 4224 + 1537) / tf / 8e12,
                   forward_mfma_frac=ns * (L * 2162688 + 786432) / tf / 2.5e15, train_ms=tt * 1e3, train_samples_per_s=ns / tt,
                   train_mfma_frac=3 * ns * (L * 2162688 + 786432) / tt / 2.5e15, loss=float(r2["loss"]),
                   mem_GB=torch.cuda.max_memory_allocated() / 1e9)
print(json.dumps({"workload": f"C5 shard: {L} layers, R=G=S=512, {B}x{T}", **out}))
