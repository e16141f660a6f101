"""Probe (GPU): kernel sizes 1..4 through the engine against the oracle -- logits, loss, every parameter gradient, in fp32 / bf16 / fp16.
Usage: python tools/micro/k_probe.py [T ...]   (default T = 640; at some lengths one ReLU pre-activation of this closed-form model sits within
rounding of zero, and the flipped mask moves a head-bias gradient by ~1/sqrt(T): fp32 "errors" of 1e-2 that are not errors)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry, backward as BW
from wavenet_autoencoders_amd.engine import WaeEngine

for T in [int(a) for a in sys.argv[1:]] or [640]:
    for k in (1, 2, 3, 4):
        cfg = dict(layers=6, stacks=2, R=128, G=192, S=128, O=256, Cc=64, Cg=32, k=k, n_speakers=7, upsample_scales=None)
        sd = O.make_state_dict(dict(cfg), 11, with_encoder=False)
        B = 2
        x = ((O.hash_fill((B, T), 91) * 0.5 + 0.5) * 256).long().clamp(0, 255)
        c = O.hash_fill((B, 64, T), 92, 1.1)
        g = torch.arange(B) % 7
        xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
        lengths = torch.tensor([T, T - 50])
        psd = {kk: v.clone().requires_grad_(True) for kk, v in sd.items() if kk.startswith("wavenet.")}
        y_ref = O.wavenet_forward(psd, dict(cfg), xin, c, g)
        loss_ref = O.masked_ce_loss(y_ref, x.unsqueeze(-1), lengths)
        loss_ref.backward()
        for dtype in ("fp32", "bf16", "fp16"):
            eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
            eng.load_state_dict(sd)
            out = eng.decoder_forward(x.cuda(), c.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), train=True, c_is_upsampled=True)
            BW.decoder_backward(eng, x.cuda(), x.cuda(), lengths, g.cuda())
            grads = BW.finish_grads(eng)
            torch.cuda.synchronize()
            el = float((out["logits"].cpu() - y_ref.detach()).abs().max() / y_ref.detach().abs().max())
            worst, wk = 0.0, None
            for kk, v in psd.items():
                gref = v.grad if v.grad is not None else torch.zeros_like(v)
                got = grads[eng.lay.off(kk):eng.lay.off(kk) + eng.lay.numel(kk)].view(eng.lay.shapes[kk]).cpu()
                e = float((got - gref).abs().max()) / max(float(gref.abs().max()), 1e-6)
                if e > worst:
                    worst, wk = e, kk
            print(f"T={T} k={k} {dtype}: logits rel err {el:.2e}, loss {float(out['loss']):.5f} vs {float(loss_ref):.5f}, "
                  f"worst gradient rel err {worst:.2e} at {wk}")
