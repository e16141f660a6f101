import sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry, backward as BW
from wavenet_autoencoders_amd.engine import WaeEngine
for k, T, ragged in [(3, 640, True), (2, 640, True), (4, 640, True), (1, 640, True)]:
  for dtype in ("bf16", "fp16"):
    print("case", k, T, ragged)
    cfg = dict(layers=6, stacks=2, R=128, G=192, S=128, O=256, Cc=64, Cg=32, k=k, n_speakers=7, upsample_scales=None)
    try:
        sd = O.make_state_dict(dict(cfg), 11, with_encoder=False)
        B = 2
        x = ((O.hash_fill((B, T), 91) * 0.5 + 0.5) * 256).long().clamp(0, 255)
        c = O.hash_fill((B, 64, T), 92, 1.1)
        g = torch.arange(B) % 7
        xin = torch.nn.functional.one_hot(x, 256).float().transpose(1, 2).contiguous()
        lengths = torch.tensor([T, T - 50] if ragged else [T, T])
        psd = {kk: v.clone().requires_grad_(True) for kk, v in sd.items() if kk.startswith("wavenet.")}
        y_ref = O.wavenet_forward(psd, dict(cfg), xin, c, g)
        loss_ref = O.masked_ce_loss(y_ref, x.unsqueeze(-1), lengths)
        loss_ref.backward()
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
        eng.load_state_dict(sd)
        out = eng.decoder_forward(x.cuda(), c.cuda(), g.cuda(), targets=x.cuda(), lengths=lengths.cuda(), train=True, c_is_upsampled=True)
        BW.decoder_backward(eng, x.cuda(), x.cuda(), lengths, g.cuda())
        grads = BW.finish_grads(eng)
        torch.cuda.synchronize()
        el = float((out["logits"].cpu() - y_ref.detach()).abs().max() / y_ref.detach().abs().max())
        worst = 0.0
        for kk, v in psd.items():
            gref = v.grad if v.grad is not None else torch.zeros_like(v)
            got = grads[eng.lay.off(kk):eng.lay.off(kk) + eng.lay.numel(kk)].view(eng.lay.shapes[kk]).cpu()
            e = float((got - gref).abs().max()) / max(float(gref.abs().max()), 1e-6)
            if e > worst: worst, wk = e, (kk, float(gref.abs().max()))
        # AR teacher-forced logits
        ea = 0.0
        if worst > 1e-4: print("FAIL", end=" ")
        print(f"k={k} T={T} {dtype}: logits rel err {el:.2e}, loss {float(out['loss']):.5f} vs {float(loss_ref):.5f}, worst grad rel err {worst:.2e} at {wk}, AR rel err {ea:.2e}")
    except Exception as e:
        if worst > 1e-4: print("FAIL", end=" ")
        print(f"k={k} T={T} {dtype}: FAILED {type(e).__name__}: {str(e)[:300]}")
