import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
torch.manual_seed(0)
bad = 0
for cfg_name, cfg in (("C2", dict(bench.C2)), ("vqwae-dec", dict(layers=20, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=153, upsample_scales=[4, 4, 8, 5], cin_pad=0)),
                      ("C5-6layers", dict(layers=6, stacks=2, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=10, upsample_scales=[4, 4, 8, 5], cin_pad=0))):
    sd = O.make_state_dict(dict(cfg), salt=5, with_encoder=False)
    hop = 1
    for s in cfg["upsample_scales"]:
        hop *= s
    outs = {}
    for pairmode in ("0", "1"):
        os.environ["WAE_GLU_PAIR"] = pairmode
        eng = WaeEngine(Geometry.from_cfg(cfg), dtype="bf16")
        eng.load_state_dict(sd)
        for (B, F) in ((1, 3), (2, 7), (3, 13), (8, 25 if hop == 320 else 8)):
            T = F * hop
            x = torch.randint(0, 256, (B, T), generator=torch.Generator().manual_seed(B * 100 + F)).to(torch.int32).cuda()
            lat = torch.randn(B, 64, F, generator=torch.Generator().manual_seed(7)).cuda()
            g = torch.randint(0, cfg["n_speakers"], (B,), generator=torch.Generator().manual_seed(9)).cuda()
            for rep in range(6 if pairmode == "1" else 1):
                y = eng.decoder_forward(x, lat, g)["logits"]
                torch.cuda.synchronize()
                key = (B, F)
                if pairmode == "0":
                    outs[key] = y.clone()
                elif not torch.equal(y, outs[key]):
                    bad += 1
                    print("MISMATCH", cfg_name, key, rep, float((y - outs[key]).abs().max()))
    print(cfg_name, "done; shapes", list(outs))
print("mismatches:", bad)
