import sys, os, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_wide as W
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
sd, x, xin, c, g = W._wide_inputs(T=640)
T = x.shape[1]
eng = WaeEngine(Geometry.from_cfg(W.WIDE), dtype="fp32")
eng.load_state_dict(sd)
res = W._grad_check(eng, W.WIDE, sd, x, xin, c, g, torch.tensor([T, T - 137]), dict(layers=4, stacks=2, cin_pad=0), ref_dtype=torch.float64)
for k, v in sorted(res.items(), key=lambda kv: -kv[1][0] / max(kv[1][1], 1e-12)):
    print("%-50s err %.3e ref %.3e rel %.3e" % (k, v[0], v[1], v[0] / max(v[1], 1e-12)))
print("---- direct check of the head backward buffers")
B = 2
fw = eng._ws[(B, T, True)]; ws = eng._ws[("bwd", B, T)]
G = eng.g
W3 = eng.eff[eng.lay.off("wavenet.last_conv_layers.3.weight_v"):][:G.O * G.S].view(G.O, G.S)
W1 = eng.eff[eng.lay.off("wavenet.last_conv_layers.1.weight_v"):][:G.S * G.S].view(G.S, G.S)
dy, h1, h0 = ws["dy"].float(), fw["h1"].float(), fw["h0"].float()
dh1_ref = (dy[..., :G.O] @ W3) * (h1 > 0)
err = (ws["dh1"].float() - dh1_ref).abs()
print("dh1 err", float(err.max()), "ref", float(dh1_ref.abs().max()), "per-128-col-block max", [float(err[..., i:i + 128].max()) for i in range(0, 512, 128)])
print("   per 128-row time block of clip 0:", [float(err[0, i:i + 128].max()) for i in range(0, T, 128)])
dsk_ref = (ws["dh1"].float() @ W1) * (h0 > 0) * (1.0 / G.layers) ** 0.5
err = (ws["dskip"].float() - dsk_ref).abs()
print("dskip err", float(err.max()), "ref", float(dsk_ref.abs().max()), [float(err[..., i:i + 128].max()) for i in range(0, 512, 128)])
u = fw["u"].float()
print("---- weight-gradient tiles")
sm = eng.sm
c1h = eng.cview["c1h"].view(G.Sp, sm["ldh"])
dh1 = ws["dh1"].float().reshape(-1, G.Sp).double(); h0d = h0.reshape(-1, G.Sp).double()
ref = (dh1.t() @ h0d).float()
err = (c1h[:, :G.Sp] - ref).abs()
print("dW1h err", float(err.max()), "ref", float(ref.abs().max()), "blocks", [[round(float(err[i:i+128, j:j+128].max() / ref.abs().max()), 4) for j in range(0, 512, 128)] for i in range(0, 512, 128)])
bsum = c1h[:, G.Sp:G.Sp + 128].sum(1)
bref = dh1.sum(0).float()
print("db1 err", float((bsum - bref).abs().max()), float(bref.abs().max()))
print("ones cols nonzero:", (c1h[:, G.Sp:].abs().sum(0) > 0).nonzero().flatten().tolist())
tt = ws["tt_head"]; print("tiles", tt.n, "splits", tt.splits)
print("---- forward intermediates vs oracle")
import math, torch.nn.functional as F
from oracle import wae_oracle as O
with torch.no_grad():
    y, _, inter = O.wavenet_forward(sd, dict(layers=4, stacks=2, upsample_scales=None, cin_pad=0), xin, c, g, return_intermediates=True)
    skips = sum(s for _, s in inter) * math.sqrt(1.0 / 4)
    h0o = F.relu(skips)
    p1 = F.conv1d(h0o, O.eff_weight(sd, "wavenet.last_conv_layers.1"), sd["wavenet.last_conv_layers.1.bias"])
    h1o = F.relu(p1)
h0g = fw["h0"].float().cpu().transpose(1, 2); h1g = fw["h1"].float().cpu().transpose(1, 2)
print("h0 err", float((h0g - h0o).abs().max()), float(h0o.abs().max()), "mask flips", int(((h0g > 0) != (h0o > 0)).sum()), "of", h0o.numel())
print("h1 err", float((h1g - h1o).abs().max()), float(h1o.abs().max()), "mask flips", int(((h1g > 0) != (h1o > 0)).sum()))
print("|pre1| < 1e-5:", int((p1.abs() < 1e-5).sum()), " skips |.|<1e-5:", int((skips.abs() < 1e-5).sum()))
for i, (hh, ss) in enumerate(inter):
    xg = fw["x"][i + 1].float().cpu().transpose(1, 2)[:, :512] if i + 1 < 4 else None
    if xg is not None: print("layer", i, "x' err", float((xg - hh).abs().max()), float(hh.abs().max()))
