import os, sys, torch
sys.path.insert(0, "/root/repo")
import bench
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
dev = torch.device("cuda:0")
x, lat, g = bench.synth_inputs(0, dev)
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16", device="cuda:0")
eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
eng.prepare_weights()
for _ in range(3):
    eng.decoder_forward(x, lat, g, targets=x, lengths=None, want_logits=False, train=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    eng.decoder_forward(x, lat, g, targets=x, lengths=None, want_logits=False, train=False)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("WAE_LIB_PATH", "default"), "forward ms", e0.elapsed_time(e1) / 20)
