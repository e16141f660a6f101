"""Which torch operators launch kernels inside one C2 train step (everything else in the step is the library's own launches)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import wae_oracle as O
from wavenet_autoencoders_amd import Geometry
from wavenet_autoencoders_amd.engine import WaeEngine
dev = torch.device("cuda:0")
x, lat, g = bench.synth_inputs(0, dev)
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16", device="cuda:0")
eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
eng.init_optimizer()
for _ in range(3):
    eng.train_step(x, lat, g, lengths=None)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    eng.train_step(x, lat, g, lengths=None)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.device_time_total > 0 and e.key.startswith("aten::")]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:25]:
    st = [s for s in e.stack if "wavenet_autoencoders_amd" in s or "bench" in s]
    print(f"{e.key:28s} x{e.count:3d} {e.device_time_total:8.1f} us  {st[0] if st else ''}")
