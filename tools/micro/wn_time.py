"""Times wae_weight_norm_fwd / _bwd on the C2 arena, warm (back to back) and cold (512 MB written in between)."""
import os, sys, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from wavenet_autoencoders_amd import Geometry, _lib as L
from wavenet_autoencoders_amd.engine import WaeEngine
from oracle import wae_oracle as O
eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16", device="cuda:0")
eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
lib, lay = eng.lib, eng.lay
st = eng.stream()
d_eff = torch.randn_like(eng.params)
grads = torch.zeros_like(eng.params)
junk = torch.zeros(128 << 20, dtype=torch.float32, device="cuda:0")
def fwd():
    L.check(lib.wae_weight_norm_fwd(L.ptr(eng.params), L.ptr(eng.eff), lay.total, L.ptr(eng.wn_v), L.ptr(eng.wn_g), L.ptr(eng.wn_c), len(lay.wn_cols), st), "f")
def bwd():
    L.check(lib.wae_weight_norm_bwd(L.ptr(eng.params), L.ptr(d_eff), L.ptr(grads), lay.total, L.ptr(eng.wn_v), L.ptr(eng.wn_g), L.ptr(eng.wn_c), len(lay.wn_cols), st), "b")
def timeit(fn, cold):
    ts = []
    for _ in range(8):
        if cold:
            junk.add_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    print(name, "warm %.1f us  cold %.1f us" % (timeit(fn, False), timeit(fn, True)))
