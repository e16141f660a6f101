#!/usr/bin/env python
"""Same-data timing of the backward pair launch (csrc/glu_bwd.hip: residual(l) + gate(l-1) [+ dc]) under variant builds of the library:
tools/time_pair.py <libA.so> [<libB.so> ...].  One C2 forward + backward with the PRODUCT library fills the buffers (z, dz, dx-hat, dS);
every library then runs the launch of a mid-stack layer 20 times on those same buffers (HIP events; the inputs never change)."""
import ctypes, math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import wae_oracle as O  # noqa: E402
from wavenet_autoencoders_amd import _lib as L, Geometry, backward as BW  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

conf = bench.CONFIGS[os.environ.get("PAIR_CONFIG", "c2")]
dev = torch.device("cuda:0")
eng = WaeEngine(Geometry.from_cfg(conf["cfg"]), dtype=conf["dtype"], device="cuda:0")
eng.load_state_dict(O.make_state_dict(dict(conf["cfg"]), salt=conf["salt"], with_encoder=conf["encoder"]))
eng.init_optimizer()
x, lat, gid = bench.synth_inputs(0, dev, conf)
xi = x.to(torch.int32)
eng.train_step(xi, lat, gid, lengths=None, lr=0.0)
torch.cuda.synchronize()
g = eng.g
B, T = xi.shape
fw, ws = eng._ws[(B, T, True)], eng._ws[("bwd", B, T)]
es = 2
Z2 = 2 * g.Hp
dzs = g.layers * Z2
ck = 64
us_off = (g.Rp // ck) * g.NP * 4 * 1024
cbytes = (Z2 // 64) * 8192
ngx = len(ws["gx"])
sig = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p] * 11 + [ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
for path in sys.argv[1:]:
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.wae_glu_bwd_fused_dc
    fn.argtypes, fn.restype = sig, ctypes.c_int32
    for l, four in [(int(a), f) for a in os.environ.get("PAIR_LAYERS", "13,4").split(",") for f in (0, 4)]:
        d = L.GluBwdDesc(eng.dt, B, T, g.Rp, g.Hp, g.Sp, g.k, g.dilations[l], math.sqrt(0.5))
        lp = l - 1
        args = [ctypes.byref(d), ctypes.c_void_p(ws["dz"].data_ptr() + l * Z2 * es), dzs, L.ptr(ws["gx"][(l + 1) % ngx]),
                L.ptr(ws["gx"][l % ngx]), L.ptr(ws["dskip"]), L.ptr(fw["z"][lp]), ctypes.c_void_p(ws["dz"].data_ptr() + lp * Z2 * es),
                ctypes.c_void_p((eng.w_bxf if hasattr(eng, "w_bxf") else eng.w_bx).data_ptr() + l * eng.n_bx * es),
                ctypes.c_void_p(eng.w_buo.data_ptr() + lp * eng.n_buo * es),
                ctypes.c_void_p(eng.w_bu.data_ptr() + lp * eng.n_bu * es + us_off),
                ctypes.c_void_p(eng.w_bc.data_ptr() + l * cbytes), L.ptr(ws["dc32"]), L.ptr(ws["dc"]), 1 | four, 0, None]
        dz_keep = ws["dz"].clone()
        for _ in range(3):
            assert fn(*args) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        ws["dz"].copy_(dz_keep)          # (a timing-only variant writes garbage into dz_{l-1}: the next library starts from the same data)
        print(f"{os.path.basename(path):28s} layer {l:2d} (dilation {g.dilations[l]:4d}) {'4 waves x 2 workgroups' if four else '8 waves, operands via LDS':26s}: "
              f"{e0.elapsed_time(e1) / 100 * 1e3:6.1f} us", flush=True)
