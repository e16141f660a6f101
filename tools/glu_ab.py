#!/usr/bin/env python
"""Same-box A/B of builds of the fused layer kernel at one geometry: glu_ab.py [--geom c2|c3|c5] [--rounds N] lib.so [lib.so ...]
Per library (each in its own process): bitwise comparison of the static-schedule kernel against the generic one (WAE_GLU_GENERIC)
on x', u and z, then microseconds per launch (HIP events over 20 launches) for inference / training launches at two dilations."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GEOMS = {
    "c2": (dict(layers=2, stacks=1, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None), 8, 8000),
    "c3": (dict(layers=2, stacks=1, R=256, G=256, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None), 8, 5120),
    "c5": (dict(layers=2, stacks=1, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=4, upsample_scales=None), 16, 5120),
}


def child(lib, geom, dtype):
    import torch
    sys.path.insert(0, ROOT)
    from wavenet_autoencoders_amd import _lib as L
    L.LIB_PATH = os.path.abspath(lib)
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    cfg, B, T = GEOMS[geom]
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype=dtype)
    torch.manual_seed(0)
    eng.params.normal_(0, 0.05)
    eng.prepare_weights()
    g = eng.g
    x = (torch.randn(B, T, g.Rp, device="cuda") * 0.5).to(eng.tdtype)
    c = (torch.randn(B, T, g.Ccp, device="cuda") * 0.5).to(eng.tdtype)
    zb = torch.randn(B, 2 * g.Hp, device="cuda") * 0.1
    st = eng.stream()

    def run(xo, u, z, flags, d):
        desc = L.GluDesc(eng.dt, B, T, g.Rp, g.Ccp, g.Hp, g.k, d, flags)
        L.check(eng.lib.wae_glu_layer_fwd(ctypes.byref(desc), L.ptr(x), L.ptr(xo), L.ptr(c), L.ptr(u), g.Hp, L.ptr(zb), 2 * g.Hp,
                                          L.ptr(z) if flags & L.GLU_SAVE_Z else None, L.ptr(eng.w_glu), L.ptr(eng.b_glu), st))

    name = os.path.basename(lib)
    bad = 0
    for d in (1, 2, 64, 512, 2048):
        outs = []
        for gen in (L.GLU_GENERIC, 0):
            xo, u, z = torch.zeros_like(x), torch.zeros(B, T, g.Hp, device="cuda").to(eng.tdtype), torch.zeros(B, T, 2 * g.Hp, device="cuda").to(eng.tdtype)
            run(xo, u, z, L.GLU_SAVE_Z | gen, d)
            torch.cuda.synchronize()
            outs.append((xo, u, z))
        eq = [bool(torch.equal(a.view(torch.int16), b_.view(torch.int16))) for a, b_ in zip(*outs)]
        if not all(eq):
            bad += 1
            diffs = [(a.float() - b_.float()).abs().max().item() for a, b_ in zip(*outs)]
            print(f"{name}: d={d}: static vs generic x'/u/z equal: {eq}  max|diff| {diffs}", flush=True)
    print(f"{name}: bitwise vs generic over 5 dilations: {'EQUAL' if bad == 0 else f'{bad} MISMATCHES'}", flush=True)
    xo, u, z = torch.zeros_like(x), torch.zeros(B, T, g.Hp, device="cuda").to(eng.tdtype), torch.zeros(B, T, 2 * g.Hp, device="cuda").to(eng.tdtype)
    res = []
    for gen in ((0, L.GLU_GENERIC | L.GLU_PAIR) if os.environ.get("AB_GENERIC") else (0,)):
        for flags in (0, L.GLU_SAVE_Z):
            for d in (1, 64):
                for _ in range(5):
                    run(xo, u, z, flags | gen, d)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    run(xo, u, z, flags | gen, d)
                e1.record()
                torch.cuda.synchronize()
                res.append(f"{'gen ' if gen else ''}{'z' if flags else 'inf'} d={d}: {e0.elapsed_time(e1) / 30 * 1e3:6.1f}")
    print(f"{name:34s} us/launch  " + "  ".join(res), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        args = sys.argv[1:]
        geom, rounds, dtype = "c2", 2, "bf16"
        while args and args[0].startswith("--"):
            k = args.pop(0)
            v = args.pop(0)
            if k == "--geom":
                geom = v
            elif k == "--rounds":
                rounds = int(v)
            elif k == "--dtype":
                dtype = v
        for _ in range(rounds):
            for lib in args:
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, geom, dtype], check=False)
