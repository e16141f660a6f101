#!/usr/bin/env python
"""Timing-only ablation of the two per-layer wae_gemm_tm launches of the backward pass at the C2 shape.
usage: ablate_tm.py [lib suffix ...]   (libraries built by tools/ablate_tm.sh; '' = the product library)
Each library runs in its own process (a ctypes library cannot be swapped once loaded)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "full", 1: "no operand loads", 2: "no weight DMA", 4: "no MFMA", 8: "no epilogue", 3: "MFMA + epilogue only",
         7: "epilogue only", 12: "loads + DMA only", 11: "MFMA only", 15: "empty"}


def child(bits):
    import torch
    sys.path.insert(0, ROOT)
    from wavenet_autoencoders_amd import _lib as L
    if bits:
        L.LIB_PATH = os.path.join(ROOT, "wavenet_autoencoders_amd", f"libwae_tmabl{bits}.so")
    from oracle import wae_oracle as O
    from wavenet_autoencoders_amd import Geometry, backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    C2 = dict(layers=24, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=153, upsample_scales=[4, 4, 4, 5],
              cin_pad=0)
    B, T = 8, 8000
    eng = WaeEngine(Geometry.from_cfg(C2), dtype="bf16")
    eng.load_state_dict(O.make_state_dict(dict(C2), salt=5, with_encoder=False))
    x = torch.randint(0, 256, (B, T), device="cuda").to(torch.int32)
    lat = torch.randn(B, 64, T // 320, device="cuda")
    gid = torch.randint(0, 153, (B,), device="cuda")
    eng.train_step(x, lat, gid)
    l = 5

    def timeit(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    ku, kx = BW._debug_kernels(eng, B, T, l)
    print(f"{NAMES.get(bits, bits):22s} du/dz {timeit(ku):7.1f} us   dx {timeit(kx):7.1f} us", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        for b in (sys.argv[1:] or ["0"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", b], check=False)
