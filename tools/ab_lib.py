#!/usr/bin/env python
"""Same-box A/B of two builds of the library over the C2 inference forward and train step: ab_lib.py <libA.so> <libB.so> [rounds].
Each measurement runs in its own process (a ctypes library cannot be swapped once loaded), builds alternate."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(lib):
    import torch
    sys.path.insert(0, ROOT)
    from wavenet_autoencoders_amd import _lib as L
    L.LIB_PATH = os.path.abspath(lib)
    import bench
    from oracle import wae_oracle as O
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    dev = torch.device("cuda:0")
    x, lat, g = bench.synth_inputs(0, dev)
    xi = x.to(torch.int32)
    eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype=os.environ.get("AB_DTYPE", "bf16"), device="cuda:0")
    eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
    eng.init_optimizer()

    def timed(fn, n):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    fwd = timed(lambda: eng.decoder_forward(xi, lat, g, targets=xi, want_logits=False), 20)
    trn = timed(lambda: eng.train_step(xi, lat, g, lengths=None), 20)
    print(f"{os.path.basename(lib):28s} forward {fwd:.3f} ms   train step {trn:.3f} ms", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        a, b = sys.argv[1], sys.argv[2]
        for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
            for lib in (a, b):
                subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib], check=False)
