#!/usr/bin/env python
"""Does running a layer sweep as TWO half-batch chains on two streams, one started `delay` microseconds late, beat one chain of
full-batch launches?  (Every workgroup of a launch stores its z / u / x' at the same time: ~20 us of a 55-us forward launch in which
nothing computes.  A single launch cannot be de-phased against itself -- it ends with its late half -- but two chains of launches can.)

    python tools/two_chain.py [--geom c2|c3] [--layers 24] [--delays 0,10,20,30] [--kind fwd|bwd]

Timing only over synthetic operands; the kernels are the product's, called through the C ABI with B / 2 clips and offset pointers."""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavenet_autoencoders_amd import Geometry  # noqa: E402
from wavenet_autoencoders_amd import _lib as L  # noqa: E402
from wavenet_autoencoders_amd.engine import WaeEngine  # noqa: E402

GEOMS = {
    "c2": (dict(layers=2, stacks=1, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None), 8, 8000),
    "c3": (dict(layers=2, stacks=1, R=256, G=256, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=4, upsample_scales=None), 8, 5120),
    "c5": (dict(layers=2, stacks=1, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=4, upsample_scales=None), 16, 5120),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geom", default="c2")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--delays", default="0,10,20,30,40")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--chains", default="2")
    ap.add_argument("--unique", type=int, default=0, help="1: every layer has its own copy of the weights / x / u column block (as in a train step)")
    a = ap.parse_args()
    cfg, B, T = GEOMS[a.geom]
    eng = WaeEngine(Geometry.from_cfg(cfg), dtype="fp16" if a.geom == "c5" else "bf16")
    torch.manual_seed(0)
    eng.params.normal_(0, 0.05)
    eng.prepare_weights()
    g, lib, es = eng.g, eng.lib, 2
    dev = eng.device
    xs = [(torch.randn(B, T, g.Rp, device=dev) * 0.5).to(eng.tdtype) for _ in range(2)]
    c = (torch.randn(B, T, g.Ccp, device=dev) * 0.5).to(eng.tdtype)
    zb = torch.randn(B, 2 * g.Hp, device=dev) * 0.1
    NL = a.layers
    u = torch.zeros(B, T, g.Hp, device=dev).to(eng.tdtype)
    z = [torch.zeros(B, T, 2 * g.Hp, device=dev).to(eng.tdtype) for _ in range(NL)]
    dil = [1 << (i % 6) for i in range(NL)]
    if a.unique:
        wl = [eng.w_glu.clone() for _ in range(NL)]
        xs = [(torch.randn(B, T, g.Rp, device=dev) * 0.5).to(eng.tdtype) for _ in range(NL + 1)]
        u = torch.zeros(B, T, g.Hp * NL, device=dev).to(eng.tdtype)
    s0 = torch.cuda.Stream(dev)
    clock_mhz = 100.0      # s_memrealtime / torch.cuda._sleep count cycles of a ~100 MHz..2 GHz clock: calibrated below

    def launch(i, b0, nb, st):
        d = L.GluDesc(eng.dt, nb, T, g.Rp, g.Ccp, g.Hp, g.k, dil[i], L.GLU_SAVE_Z)
        xin, xout = (xs[i], xs[i + 1]) if a.unique else (xs[i % 2], xs[(i + 1) % 2])
        ku = g.Hp * NL if a.unique else g.Hp
        wp = wl[i] if a.unique else eng.w_glu
        L.check(lib.wae_glu_layer_fwd(ctypes.byref(d), ctypes.c_void_p(xin.data_ptr() + b0 * T * g.Rp * es),
                                      ctypes.c_void_p(xout.data_ptr() + b0 * T * g.Rp * es),
                                      ctypes.c_void_p(c.data_ptr() + b0 * T * g.Ccp * es),
                                      ctypes.c_void_p(u.data_ptr() + (b0 * T * ku + (i * g.Hp if a.unique else 0)) * es), ku,
                                      ctypes.c_void_p(zb.data_ptr() + b0 * 2 * g.Hp * 4), 2 * g.Hp,
                                      ctypes.c_void_p(z[i].data_ptr() + b0 * T * 2 * g.Hp * es), L.ptr(wp), L.ptr(eng.b_glu),
                                      ctypes.c_void_p(st.cuda_stream)))

    def one_chain():
        for i in range(NL):
            launch(i, 0, B, s0)

    side = [torch.cuda.Stream(dev) for _ in range(7)]

    def n_chains(nc, sleep_cycles):
        """chain k = clips [B k / nc, B (k + 1) / nc) on its own stream, started k * delay late"""
        sts = [s0] + side[:nc - 1]
        cuts = [B * k // nc for k in range(nc + 1)]
        ev = torch.cuda.Event()
        ev.record(s0)
        for k in range(1, nc):
            sts[k].wait_event(ev)
            if sleep_cycles > 0:
                with torch.cuda.stream(sts[k]):
                    torch.cuda._sleep(int(sleep_cycles * k))
        for i in range(NL):
            for k in range(nc):
                launch(i, cuts[k], cuts[k + 1] - cuts[k], sts[k])
        for k in range(1, nc):
            e2 = torch.cuda.Event()
            e2.record(sts[k])
            s0.wait_event(e2)

    def timed(fn):
        best = []
        for _ in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record(s0)
            fn()
            e1.record(s0)
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) * 1e3)
        best.sort()
        return best[0], best[len(best) // 2]

    # calibrate torch.cuda._sleep: cycles per microsecond
    def sleeper(n):
        def f():
            with torch.cuda.stream(s0):
                torch.cuda._sleep(n)
        return f
    t1 = timed(sleeper(100000))[0]
    t2 = timed(sleeper(1100000))[0]
    cyc_per_us = 1000000 / max(t2 - t1, 1e-3)
    print(f"torch.cuda._sleep: {cyc_per_us:.1f} cycles per us")
    one_chain()
    lo, med = timed(one_chain)
    print(f"{a.geom}: one chain of {NL} full-batch launches: {lo:8.1f} us (median {med:8.1f}) = {lo / NL:5.1f} us per layer")
    for nc in [int(v) for v in a.chains.split(",")]:
        for dl in [float(v) for v in a.delays.split(",")]:
            fn = lambda: n_chains(nc, dl * cyc_per_us)
            fn()
            lo, med = timed(fn)
            print(f"{a.geom}: {nc} chains, each {dl:5.1f} us behind the previous: {lo:8.1f} us (median {med:8.1f}) = {lo / NL:5.1f} us per layer")


if __name__ == "__main__":
    main()
