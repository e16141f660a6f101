#!/usr/bin/env python
"""Timing-only ablation of the one weight-gradient launch (gemm_tn_stream_kernel) of a C2 train step.
usage: ablate_ts.py [bits ...]   (libraries built by tools/ablate_ts.sh; 0 = the product library)
Each library runs in its own process (a ctypes library cannot be swapped once loaded)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {0: "full", 1: "no operand DMA", 2: "no LDS reads, no MFMA", 4: "no LDS reads", 8: "no flush", 16: "no MFMA", 5: "MFMA + flush",
         17: "LDS reads + flush", 3: "barriers + flush only", 20: "DMA + barriers only (asm waits kept)"}


def child(bits):
    bits = int(bits) if bits.isdigit() else bits
    import torch
    sys.path.insert(0, ROOT)
    from wavenet_autoencoders_amd import _lib as L
    if bits != 0:
        L.LIB_PATH = os.path.join(ROOT, "wavenet_autoencoders_amd", f"libwae_tsabl{bits}.so")
    import bench
    from oracle import wae_oracle as O
    from wavenet_autoencoders_amd import Geometry
    from wavenet_autoencoders_amd.engine import WaeEngine
    dev = torch.device("cuda:0")
    x, lat, g = bench.synth_inputs(0, dev)
    xi = x.to(torch.int32)
    eng = WaeEngine(Geometry.from_cfg(bench.C2), dtype="bf16", device="cuda:0")
    eng.load_state_dict(O.make_state_dict(dict(bench.C2), salt=5, with_encoder=False))
    eng.init_optimizer()
    eng.train_step(xi, lat, g, lengths=None)
    torch.cuda.synchronize()
    ws = eng._ws[("bwd",) + tuple(xi.shape)]
    st = ws["stream"]
    if os.environ.get("WAE_TS_CONTIG", "0") != "0":
        # timing only: every operand as its own contiguous (B*T, cols) array (layer l of dz / u at offset l * B*T*cols of the same
        # allocation) instead of a column slice of the (B, T, L*cols) arrays; the gradients are garbage
        g = eng.g
        Bn, Tn = xi.shape
        es = 2
        dz0, u0 = ws["dz"].data_ptr(), eng._ws[(Bn, Tn, True)]["u"].data_ptr()
        Z2 = 2 * g.Hp
        for l, grp in enumerate(st.groups):
            for jb in grp:
                if jb.p_stride == g.layers * Z2:
                    jb.P = dz0 + l * Bn * Tn * Z2 * es
                    jb.p_stride = Z2
                if jb.q_stride == g.Ku and g.Ku != g.Hp:
                    jb.Q = u0 + l * Bn * Tn * g.Hp * es
                    jb.q_stride = g.Hp
        st.finalize()
    eng.cbuf.zero_()
    st.launch()
    torch.cuda.synchronize()
    chk = float(eng.cbuf.double().abs().sum())
    ts = []
    for i in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st.launch()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    tag = " contiguous operands" if os.environ.get("WAE_TS_CONTIG", "0") != "0" else ""
    print(f"{str(NAMES.get(bits, bits)) + tag:44s} launch ms min {min(ts[2:]):.3f} median {sorted(ts[2:])[2]:.3f}   sum|dW| {chk:.6e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for b in (sys.argv[1:] or ["0"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", b], check=False)
