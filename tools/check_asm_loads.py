#!/usr/bin/env python
"""The inline-asm operand requests (glu_fwd, gemm_tm, bf16) against the same kernels with compiler-managed loads: identical
arithmetic in identical order, so logits and gradients must agree BITWISE (gradients: up to the atomics' arrival order).  A
fragment register spilled or copied between its request and the counted wait would show up here.
Build the comparison library first:  make -C wavenet_autoencoders_amd/csrc clean && make -C wavenet_autoencoders_amd/csrc -j4 \\
    EXTRA=-DWAE_GLU_PLAIN_LOADS && cp wavenet_autoencoders_amd/libwae_hip.so build/libwae_plain.so ; then rebuild the product.
usage: check_asm_loads.py            (runs both libraries in child processes and compares)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = {
    "C2": dict(layers=6, stacks=2, R=256, G=368, S=256, O=256, Cc=64, Cg=64, k=3, n_speakers=8, upsample_scales=None, cin_pad=0),
    "C1": dict(layers=6, stacks=2, R=256, G=256, S=256, O=256, Cc=64, Cg=32, k=3, n_speakers=8, upsample_scales=None, cin_pad=0),
    "C5": dict(layers=4, stacks=2, R=512, G=512, S=512, O=256, Cc=64, Cg=32, k=3, n_speakers=8, upsample_scales=None, cin_pad=0),
    "H96": dict(layers=4, stacks=2, R=128, G=192, S=128, O=256, Cc=64, Cg=32, k=3, n_speakers=8, upsample_scales=None, cin_pad=0),
}


def child(libpath, out):
    import torch
    sys.path.insert(0, ROOT)
    from wavenet_autoencoders_amd import _lib as L
    L.LIB_PATH = libpath
    from oracle import wae_oracle as O
    from wavenet_autoencoders_amd import Geometry, backward as BW
    from wavenet_autoencoders_amd.engine import WaeEngine
    res = {}
    for name, cfg in CONFIGS.items():
        for nw in (8, 4):
            sd = O.make_state_dict(dict(cfg), salt=9, with_encoder=False)
            eng = WaeEngine(Geometry.from_cfg(cfg), dtype="bf16")
            eng.glu_flags |= L.GLU_WAVES4 if nw == 4 else 0          # wae_glu_desc.flags of every layer launch
            eng.load_state_dict(sd)
            B, T = 3, 2000
            x = ((O.hash_fill((B, T), 41) * 0.5 + 0.5) * 256).long().clamp(0, 255).cuda()
            c = O.hash_fill((B, cfg["Cc"], T), 42, 1.2).cuda()
            g = torch.tensor([1, 5, 2]).cuda()
            o = eng.decoder_forward(x, c, g, targets=x, train=True, c_is_upsampled=True)
            res[f"{name}/nw{nw}/logits"] = o["logits"].cpu()
            BW.decoder_backward(eng, x, x, None, g)
            res[f"{name}/nw{nw}/dz"] = eng._ws[("bwd", B, T)]["dz"].float().cpu()
            res[f"{name}/nw{nw}/gx0"] = eng._ws[("bwd", B, T)]["gx"][0].float().cpu()
            torch.cuda.synchronize()
            del eng
    torch.save(res, out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
    else:
        import torch
        libs = {"asm": os.path.join(ROOT, "wavenet_autoencoders_amd", "libwae_hip.so"), "plain": os.path.join(ROOT, "build", "libwae_plain.so")}
        for k, v in libs.items():
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v, f"/tmp/asmcheck_{k}.pt"], check=True)
        a, b = torch.load("/tmp/asmcheck_asm.pt"), torch.load("/tmp/asmcheck_plain.pt")
        bad = 0
        for k in a:
            same = torch.equal(a[k], b[k])
            d = float((a[k] - b[k]).abs().max())
            print(f"{k:24s} bitwise equal: {same}   max |diff| {d:.3e}")
            bad += 0 if same else 1
        print("ALL BITWISE EQUAL" if bad == 0 else f"{bad} tensors differ")
        sys.exit(1 if bad else 0)
