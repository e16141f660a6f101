#!/usr/bin/env python
"""A/B of the work split of wae_gemm_tn_stream at C2: teams of one workgroup per job over equal slab ranges (WAE_TN_SHARES=teams)
against one workgroup per equal-time share of the (layer, job, slab) list (weighted; argv: values of the fixed cost fraction).
Stand-alone launch time of the one weight-gradient launch of a train step, and the gradients against the team split."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(mode, fixed):
    os.environ["WAE_TN_SHARES"] = mode
    os.environ["WAE_TN_SHARE_FIXED"] = fixed
    import torch
    sys.path.insert(0, ROOT)
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
    st = eng._ws[("bwd",) + tuple(xi.shape)]["stream"]
    eng.cbuf.zero_()
    st.launch()
    torch.cuda.synchronize()
    c = eng.cbuf.double()
    ts = []
    for i in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        st.launch()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    # a whole train step, too: the launch runs right after the backward sweep there (warm L2 / MALL, other clocks)
    for _ in range(3):
        eng.train_step(xi, lat, g, lengths=None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.train_step(xi, lat, g, lengths=None)
    e1.record()
    torch.cuda.synchronize()
    print(f"{mode:9s} fixed {fixed:5s}: launch ms min {min(ts[2:]):.3f} median {sorted(ts[2:])[3]:.3f}; train step {e0.elapsed_time(e1) / 10:.3f} ms; "
          f"segments {len(st.segs_dev) // 12}; sum|dW| {float(c.abs().sum()):.6e} sum dW^2 {float((c * c).sum()):.8e}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
    else:
        runs = [("teams", "0")] + [("weighted", f) for f in (sys.argv[1:] or ["0.3", "0.45", "0.6"])] + [("teams", "0")]
        for m, f in runs:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", m, f], check=False)
