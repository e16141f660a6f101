#!/usr/bin/env python
"""Diagnostic: where wave 0 of every gemm_tn_stream workgroup spends its shader-clock ticks (s_memtime stamps around the phases of
the slab loop; builds made by tools/ablate_ts.sh "stamps:-DWAE_TS_STAMPS" and e.g. "stamps1:-DWAE_TS_STAMPS -DWAE_TS_ABLATE=1").
usage: stamps_ts.py [tag ...]     Never quote run times from these builds."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(tag):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from wavenet_autoencoders_amd import _lib as L
    L.LIB_PATH = os.path.join(ROOT, "wavenet_autoencoders_amd", f"libwae_tsabl{tag}.so")
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
    st = None
    # the stamp array must be in place before the first launch of this build
    from wavenet_autoencoders_amd import backward as BW
    orig = BW.StreamTable.finalize

    def fin(self):
        orig(self)
        self.pace = torch.zeros(self.nwg * 8, dtype=torch.int64, device=dev)
        return self
    BW.StreamTable.finalize = fin
    eng.train_step(xi, lat, g, lengths=None)
    torch.cuda.synchronize()
    st = eng._ws[("bwd",) + tuple(xi.shape)]["stream"]
    for _ in range(3):
        st.launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st.launch()
    e1.record()
    torch.cuda.synchronize()
    s = st.pace.cpu().numpy().reshape(-1, 8)
    s = s[s[:, 6] > 0]
    it = s[:, 6].astype(np.float64)
    mhz = s[:, 0] / (s[:, 1] / 100.0)
    print(f"[{tag}] launch {e0.elapsed_time(e1):.3f} ms; {len(s)} workgroups, slabs per workgroup median {int(np.median(it))}, "
          f"segments {int(np.median(s[:, 7]))}; shader clock {np.median(mhz):.0f} MHz (p10 {np.percentile(mhz, 10):.0f}, p90 "
          f"{np.percentile(mhz, 90):.0f}); workgroup life median {np.median(s[:, 1]) / 100.0:.1f} us")
    for i, nme in ((2, "wait for the slab (vmcnt)"), (3, "zero-fill + barrier"), (4, "issue next slab + ones column"),
                   (5, "LDS reads + MFMA + cursors")):
        per = s[:, i] / it
        print(f"    {nme:32s} ticks per slab: median {np.median(per):7.0f}  p10 {np.percentile(per, 10):7.0f}  p90 "
              f"{np.percentile(per, 90):7.0f}")
    tot = s[:, 2:6].sum(axis=1) / it
    print(f"    {'sum':32s} ticks per slab: median {np.median(tot):7.0f};  whole life / slabs {np.median(s[:, 0] / it):7.0f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for b in (sys.argv[1:] or ["stamps"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", b], check=False)
