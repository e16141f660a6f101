#!/usr/bin/env python
"""Diagnostic: where wave 0 of every gemm_tn_static workgroup spends its shader-clock ticks per 16-row half-slab (s_memtime stamps
around the counted wait, the barrier and the MFMA / request body).  Needs a library built with -DWAE_TQ_STAMPS:
  tools/build_variant.sh tqstamps gemm_tn_static.hip "-DWAE_TQ_STAMPS"
usage: stamps_tq.py [libwae_<name>.so]     Never quote run times from this build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(lib):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    from wavenet_autoencoders_amd import _lib as L
    L.LIB_PATH = os.path.join(ROOT, "wavenet_autoencoders_amd", lib)
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
    assert hasattr(st, "stamps"), "the static stream table is not in use"
    for _ in range(3):
        st.launch()
    torch.cuda.synchronize()
    st.stamps = torch.zeros(st.nwg * 64, dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    st.launch()
    e1.record()
    torch.cuda.synchronize()
    s = st.stamps.cpu().numpy().reshape(-1, 16, 4)
    bid = np.arange(st.nwg)
    if st.nwg % 8 == 0:      # the kernel's team map: whole teams inside an XCD, the XCDs' leftover workgroups form the last teams
        per, ts = st.nwg >> 3, st.team_size
        xcd, slot = bid & 7, bid >> 3
        whole, rest = per // ts, per % ts
        member = np.where(slot < whole * ts, slot % ts, (xcd * rest + slot - whole * ts) % ts)
    else:
        member = bid % st.team_size
    ok = s[:, 0, 3] > 0
    life_t, life_w = s[:, 15, 0].astype(np.float64), s[:, 15, 1].astype(np.float64)
    mhz = life_t[ok] / (life_w[ok] / 100.0)
    print(f"launch {e0.elapsed_time(e1):.3f} ms; {int(ok.sum())} workgroups; shader clock {np.median(mhz):.0f} MHz; workgroup life median "
          f"{np.median(life_w[ok]) / 100.0:.1f} us (p10 {np.percentile(life_w[ok], 10) / 100.0:.1f}, p90 {np.percentile(life_w[ok], 90) / 100.0:.1f})")
    names = {0: "tap 0", 1: "tap 1", 2: "tap 2", 3: "cond", 4: "out+skip"}
    for m in range(st.team_size):
        sel = ok & (member == m)
        it = s[sel, 0, 3].astype(np.float64)
        print(f"  member {m} ({names.get(m, '?'):8s}): half-slabs {int(np.median(it)):5d}  life/half-slab {np.median(life_t[sel] / it):6.0f}  life us "
              f"{np.median(life_w[sel]) / 100.0:7.1f};  per half-slab and wave (wait / barrier / body):")
        row = []
        for w in range(12):
            itw = np.maximum(s[sel, w, 3].astype(np.float64), 1)
            row.append(f"w{w}: {np.median(s[sel, w, 0] / itw):4.0f}/{np.median(s[sel, w, 1] / itw):4.0f}/{np.median(s[sel, w, 2] / itw):4.0f}")
        print("      " + "  ".join(row[:6]))
        print("      " + "  ".join(row[6:]))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "libwae_tqstamps.so")
