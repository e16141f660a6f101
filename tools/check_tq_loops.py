#!/usr/bin/env python
"""Static check of csrc/gemm_tn_static.hip's hot loops in the built ISA (hipcc -S): every innermost block that carries the barrier and
the LDS-DMA requests must be free of scratch traffic, SGPR spill lanes, scalar / global loads and compiler-made vmcnt(0) drains
(anything of that kind stalls or drains the request ring once per 16 time rows).
usage: check_tq_loops.py [listing.s]   (default: builds the listing from the source)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = "/tmp/wae_tq.s"
        src = os.path.join(ROOT, "wavenet_autoencoders_amd", "csrc", "gemm_tn_static.hip")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-inline-asm", "-Wno-array-bounds",
                               "-S", "--cuda-device-only", src, "-o", path], stderr=subprocess.DEVNULL)
    lines = open(path).read().split("\n")
    blocks, cur, kern = [], None, "?"
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern = m.group(1)
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            cur = [kern, m.group(1), i, []]
            blocks.append(cur)
        elif cur is not None:
            cur[3].append(l)
    bad = 0
    for kern, name, i, body in blocks:
        txt = "\n".join(body)
        if "s_barrier" not in txt or "buffer_load_dwordx4" not in txt or ("s_cbranch" not in txt):
            continue
        ins = [b for b in body if b.strip() and not b.strip().startswith(";")]
        c = lambda pat: len(re.findall(pat, txt))  # noqa: E731
        row = dict(instr=len(ins), mfma=c(r"v_mfma"), dma=c(r"buffer_load_dwordx4"), ds=c(r"ds_read"), valu=c(r"\n\tv_(?!mfma)"),
                   salu=c(r"\n\ts_(?!waitcnt|barrier|nop)"), scratch=c(r"scratch_"), lanes=c(r"v_readlane|v_writelane"), s_load=c(r"s_load"),
                   gload=c(r"global_load"), vmcnt0=c(r"vmcnt\(0\)"), lgkm0=c(r"lgkmcnt\(0\)"))
        hot = c(r"s_cbranch_scc1 " + re.escape(name)) + c(r"s_cbranch_scc0 " + re.escape(name)) + c(r"s_cbranch_vcc\w+ " + re.escape(name))
        flag = row["scratch"] or row["lanes"] or row["s_load"] or row["gload"] or row["vmcnt0"]
        bad += 1 if flag else 0
        print(("BAD " if flag else "ok  ") + f"{kern[:40]:40s} {name:12s} line {i:6d} self-loop {hot} " + " ".join(f"{k}={v}" for k, v in row.items()))
    print("blocks with spills / drains:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
