#!/usr/bin/env python3
"""Static check of the autoregressive kernels' hand-allocated register banks (csrc/ar_coop.hip): in every instantiation that keeps layer
packets in a[0:252] (and, ar_coop_fast_vb_kernel, in v[187:255]) no compiler-generated instruction may name an AGPR or a VGPR of the
bank.  The bank's own instructions carry their register numbers as expressions (`a[23*K+i]`, `v[187+23*K+i]`), which is how they are
told apart.  Usage: tools/check_ar_banks.py [ar_coop.s]  (emits the ISA itself when no file is given; CPU only)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    path = sys.argv[1]
else:
    path = os.path.join(tempfile.mkdtemp(), "ar_coop.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                    os.path.join(ROOT, "wavenet_autoencoders_amd/csrc/ar_coop.hip"), "-o", path], check=True, stderr=subprocess.DEVNULL)
s = open(path).read()
rc = 0
for m in re.finditer(r"^(_Z\d+ar_coop_fast\w*):[^\n]*\n", s, re.M):
    name = m.group(1)
    body = s[m.end():s.index(".end_amdhsa_kernel", m.end())]
    k = s.index(".set " + name + ".num_vgpr")
    nv, na = (int(re.search(r"\.num_%s, (\d+)" % w, s[k:k + 400]).group(1)) for w in ("vgpr", "agpr"))
    scratch = int(re.search(r"private_seg_size, (\d+)", s[k:k + 1200]).group(1))
    if na == 0:
        continue
    vb = "vb_kernel" in name
    bad, mx = [], 0
    for ln in body.split("\n"):
        code = ln.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        mine = "a[23*" in code or "v[187+23*" in code
        if mine:
            continue
        for r in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", code):
            hi = int(r.group(1) or r.group(3))
            mx = max(mx, hi)
            if vb and hi >= 186:
                bad.append(code)
        if re.search(r"\ba\d+\b|\ba\[\d+:\d+\]", code):
            bad.append(code)
    print(f"{name}: {nv} VGPRs + {na} AGPRs, scratch {scratch}; compiler's highest VGPR v{mx}; {len(bad)} violation(s)" + ("" if not bad else ": " + "; ".join(bad[:3])))
    rc |= bool(bad) or scratch > 0
sys.exit(rc)
