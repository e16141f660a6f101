#!/usr/bin/env python
"""Static check for accidental drains of an asynchronous load pipeline (LDS-DMA rings, inline-asm operand requests).

A kernel that keeps loads in flight across loop iterations (global_load_lds pieces, gload_async fragments) is only as deep as the
weakest wait in its loop: one compiler-generated `s_waitcnt vmcnt(0)` -- typically in front of the first use of a register that
hipcc parked in scratch (`scratch_load ... Folded Reload`) -- drains the whole ring on every iteration.  The paced instantiation
of gemm_tn_stream did exactly that in round 2 (168-register budget, one VGPR over), which is why "pacing" measured slower whatever
the window was.  This tool lists, for every loop of every kernel in the given ISA file that contains such asynchronous requests:
scratch reloads and compiler-generated (outside ASMSTART/ASMEND) `s_waitcnt vmcnt(0)` inside the loop.

    python tools/check_asm_drains.py <file.s> [kernel-symbol-substring]
    tools/check_asm_all.sh          # runs it over every csrc/*.hip -> profiles/
"""
import re
import sys


def main():
    path = sys.argv[1]
    sym = sys.argv[2] if len(sys.argv) > 2 else ""
    src = open(path).read().split("\n")
    starts = [i for i, l in enumerate(src) if re.match(r"^_Z\w+:", l) and sym in l]
    total = 0
    for s in starts:
        name = src[s].split(":")[0]
        e = next(i for i in range(s, len(src)) if src[i].startswith(".Lfunc_end"))
        body = src[s:e]
        in_asm, asm = False, []
        for l in body:
            if "ASMSTART" in l:
                in_asm = True
            asm.append(in_asm)
            if "ASMEND" in l:
                in_asm = False
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = i
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] <= i:
                loops.append((labels[m.group(1)], i))
        # merge back-edges to the same header
        ext = {}
        for h, t in loops:
            ext[h] = max(ext.get(h, h), t)
        findings = []
        def is_async(h, t):
            return any("global_load_lds" in body[i] or (asm[i] and "global_load_dwordx4" in body[i]) for i in range(h, t + 1))
        al = [(h, t) for h, t in sorted(ext.items()) if is_async(h, t)]
        # the innermost loops with requests are the steady state (an outer tile / segment loop drains legitimately between tiles)
        inner = [(h, t) for h, t in al if not any((h2, t2) != (h, t) and h <= h2 and t2 <= t for h2, t2 in al)]
        for h, t in inner:
            seg = range(h, t + 1)
            hits = [(i, body[i].strip()) for i in seg
                    if "scratch_load" in body[i] or (not asm[i] and re.search(r"s_waitcnt\s+vmcnt\(0\)", body[i]))]
            n = sum(1 for i in seg if body[i].startswith("\t") and not body[i].strip().startswith((";", ".")))
            findings.append((body[h].split(":")[0], n, hits))
        bad = sum(len(hh) for _, _, hh in findings)
        total += bad
        print(f"{name[:110]}: {len(findings)} loop(s) with asynchronous requests, {bad} scratch reload(s) / compiler drain(s) inside them")
        for lab, n, hits in findings:
            for i, t in hits[:6]:
                print(f"    loop {lab} ({n} instructions) +{i}: {t[:100]}")
    return 0


if __name__ == "__main__":
    try:
        sys.exit(main())
    except BrokenPipeError:
        pass
