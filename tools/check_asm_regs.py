#!/usr/bin/env python
"""Static check of the inline-asm operand loads of a kernel (csrc/glu_fwd.hip: gload_async + counted waits).

A fragment requested by `global_load_dwordx4 v[a:b]` inside an ASMSTART/ASMEND block lands asynchronously; between the request and
the counted wait that retires it NO compiler-generated instruction may read or write v[a:b] (a copy -- e.g. a spill to an AGPR --
would save the stale value, and the register, reused for something else, would later be overwritten by the landing load).
The compiler knows nothing about this contract, so the built ISA is checked: for every asm load, every instruction up to the
LOOKAHEAD-th following counted wait (`s_waitcnt vmcnt(n)` inside an asm block; text order, wrapping once to the first counted wait
of the kernel for loads near the end of the loop body) is scanned for the destination registers.  lookahead = 1 (any touch
before the FIRST counted wait: what a copy at a join or a dead-definition reuse looks like) is valid for every kernel;
glu_fwd requests two chunks ahead and may be checked with lookahead = 3.

    python tools/check_asm_regs.py <file.s> <kernel-symbol-substring> [lookahead=3]
    tools/check_asm_all.sh        # every 16-bit instantiation of glu_fwd / head_fwd / gemm_tm -> profiles/
"""
import re
import sys


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def main():
    path, sym = sys.argv[1], sys.argv[2]
    look = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and l.rstrip().endswith(sym_end(l)))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = lines[start:end]
    in_asm, kinds = False, []
    for l in body:
        if "ASMSTART" in l:
            in_asm = True
            kinds.append("marker")
            continue
        if "ASMEND" in l:
            in_asm = False
            kinds.append("marker")
            continue
        kinds.append("asm" if in_asm else "code")
    waits = [i for i, l in enumerate(body) if kinds[i] == "asm" and "s_waitcnt vmcnt" in l]
    bad = 0
    notes = 0
    nload = 0
    for i, l in enumerate(body):
        if kinds[i] != "asm" or not ("global_load_dwordx4" in l or "buffer_load_dwordx4" in l):   # (buffer form: csrc/glu_fwd_static.hip)
            continue
        if l.split(";")[0].rstrip().endswith(" lds"):     # an LDS-DMA piece has no destination register (its first operand is the address)
            continue
        nload += 1
        dst = regs_of(l.split(",")[0])
        later = [w for w in waits if w > i]
        span = list(range(i + 1, later[look - 1])) if len(later) >= look else list(range(i + 1, len(body))) + list(range(waits[0], waits[min(look - len(later), len(waits)) - 1]))
        for j in span:
            t = body[j].strip()
            if kinds[j] == "asm" and "s_waitcnt vmcnt(0)" in t:
                break                                    # a full drain retires every request
            if not t or t.startswith(";") or t.startswith(".") or kinds[j] == "marker":
                continue
            if kinds[j] == "asm" and ("global_load_dwordx4" in t or "buffer_load_dwordx4" in t or "s_waitcnt" in t or "ds_read" in t):
                # other asm loads / waits / LDS reads name their own registers
                if regs_of(t.split(",")[0]) & dst and ("global_load_dwordx4" in t or "buffer_load_dwordx4" in t) and j != i:
                    # a second request into a pending register: benign (loads return in order, the later one wins) and, in
                    # a linear scan, usually the other arm of a branch
                    notes += 1
                    break
                continue
            if regs_of(t) & dst:
                print(f"line {start + j + 1}: touches pending v{sorted(regs_of(t) & dst)} (requested at line {start + i + 1}): {t}")
                bad += 1
                break
    print(f"{sym}: {nload} asm loads checked, {bad} violation(s), {notes} repeated request(s)")
    return 1 if bad else 0


def sym_end(l):
    return l.rstrip().split()[-1] if l.rstrip() else ""


if __name__ == "__main__":
    sys.exit(main())
