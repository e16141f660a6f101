"""The contracts of the hand-scheduled kernels, checked in the ISA of the library the GPU tests run (no GPU needed).

The kernels that request operands by inline asm rely on properties hipcc knows nothing about; a toolchain bump that moves one
register breaks them silently (round 5 found a wild-pointer fault of exactly this kind in csrc/glu_bwd.hip).  Three scanners, written in
rounds 2-5 and until round 6 run by hand (profiles/r0N_asm_*.txt), read the device assembly that the product build leaves in
csrc/obj/<name>.s (a by-product of the same compile: csrc/Makefile, -save-temps):

* tools/check_asm_regs.py   -- no compiler-generated instruction touches a register between the inline-asm load that requests it and
                              the counted `s_waitcnt vmcnt(n)` that retires it (csrc/wae_common.hpp: gload_async);
* tools/check_ar_banks.py   -- no compiler-generated instruction names a register of the autoregressive kernels' hand-allocated banks
                              (a[0:252], v[187:255]: the registers behind the `a255` / `v255` clobbers -Winline-asm calls reserved), no
                              scratch in those kernels;
* tools/check_tq_loops.py   -- the hot loops of the weight-gradient kernel (its own register bank v[16:167]) are free of scratch,
                              scalar / global loads and compiler-made drains;
* tools/check_asm_drains.py -- no scratch reload and no compiler-generated `s_waitcnt vmcnt(0)` inside the innermost loops that keep
                              LDS-DMA pieces / asm requests in flight, for the kernels of the benchmarked 16-bit step.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wavenet_autoencoders_amd", "csrc")
OBJ = os.path.join(CSRC, "obj")
TOOLS = os.path.join(ROOT, "tools")


@pytest.fixture(scope="module")
def isa():
    """csrc/obj/*.s, up to date with the sources (make is a no-op after __graft_entry__.build(); a fresh checkout compiles here)."""
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    files = {f[:-2]: os.path.join(OBJ, f) for f in os.listdir(OBJ) if f.endswith(".s")}
    assert {"glu_fwd", "glu_fwd_static", "head_fwd", "gemm_tm", "gemm_tm8", "glu_bwd", "ar_coop", "gemm_tn_static"} <= set(files)
    return files


def _run(tool, *args):
    r = subprocess.run([sys.executable, os.path.join(TOOLS, tool)] + list(args), capture_output=True, text=True)
    return r.returncode, r.stdout + r.stderr


def _kernels16(path):
    return sorted(set(re.findall(r"^(_Z[0-9]*[a-z_0-9]*kernelIDF16[b_][A-Za-z0-9_]*):", open(path).read(), re.M)))


# (csrc/gemm_tm8.hip requests everything by LDS-DMA: it has no register with a load in flight; its loops are in the drain test below)
@pytest.mark.parametrize("name,lookahead", [("glu_fwd", 3), ("glu_fwd_static", 1), ("head_fwd", 1), ("gemm_tm", 1), ("glu_bwd", 1)])
def test_no_instruction_touches_a_register_with_a_load_in_flight(isa, name, lookahead):
    kernels = _kernels16(isa[name])
    assert kernels, name
    bad, nloads = [], 0
    for k in kernels:
        rc, out = _run("check_asm_regs.py", isa[name], k, str(lookahead))
        last = out.strip().split("\n")[-1]
        m = re.search(r"(\d+) asm loads checked, (\d+) violation", last)
        assert m, (k, out[-400:])
        nloads += int(m.group(1))
        if int(m.group(2)):
            bad.append(last)
    assert not bad, bad
    assert nloads > 0 or name == "head_fwd", "the scanner found no inline-asm loads at all: its patterns no longer match the ISA"


def test_register_banks_of_the_autoregressive_kernels(isa):
    rc, out = _run("check_ar_banks.py", isa["ar_coop"])
    lines = [l for l in out.split("\n") if l.startswith("_Z")]
    assert len(lines) >= 4, out          # bf16 + fp16 instantiations of the banked kernels
    assert rc == 0 and all(" 0 violation(s)" in l and "scratch 0;" in l for l in lines), out


def test_weight_gradient_kernel_hot_loops(isa):
    rc, out = _run("check_tq_loops.py", isa["gemm_tn_static"])
    rows = [l for l in out.split("\n") if l[:4] in ("ok  ", "BAD ")]
    loops = [l for l in rows if re.search(r"self-loop [1-9]", l)]          # the steady-state blocks (a block that branches to itself)
    assert len(loops) >= 20, out[-1500:]
    bad = [l for l in loops if l.startswith("BAD")]
    assert not bad, bad
    # (blocks that are not loops -- a segment's last half-slab, flushes -- may drain: one such block per instantiation does)


@pytest.mark.parametrize("name,pattern", [("glu_fwd_static", "glu_fwd_static"), ("gemm_tm", "gemm_tm_kernelIDF16"), ("gemm_tm8", "gemm_tm8"),
                                          ("glu_bwd", "glu_bwd_pair_kernel"), ("gemm_tn_stream", "gemm_tn_stream_kernelIDF16")])
def test_no_drain_inside_the_asynchronous_loops_of_the_16bit_step(isa, name, pattern):
    rc, out = _run("check_asm_drains.py", isa[name], pattern)
    lines = [l for l in out.split("\n") if l.startswith("_Z")]
    assert lines, out[-400:]
    # the C5 instantiation of the static layer kernel (Rp 512, Hp 256: 8 tile pairs, 8 column blocks) keeps loops and spills inside
    # them (a known cost of that geometry): bounded here so that it cannot grow unnoticed
    wide = [l for l in lines if "Li8ELi4ELb0ELi8E" in l]
    bad = [l for l in lines if l not in wide and not re.search(r", 0 scratch reload", l)]
    assert not bad, bad
    for l in wide:
        assert int(re.search(r", (\d+) scratch reload", l).group(1)) <= 24, l
