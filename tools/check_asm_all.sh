#!/bin/bash
# Static check of every kernel that requests operands by inline-asm loads (gload_async): emit the gfx950 ISA and scan it with
# tools/check_asm_regs.py.  CPU only (hipcc cross-compiles).  Usage: tools/check_asm_all.sh [out.txt]
OUT=${1:-profiles/r03_asm_load_check.txt}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
{
echo "# tools/check_asm_all.sh: no compiler-generated instruction may touch a register between its inline-asm request and the"
echo "# counted wait that retires it (csrc/wae_common.hpp: gload_async).  hipcc $(/opt/rocm/bin/hipcc --version | grep -o 'HIP version.*')"
# (the ISA is a by-product of the product build: csrc/obj/<name>.s, csrc/Makefile)
make -C $ROOT/wavenet_autoencoders_amd/csrc -j8 > /dev/null
for f in glu_fwd glu_fwd_static head_fwd gemm_tm gemm_tm8 glu_bwd; do
  cp $ROOT/wavenet_autoencoders_amd/csrc/obj/$f.s $TMP/$f.s
  # lookahead: glu_fwd requests two chunks ahead; glu_fwd_static mixes two-chunks-ahead fragments with residual rows that are retired
  # by the very next counted wait (text-order scanning cannot tell them apart): the always-valid lookahead 1
  la=1; [ $f = glu_fwd ] && la=3
  for k in $(grep -o "^_Z[0-9]*[a-z_]*kernelIDF16[b_][A-Za-z0-9_]*" $TMP/$f.s | sort -u); do
    python3 $ROOT/tools/check_asm_regs.py $TMP/$f.s $k $la | tail -1
  done
done
} > $OUT
# second check: accidental drains of an asynchronous pipeline (scratch reloads / compiler-generated vmcnt(0) inside the innermost
# loops that keep LDS-DMA pieces or inline-asm requests in flight) -- tools/check_asm_drains.py
DR=${OUT%_load_check.txt}_drain_check.txt
[ "$DR" = "$OUT" ] && DR=${OUT%.txt}_drains.txt
{
echo "# tools/check_asm_drains.py over every 16-bit kernel with asynchronous requests: scratch reloads and compiler-generated"
echo "# s_waitcnt vmcnt(0) inside the innermost loops that issue LDS-DMA pieces / inline-asm loads (fp32 instantiations use plain"
echo "# loads and are expected to wait; they are not listed)"
for f in gemm_tn_stream glu_fwd head_fwd gemm_tm head_bwd gemm_tn glu_bwd; do
  [ -f $TMP/$f.s ] || cp $ROOT/wavenet_autoencoders_amd/csrc/obj/$f.s $TMP/$f.s
  python3 $ROOT/tools/check_asm_drains.py $TMP/$f.s IDF16 | grep -v "^_Z.*: 0 loop"
done
} > $DR
rm -rf $TMP
# third check: the hand-allocated register banks of the autoregressive kernels (tools/check_ar_banks.py)
python3 $ROOT/tools/check_ar_banks.py $ROOT/wavenet_autoencoders_amd/csrc/obj/ar_coop.s > ${OUT%_load_check.txt}_ar_banks.txt || echo "AR register bank check FAILED"
grep -c " 0 violation" $OUT | sed 's/$/ kernels clean/'
grep -c ", 0 scratch" $DR | sed 's/$/ kernels without drains in their asynchronous loops/'
grep "^_Z" $DR | grep -v ", 0 scratch" | sed 's/: .*loop(s) with asynchronous requests,/:/' | head -20
grep -v " 0 violation" $OUT | grep -v "^#" | head
