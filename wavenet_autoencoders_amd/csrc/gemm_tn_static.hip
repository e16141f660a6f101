// wae_gemm_tn_static: ALL weight-gradient contractions of a backward pass in ONE launch, static schedule (round 4).
//
//   C_j[m][n] += alpha_j * sum_{clip b, t} P_j[b,t][m] * Q_j[b, t + shift_j][n]        for every job j
//
// Same arithmetic and the same stream-K decomposition as csrc/gemm_tn_stream.hip (autograd of the 1x1 and dilated convolutions of
// ResidualConv1dGLU, modules.py:134-136,141-145,157-160): a job keeps a 96-tile output region (12 waves x 8 accumulator tiles)
// resident while it streams its share of the time axis; teams of workgroups -- the jobs of one layer -- walk the same segment list.
// What differs is everything around the MFMAs (the slab loop of the round-1 kernel took 3414 clocks per 32 time rows for 1536 clocks
// of matrix pipe: DMA wait 352 + zero-fill and barrier 1348 + request issue 601 + reads and MFMAs 1105, profiles/r03_ablate_tm.txt):
//   * Operand rows come through per-clip BUFFER DESCRIPTORS (buffer_load_dwordx4 ... lds): a row before the clip's first or past its
//     last sample (the causal shift of a tap, a ragged tail), a column past the job's width and the row padding of the LDS image are
//     out of range and the hardware writes zeros -- no zero-fill pass, no clamped-row path, no exec masks, no 64-bit address
//     arithmetic per slab (a request is one v_add of a per-lane constant and a scalar row offset).
//   * The ring holds 6-7 HALF-slabs of 16 time rows (one MFMA k-step), requests run NS - 1 half-slabs ahead, and the transposed LDS
//     reads of half-slab h + 1 are issued between the MFMAs of half-slab h, each fragment into the registers the last MFMA that
//     needed them has just read (24 hand-allocated operand registers, progressive refill): the barrier at the top of a half-slab
//     no longer exposes an LDS round trip, and the requests of a half-slab are spread over its MFMAs instead of queueing in front
//     of them.  All waits are counted immediates.
//   * Job KINDS are compile-time shapes: TAPS (dz x x[t - shift], 384 x 256), COND (dz x [c | per-clip ones], one M-tile per wave,
//     the ones operand in registers) and OUTSKIP (u x [Ghat | dS], 192 x 512): conv1x1_out and conv1x1_skip of a layer share the
//     operand u, so the layer's skip-weight gradient rides in what used to be the padding of the dW_out job (its bias gradient, the
//     column sums of Ghat, are VALU sums over the B fragments two waves hold anyway) and the separate tile launch over
//     dS x u of all layers is gone.
// The operand bank v[144:167] is reserved from the compiler (amdgpu_num_vgpr): reads stay in flight across asm statements and
// compiler-generated code in between (DMA requests, cursors) can never touch them.
#include "wae_common.hpp"

// (the bank registers are reserved from the compiler on purpose; naming one as a clobber is what sizes the kernel's register file)
#pragma clang diagnostic ignored "-Winline-asm"

struct TqJob {
  const char* P;    // (B,T,p_stride) 16-bit, already offset to the job's first column
  const char* Q0;   // (B,T,q0_stride)
  const char* Q1;   // OUTSKIP: second half of the N axis (dS); else null
  float* C0;        // output rows m, columns n < 256
  float* C1;        // OUTSKIP: columns n >= 256
  float* Cb;        // OUTSKIP: column sums of Q0 (n0_valid floats) or null
  int64_t p_stride, q0_stride, q1_stride, ldc0, ldc1;   // elements
  int m_valid, n0_valid, n1_valid;                      // multiples of 8
  int shift;                                            // Q row = t + shift
  int ones_col;                                         // COND: C0 column of clip 0's sums (n0_valid <= ones_col, ones_col % 32 == 0)
  int kind;
  float alpha;
  int pad_;
};
struct TqSeg {
  int job, slab_begin, slab_end;   // job of team member 0; 32-row slabs numbered b * slabs_per_clip + t / 32
};
// A job as wave-uniform scalars.  The table is read with vector loads (the kernel's own atomics may alias it as far as the compiler
// can tell, so it will not use the scalar cache) and every word goes through v_readfirstlane: nothing of a job lives in a VGPR or --
// what a by-value copy of the struct did, once a field was selected by a run-time index -- in scratch.
struct TqJobS {
  const char *P, *Q0, *Q1;
  float *C0, *C1, *Cb;
  unsigned sbP, sbQ0, sbQ1;   // row strides in BYTES
  int ldc0, ldc1;
  int m_valid, n0_valid, n1_valid, shift, ones_col, kind;
  float alpha;
};
// (by value: `c ? jb.x : jb.y` on two lvalues is an lvalue, i.e. ONE load through a selected address, and that keeps the record in scratch)
template <typename T>
__device__ __forceinline__ T tq_sel3(int sub, T a, T b, T c) { return sub == 0 ? a : (sub == 1 ? b : c); }
__device__ __forceinline__ int tq_ldw(const int* q, int i) { return __builtin_amdgcn_readfirstlane(q[i]); }
__device__ __forceinline__ char* tq_ldp(const int* q, int i) {
  const unsigned long long lo = (unsigned)tq_ldw(q, i), hi = (unsigned)tq_ldw(q, i + 1);
  return (char*)(lo | (hi << 32));
}
__device__ __forceinline__ TqJobS tq_load_job(const TqJob* j) {
  const int* q = (const int*)j;
  TqJobS r;
  r.P = tq_ldp(q, 0); r.Q0 = tq_ldp(q, 2); r.Q1 = tq_ldp(q, 4);
  r.C0 = (float*)tq_ldp(q, 6); r.C1 = (float*)tq_ldp(q, 8); r.Cb = (float*)tq_ldp(q, 10);
  r.sbP = (unsigned)tq_ldw(q, 12) * 2; r.sbQ0 = (unsigned)tq_ldw(q, 14) * 2; r.sbQ1 = (unsigned)tq_ldw(q, 16) * 2;   // (strides < 2^30)
  r.ldc0 = tq_ldw(q, 18); r.ldc1 = tq_ldw(q, 20);
  r.m_valid = tq_ldw(q, 22); r.n0_valid = tq_ldw(q, 23); r.n1_valid = tq_ldw(q, 24);
  r.shift = tq_ldw(q, 25); r.ones_col = tq_ldw(q, 26); r.kind = tq_ldw(q, 27);
  r.alpha = __builtin_bit_cast(float, tq_ldw(q, 28));
  return r;
}
struct TqArgs {
  const TqJob* jobs;
  const TqSeg* segs;
  const int* team_seg;
  int nteams, team_size;
  int B, T, spc;
  long long* stamps;   // WAE_TQ_STAMPS builds: [workgroup][16][4]
  unsigned* pace;             // [nteams] words of three 10-bit request positions (the tap members, in 32-row slabs), zeroed by the caller; or null
  int window;                 // a paced member requests at most `window` slabs beyond the slowest other tap member (0: no pacing)
  int ntaps;                  // members [0, ntaps) are the taps, member ntaps is COND
  int window_cond;            // the same bound for COND (may be negative: stay that many slabs BEHIND the slowest tap, where every dz
                              // half-slab it asks for is already in L2)
  int dump_off;               // byte offset of a 1-KiB LDS area behind the rings (the pacing wave's filler requests land there)
};

enum { TQ_TAPS = 0, TQ_COND = 1, TQ_OUTSKIP = 2 };
#define TQ_NW 12
#define TQ_OOB 0x80000000u    // per-lane marker: row padding / columns past the job's width (any row offset keeps it out of range)
#define TQ_DEAD 0x40000000u   // scalar row offset of requests behind the segment's end (num_records < 2^30: host check)

template <int KIND> struct TqGeo;
template <> struct TqGeo<TQ_TAPS> { static constexpr int PP = 832, QP = 576, NQ = 1, NS = 7, PIECES_MAX = 2; };
template <> struct TqGeo<TQ_COND> { static constexpr int PP = 832, QP = 192, NQ = 1, NS = 7, PIECES_MAX = 2; };
template <> struct TqGeo<TQ_OUTSKIP> { static constexpr int PP = 448, QP = 576, NQ = 2, NS = 6, PIECES_MAX = 3; };
template <int KIND> struct TqDer {
  using G = TqGeo<KIND>;
  static constexpr int NPP = 16 * G::PP / 1024, NPQ = 16 * G::QP / 1024, NPIECES = NPP + G::NQ * NPQ, SLOT = NPIECES * 1024;
  static constexpr int PU = G::PP / 16, QU = G::QP / 16, D = G::NS - 1;
  static_assert(NPP * 1024 == 16 * G::PP && NPQ * 1024 == 16 * G::QP, "sub-images are whole 1-KiB pieces");
  static_assert((G::PP % 256 == 64 || G::PP % 256 == 192) && (G::QP % 256 == 64 || G::QP % 256 == 192), "conflict-free transposed reads");
  static_assert(G::NS * SLOT <= 160 * 1024, "LDS");
  static_assert((G::PIECES_MAX - 1) * TQ_NW < NPIECES && NPIECES <= G::PIECES_MAX * TQ_NW, "pieces per wave");
};

// ---- the register bank -----------------------------------------------------------------------------------------------------------
// v10        the offset of the request being issued
// v[11:13]   per lane and piece: row * stride + column bytes inside a half-slab, or TQ_OOB      (set once per segment); the wave
//            that paces its workgroup has one piece (v11); v12 carries its pacing atomic's data, v13 the team's word as last returned
// v14, v15   per lane: LDS addresses of the A / B fragments being READ (they run over the ring with the slot of the next half-slab)
// v[16:143]  the wave's accumulator tiles, tile k at v[16 + 16 k : 31 + 16 k]   (2 x 4 layout: k = 4 i + j; COND: k = 0, 1, ones)
// v[144:167] operand fragments: A0 v[144:147], A1 v[148:151], B0..B3 v[152:167]   (COND: A0, B0, B1, ones = v[160:163])
// The compiler allocates v0..v9 only (amdgpu_num_vgpr on the kernel), and the hot loop uses NONE of them: every vector instruction
// of a half-slab -- MFMAs, fragment reads, request offsets, the requests themselves -- is asm on bank registers; what the compiler
// contributes is scalar (descriptors, row offsets, ring cursors, the loop).  With compiler-managed accumulators ("+v" operands) every
// change of the surrounding code moved tiles in and out of scratch inside the loop; with compiler-managed request offsets it parked
// them in scratch and reloaded them per request -- and a reload waits vmcnt(0), i.e. drains the request ring every half-slab.
// Only "v167" is named as a clobber: it makes the kernel descriptor cover the bank.
#define TQ_BANK "v167"
typedef int tq_i32x4 __attribute__((ext_vector_type(4)));
#define TQ_ACC0 16

template <int CNT>
__device__ __forceinline__ void tq_wait_vm() {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}

// accumulator tile k as an asm register range
#define TQ_T0 "v[16:31]"
#define TQ_T1 "v[32:47]"
#define TQ_T2 "v[48:63]"
#define TQ_T3 "v[64:79]"
#define TQ_T4 "v[80:95]"
#define TQ_T5 "v[96:111]"
#define TQ_T6 "v[112:127]"
#define TQ_T7 "v[128:143]"
#define TQ_MFMA(MF, T, A, B) "v_mfma_f32_32x32x16_" MF " " T ", " A ", " B ", " T "\n\t"
#define TQ_A0 "v[144:147]"
#define TQ_A1 "v[148:151]"
#define TQ_B0 "v[152:155]"
#define TQ_B1 "v[156:159]"
#define TQ_B2 "v[160:163]"
#define TQ_B3 "v[164:167]"

__device__ __forceinline__ void tq_zero_acc() {   // all 128 accumulator registers
  asm volatile(
      "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\t"
      "v_mov_b64 v[18:19], v[16:17]\n\tv_mov_b64 v[20:21], v[16:17]\n\tv_mov_b64 v[22:23], v[16:17]\n\tv_mov_b64 v[24:25], v[16:17]\n\t"
      "v_mov_b64 v[26:27], v[16:17]\n\tv_mov_b64 v[28:29], v[16:17]\n\tv_mov_b64 v[30:31], v[16:17]\n\t"
      "v_mov_b64 v[32:33], v[16:17]\n\tv_mov_b64 v[34:35], v[16:17]\n\tv_mov_b64 v[36:37], v[16:17]\n\tv_mov_b64 v[38:39], v[16:17]\n\t"
      "v_mov_b64 v[40:41], v[16:17]\n\tv_mov_b64 v[42:43], v[16:17]\n\tv_mov_b64 v[44:45], v[16:17]\n\tv_mov_b64 v[46:47], v[16:17]\n\t"
      "v_mov_b64 v[48:49], v[16:17]\n\tv_mov_b64 v[50:51], v[16:17]\n\tv_mov_b64 v[52:53], v[16:17]\n\tv_mov_b64 v[54:55], v[16:17]\n\t"
      "v_mov_b64 v[56:57], v[16:17]\n\tv_mov_b64 v[58:59], v[16:17]\n\tv_mov_b64 v[60:61], v[16:17]\n\tv_mov_b64 v[62:63], v[16:17]\n\t"
      "v_mov_b64 v[64:65], v[16:17]\n\tv_mov_b64 v[66:67], v[16:17]\n\tv_mov_b64 v[68:69], v[16:17]\n\tv_mov_b64 v[70:71], v[16:17]\n\t"
      "v_mov_b64 v[72:73], v[16:17]\n\tv_mov_b64 v[74:75], v[16:17]\n\tv_mov_b64 v[76:77], v[16:17]\n\tv_mov_b64 v[78:79], v[16:17]\n\t"
      "v_mov_b64 v[80:81], v[16:17]\n\tv_mov_b64 v[82:83], v[16:17]\n\tv_mov_b64 v[84:85], v[16:17]\n\tv_mov_b64 v[86:87], v[16:17]\n\t"
      "v_mov_b64 v[88:89], v[16:17]\n\tv_mov_b64 v[90:91], v[16:17]\n\tv_mov_b64 v[92:93], v[16:17]\n\tv_mov_b64 v[94:95], v[16:17]\n\t"
      "v_mov_b64 v[96:97], v[16:17]\n\tv_mov_b64 v[98:99], v[16:17]\n\tv_mov_b64 v[100:101], v[16:17]\n\tv_mov_b64 v[102:103], v[16:17]\n\t"
      "v_mov_b64 v[104:105], v[16:17]\n\tv_mov_b64 v[106:107], v[16:17]\n\tv_mov_b64 v[108:109], v[16:17]\n\tv_mov_b64 v[110:111], v[16:17]\n\t"
      "v_mov_b64 v[112:113], v[16:17]\n\tv_mov_b64 v[114:115], v[16:17]\n\tv_mov_b64 v[116:117], v[16:17]\n\tv_mov_b64 v[118:119], v[16:17]\n\t"
      "v_mov_b64 v[120:121], v[16:17]\n\tv_mov_b64 v[122:123], v[16:17]\n\tv_mov_b64 v[124:125], v[16:17]\n\tv_mov_b64 v[126:127], v[16:17]\n\t"
      "v_mov_b64 v[128:129], v[16:17]\n\tv_mov_b64 v[130:131], v[16:17]\n\tv_mov_b64 v[132:133], v[16:17]\n\tv_mov_b64 v[134:135], v[16:17]\n\t"
      "v_mov_b64 v[136:137], v[16:17]\n\tv_mov_b64 v[138:139], v[16:17]\n\tv_mov_b64 v[140:141], v[16:17]\n\tv_mov_b64 v[142:143], v[16:17]\n\t"
      "s_nop 1"
      : : : TQ_BANK);
}
// registers 4 Q .. 4 Q + 3 of accumulator tile K -> four compiler-visible values (the flush)
template <int K, int Q>
__device__ __forceinline__ void tq_get_quad(float (&t)[4]) {
  static_assert(K >= 0 && K < 8 && Q >= 0 && Q < 4, "tile / quad index");
  constexpr int R = TQ_ACC0 + 16 * K + 4 * Q;
  // (the register number is part of the instruction text: one asm statement per base register, chosen at compile time)
#define TQ_Q(BASE)                                                                                                       \
  if constexpr (R == BASE)                                                                                               \
    asm volatile("v_mov_b32 %0, v[" #BASE "+0]\n\tv_mov_b32 %1, v[" #BASE "+1]\n\tv_mov_b32 %2, v[" #BASE "+2]\n\tv_mov_b32 %3, v[" #BASE "+3]" \
                 : "=v"(t[0]), "=v"(t[1]), "=v"(t[2]), "=v"(t[3]))
  TQ_Q(16); TQ_Q(20); TQ_Q(24); TQ_Q(28); TQ_Q(32); TQ_Q(36); TQ_Q(40); TQ_Q(44); TQ_Q(48); TQ_Q(52); TQ_Q(56); TQ_Q(60);
  TQ_Q(64); TQ_Q(68); TQ_Q(72); TQ_Q(76); TQ_Q(80); TQ_Q(84); TQ_Q(88); TQ_Q(92); TQ_Q(96); TQ_Q(100); TQ_Q(104); TQ_Q(108);
  TQ_Q(112); TQ_Q(116); TQ_Q(120); TQ_Q(124); TQ_Q(128); TQ_Q(132); TQ_Q(136); TQ_Q(140);
#undef TQ_Q
}

// ---- per-segment lane constants, ring cursors of the fragment reads, requests -----------------------------------------------------
__device__ __forceinline__ void tq_set_lane_consts(unsigned vb0, unsigned vb1, unsigned vb2, unsigned ap, unsigned bp) {
  asm volatile("v_mov_b32 v11, %0\n\tv_mov_b32 v12, %1\n\tv_mov_b32 v13, %2\n\tv_mov_b32 v14, %3\n\tv_mov_b32 v15, %4"
               : : "v"(vb0), "v"(vb1), "v"(vb2), "v"(ap), "v"(bp) : TQ_BANK);
}
__device__ __forceinline__ void tq_step_read_slot(int delta) {   // delta: + SLOT, or - (NS - 1) SLOT at the end of the ring
  asm volatile("v_add_u32 v14, %0, v14\n\tv_add_u32 v15, %0, v15" : : "s"(delta) : TQ_BANK);
}
// piece J of this wave: LDS[lds_dst + 16 lane] <- 16 bytes at descriptor offset vb_J + ro (out of range: zeros).  M0 is written first
// and the offset formed behind it (an LDS-DMA instruction needs one instruction between the M0 write and itself).
template <int J>
__device__ __forceinline__ void tq_request(tq_i32x4 srd, unsigned ro, unsigned lds_dst) {
  static_assert(J >= 0 && J < 3, "piece index");
  if constexpr (J == 0)
    asm volatile("s_mov_b32 m0, %2\n\tv_add_u32 v10, %1, v11\n\tbuffer_load_dwordx4 v10, %0, 0 offen lds" : : "s"(srd), "s"(ro), "s"(lds_dst) : "m0", TQ_BANK);
  else if constexpr (J == 1)
    asm volatile("s_mov_b32 m0, %2\n\tv_add_u32 v10, %1, v12\n\tbuffer_load_dwordx4 v10, %0, 0 offen lds" : : "s"(srd), "s"(ro), "s"(lds_dst) : "m0", TQ_BANK);
  else
    asm volatile("s_mov_b32 m0, %2\n\tv_add_u32 v10, %1, v13\n\tbuffer_load_dwordx4 v10, %0, 0 offen lds" : : "s"(srd), "s"(ro), "s"(lds_dst) : "m0", TQ_BANK);
}

// ---- the MFMA / transposed-read streams ----------------------------------------------------------------------------------------
// 2 x 4 tiles per wave (TAPS, OUTSKIP).  In flight at the top of a half-slab, in issue order: A0 B0 B1 B2 B3 A1 (two
// ds_read_b64_tr_b16 each) = 12 LDS operations.  v14 / v15: the lane's A / B read addresses in the slot being read.
#define TQ_RD(DST, ADDR, OFF) "ds_read_b64_tr_b16 " DST ", " ADDR " offset:" OFF "\n\t"
template <int PP, int QP>
__device__ __forceinline__ void tq_read_all_2x4() {
  asm volatile(TQ_RD("v[144:145]", "v14", "0") TQ_RD("v[146:147]", "v14", "%0")
               TQ_RD("v[152:153]", "v15", "0") TQ_RD("v[154:155]", "v15", "%1")
               TQ_RD("v[156:157]", "v15", "64") TQ_RD("v[158:159]", "v15", "%2")
               TQ_RD("v[160:161]", "v15", "128") TQ_RD("v[162:163]", "v15", "%3")
               TQ_RD("v[164:165]", "v15", "192") TQ_RD("v[166:167]", "v15", "%4")
               TQ_RD("v[148:149]", "v14", "64") TQ_RD("v[150:151]", "v14", "%5")
               : : "n"(4 * PP), "n"(4 * QP), "n"(4 * QP + 64), "n"(4 * QP + 128), "n"(4 * QP + 192), "n"(4 * PP + 64) : TQ_BANK);
}
// S1: tiles (0,0) (0,1)
template <bool F16>
__device__ __forceinline__ void tq_s1() {
#define TQ_S1(MF) \
  asm volatile("s_waitcnt lgkmcnt(8)\n\t" TQ_MFMA(MF, TQ_T0, TQ_A0, TQ_B0) "s_waitcnt lgkmcnt(6)\n\t" TQ_MFMA(MF, TQ_T1, TQ_A0, TQ_B1) : : : TQ_BANK)
  if constexpr (F16) { TQ_S1("f16"); } else { TQ_S1("bf16"); }
#undef TQ_S1
}
// S2: tiles (0,2) (0,3); A0 is dead behind them -> its next fragment
template <bool F16, int PP>
__device__ __forceinline__ void tq_s2() {
#define TQ_S2(MF)                                                                                                              \
  asm volatile("s_waitcnt lgkmcnt(4)\n\t" TQ_MFMA(MF, TQ_T2, TQ_A0, TQ_B2) "s_waitcnt lgkmcnt(2)\n\t" TQ_MFMA(MF, TQ_T3, TQ_A0, TQ_B3) \
               TQ_RD("v[144:145]", "v14", "0") TQ_RD("v[146:147]", "v14", "%0") : : "n"(4 * PP) : TQ_BANK)
  if constexpr (F16) { TQ_S2("f16"); } else { TQ_S2("bf16"); }
#undef TQ_S2
}
// S3: tiles (1,0) (1,1); B0, B1 refilled.  BIAS: the wave also adds up the columns of B0, B1 (sum over its 8 k per lane).
// (column sums by plain VALU on the unpacked halves: v_dot2c_f32_bf16 measured ~120 clocks apiece beside the MFMAs of three waves --
//  the dot unit is the matrix pipe -- and made the two waves that carry the sums 5 x slower than their workgroup: 2378 against 470
//  clocks per half-slab.  v10 is free between two requests.)
#define TQ_SUM1_BF16(ACC, R) "v_lshlrev_b32 v10, 16, " R "\n\tv_add_f32 " ACC ", v10, " ACC "\n\tv_and_b32 v10, 0xffff0000, " R "\n\tv_add_f32 " ACC ", v10, " ACC "\n\t"
#define TQ_SUM1_F16(ACC, R) "v_fma_mix_f32 " ACC ", " R ", 1.0, " ACC " op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 " ACC ", " R ", 1.0, " ACC " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
#if defined(WAE_TQ_EXP) && WAE_TQ_EXP == 1      // timing experiment: the bias waves' loop without its sums
#define TQ_SUM4_bf16(ACC, R0, R1, R2, R3) ""
#elif defined(WAE_TQ_EXP) && WAE_TQ_EXP == 2    // timing experiment: one instruction per register
#define TQ_SUM4_bf16(ACC, R0, R1, R2, R3) "v_add_f32 " ACC ", " R0 ", " ACC "\n\tv_add_f32 " ACC ", " R1 ", " ACC "\n\tv_add_f32 " ACC ", " R2 ", " ACC "\n\tv_add_f32 " ACC ", " R3 ", " ACC "\n\t"
#else
#define TQ_SUM4_bf16(ACC, R0, R1, R2, R3) TQ_SUM1_BF16(ACC, R0) TQ_SUM1_BF16(ACC, R1) TQ_SUM1_BF16(ACC, R2) TQ_SUM1_BF16(ACC, R3)
#endif
#define TQ_SUM4_f16(ACC, R0, R1, R2, R3) TQ_SUM1_F16(ACC, R0) TQ_SUM1_F16(ACC, R1) TQ_SUM1_F16(ACC, R2) TQ_SUM1_F16(ACC, R3)
template <bool F16, int QP, bool BIAS>
__device__ __forceinline__ void tq_s3(float& s0, float& s1) {
#define TQ_S3(MF, SUM4)                                                                                                     \
  if constexpr (BIAS)                                                                                                            \
    asm volatile("s_waitcnt lgkmcnt(2)\n\t" TQ_MFMA(MF, TQ_T4, TQ_A1, TQ_B0) SUM4("%0", "v152", "v153", "v154", "v155") \
                 TQ_RD("v[152:153]", "v15", "0") TQ_RD("v[154:155]", "v15", "%2")                                                \
                 TQ_MFMA(MF, TQ_T5, TQ_A1, TQ_B1) SUM4("%1", "v156", "v157", "v158", "v159")                       \
                 TQ_RD("v[156:157]", "v15", "64") TQ_RD("v[158:159]", "v15", "%3")                                               \
                 : "+v"(s0), "+v"(s1) : "n"(4 * QP), "n"(4 * QP + 64) : TQ_BANK);                                                \
  else                                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(2)\n\t" TQ_MFMA(MF, TQ_T4, TQ_A1, TQ_B0)                                                     \
                 TQ_RD("v[152:153]", "v15", "0") TQ_RD("v[154:155]", "v15", "%0")                                                \
                 TQ_MFMA(MF, TQ_T5, TQ_A1, TQ_B1)                                                                                \
                 TQ_RD("v[156:157]", "v15", "64") TQ_RD("v[158:159]", "v15", "%1")                                               \
                 : : "n"(4 * QP), "n"(4 * QP + 64) : TQ_BANK)
  if constexpr (F16) { TQ_S3("f16", TQ_SUM4_f16); } else { TQ_S3("bf16", TQ_SUM4_bf16); }
#undef TQ_S3
}
// S4: tiles (1,2) (1,3); B2, B3, A1 refilled
template <bool F16, int PP, int QP, bool BIAS>
__device__ __forceinline__ void tq_s4(float& s2, float& s3) {
#define TQ_S4(MF, SUM4)                                                                                                     \
  if constexpr (BIAS)                                                                                                            \
    asm volatile(TQ_MFMA(MF, TQ_T6, TQ_A1, TQ_B2) SUM4("%0", "v160", "v161", "v162", "v163")                       \
                 TQ_RD("v[160:161]", "v15", "128") TQ_RD("v[162:163]", "v15", "%2")                                              \
                 TQ_MFMA(MF, TQ_T7, TQ_A1, TQ_B3) SUM4("%1", "v164", "v165", "v166", "v167")                       \
                 TQ_RD("v[164:165]", "v15", "192") TQ_RD("v[166:167]", "v15", "%3")                                              \
                 TQ_RD("v[148:149]", "v14", "64") TQ_RD("v[150:151]", "v14", "%4")                                               \
                 : "+v"(s2), "+v"(s3) : "n"(4 * QP + 128), "n"(4 * QP + 192), "n"(4 * PP + 64) : TQ_BANK);                       \
  else                                                                                                                           \
    asm volatile(TQ_MFMA(MF, TQ_T6, TQ_A1, TQ_B2)                                                                                \
                 TQ_RD("v[160:161]", "v15", "128") TQ_RD("v[162:163]", "v15", "%0")                                              \
                 TQ_MFMA(MF, TQ_T7, TQ_A1, TQ_B3)                                                                                \
                 TQ_RD("v[164:165]", "v15", "192") TQ_RD("v[166:167]", "v15", "%1")                                              \
                 TQ_RD("v[148:149]", "v14", "64") TQ_RD("v[150:151]", "v14", "%2")                                               \
                 : : "n"(4 * QP + 128), "n"(4 * QP + 192), "n"(4 * PP + 64) : TQ_BANK)
  if constexpr (F16) { TQ_S4("f16", TQ_SUM4_f16); } else { TQ_S4("bf16", TQ_SUM4_bf16); }
#undef TQ_S4
}

// 1 x (2 + ones) tiles per wave (COND): A0, B0, B1 from LDS, the per-clip ones operand v[160:163] lives in registers only;
// accumulator tiles 0, 1 (c columns) and 2 (the clips' sums).  In flight at the top of a half-slab, in issue order: B0 B1 A0.
template <int PP, int QP>
__device__ __forceinline__ void tq_read_all_cond() {
  asm volatile(TQ_RD("v[152:153]", "v15", "0") TQ_RD("v[154:155]", "v15", "%1") TQ_RD("v[156:157]", "v15", "64") TQ_RD("v[158:159]", "v15", "%2")
               TQ_RD("v[144:145]", "v14", "0") TQ_RD("v[146:147]", "v14", "%0")
               : : "n"(4 * PP), "n"(4 * QP), "n"(4 * QP + 64) : TQ_BANK);
}
__device__ __forceinline__ void tq_set_ones(unsigned v) {
  asm volatile("v_mov_b32 v160, %0\n\tv_mov_b32 v161, %0\n\tv_mov_b32 v162, %0\n\tv_mov_b32 v163, %0\n\ts_nop 1" : : "v"(v) : TQ_BANK);
}
template <bool F16, int QP>
__device__ __forceinline__ void tq_c1() {
#define TQ_C1(MF)                                                                                                               \
  asm volatile("s_waitcnt lgkmcnt(0)\n\t" TQ_MFMA(MF, TQ_T0, TQ_A0, TQ_B0) TQ_RD("v[152:153]", "v15", "0") TQ_RD("v[154:155]", "v15", "%0") \
               TQ_MFMA(MF, TQ_T1, TQ_A0, TQ_B1) TQ_RD("v[156:157]", "v15", "64") TQ_RD("v[158:159]", "v15", "%1")               \
               : : "n"(4 * QP), "n"(4 * QP + 64) : TQ_BANK)
  if constexpr (F16) { TQ_C1("f16"); } else { TQ_C1("bf16"); }
#undef TQ_C1
}
template <bool F16, int PP>
__device__ __forceinline__ void tq_c2() {
#define TQ_C2(MF) \
  asm volatile(TQ_MFMA(MF, TQ_T2, TQ_A0, TQ_B2) TQ_RD("v[144:145]", "v14", "0") TQ_RD("v[146:147]", "v14", "%0") : : "n"(4 * PP) : TQ_BANK)
  if constexpr (F16) { TQ_C2("f16"); } else { TQ_C2("bf16"); }
#undef TQ_C2
}


#ifdef WAE_TQ_STAMPS
__device__ __forceinline__ long long tq_clock() {
  long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#endif

// (device pass only: the buffer-descriptor type does not exist in the host pass, which still parses device templates)
#if defined(__HIP_DEVICE_COMPILE__)
// ---- one (segment, job): prologue, half-slab loop, flush ----------------------------------------------------------------------
// The list of a segment's USEFUL half-slabs (16 rows) is a sequence of per-clip runs: clip b contributes its 32-row slabs
// [u_lo, u_hi) (slabs whose every row pairs with a Q row before the clip contribute nothing and are skipped), cut by the segment's
// [slab_begin, slab_end).  Two cursors walk it: the request cursor (D half-slabs ahead; owns the buffer descriptors and the scalar
// row offsets) and the contraction cursor (a count; COND also tracks the clip for its ones operand).  Everything a cursor needs at
// a clip boundary is re-read from the job table there (a rare, slow path) instead of living in SGPRs across the loop.
// (State in one struct and plain functions over it: closures that capture other closures by reference survived into the final IR
//  as scratch objects -- descriptors loaded from scratch are VGPR values, and every request became a waterfall loop.)
template <int KIND>
struct TqState {
  static constexpr int PMAX = TqGeo<KIND>::PIECES_MAX;
  tq_i32x4 srd[PMAX];                 // per piece: the sub-image's descriptor for the clip being requested
  unsigned ro[PMAX], inc[PMAX];       // scalar row offset of the next request / its step per half-slab (bytes)
  int rq_b, rq_total, rq_run;         // request cursor: clip, useful half-slabs behind the current run, left in the current run
  int per_clip;                       // useful half-slabs of a whole clip
  unsigned dma_dst;                   // LDS address of this wave's first piece in the ring slot of the next request
  int rd_slot;                        // ring slot the fragments are being read from
  int cc_b, cc_run;                   // COND: clip of the contraction cursor / half-slabs left in it
  // team pacing (the wave that carries it: tq_run<..., PACE>)
  int rq_pos, pos_inc;                // absolute position of the next request in the team's list of half-slabs / its step (0 once dead)
  int window;
};
// Team pacing.  The members of a team stream the same time range of one layer, and four of them read the same dz half-slabs -- but an
// XCD's 4 MiB L2 is shared by six teams and holds about ten half-slabs of everything a team streams, while unpaced members drift
// apart by hundreds (COND has a third of the taps' MFMA work and runs ahead; the taps drift at random): every member then fetches
// its own copy from HBM, and HBM bytes are what bounds this launch (requests alone: 1.17 ms of 1.5).  A team has ONE 32-bit word:
// three 10-bit fields, the request positions of the tap members in units of 32-row slabs.  Once per half-slab a paced member's
// wave 11 issues one returning atomic add on it -- a tap adds its progress to its field, COND adds zero -- and so learns the
// others' positions; the returned word is looked at a few half-slabs later, when the counted wait at the loop top has retired it,
// and the wave holds its workgroup in front of the barrier while its cursor is more than `window` slabs ahead of the slowest OTHER
// tap.  The atomic takes the place of a request in the wave's operation count (wave 11 has one piece fewer than the variant it
// runs) and lives in the two registers that piece would use: v12 data, v13 the returned word.  Timing only: results never depend
// on it; the slowest member never waits; a member first publishes, then waits; a wait that does not end switches pacing off.
struct TqPace {
  unsigned* word;   // the team's word, or null
  int shift;        // 10 * member for a tap, -1 for COND (reads only)
  int pub;          // position (slabs) last published by this member
  int ntaps;
  int next;         // request position (half-slabs) at which the next atomic is due
  unsigned dump;    // LDS address of a 1-KiB area nobody reads: where the filler request of the other half-slabs lands
  int dbg_holds, dbg_spins, dbg_timeouts, dbg_pos, dbg_word, dbg_tword, dbg_tpub;   // diagnostics (written to `stamps` when passed)
};
#define TQ_PACE_INF 1023
__device__ __forceinline__ void tq_pace_init_ret() { asm volatile("v_mov_b32 v13, -1" : : : TQ_BANK); }   // nobody is behind, until the first word arrives
// one returning add of this member's unpublished progress: v13 <- the word before it.  ONE lane adds (a wave-wide atomic is 64
// adds): exec = 1 around it; v_readfirstlane reads lane 0 under full exec.
template <bool SYNC>
__device__ __forceinline__ void tq_pace_add(TqPace& pc, int pos_slabs) {
  int delta = pos_slabs - pc.pub;
  delta = __builtin_amdgcn_readfirstlane((delta > 0 && pc.shift >= 0) ? delta : 0);
  pc.pub += delta;
  const unsigned d = __builtin_amdgcn_readfirstlane((unsigned)delta << (pc.shift >= 0 ? pc.shift : 0));
  unsigned long long save;
  if constexpr (SYNC)
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v12, %1\n\t"
                 "global_atomic_add v13, v10, v12, %2 sc0\n\ts_waitcnt vmcnt(0)\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(d), "s"(pc.word) : TQ_BANK);
  else
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tv_mov_b32 v10, 0\n\tv_mov_b32 v12, %1\n\t"
                 "global_atomic_add v13, v10, v12, %2 sc0\n\ts_mov_b64 exec, %0"
                 : "=&s"(save) : "s"(d), "s"(pc.word) : TQ_BANK);
}
// Every TQ_PACE_EVERY-th half-slab only: a returning atomic stays in the wave's in-order queue for ~2.3 us under this launch's
// traffic, longer than the ~2 us the counted wait at the loop top gives it -- one per half-slab cost 0.5 ms of 1.5.  The half-slabs
// in between fill the operation slot with a request whose every lane is out of range: zeros into the dump area, no memory traffic.
#define TQ_PACE_EVERY 4
__device__ __forceinline__ void tq_pace_step(TqPace& pc, int pos) {
  if (pos >= pc.next) {
    pc.next = pos + TQ_PACE_EVERY;
    tq_pace_add<false>(pc, pos >> 1);
  } else {
    const tq_i32x4 none = {0, 0, 0, 0x00020000};   // num_records 0: everything is out of range
    asm volatile("s_mov_b32 m0, %1\n\tv_mov_b32 v10, 0\n\tbuffer_load_dwordx4 v10, %0, 0 offen lds" : : "s"(none), "s"(pc.dump) : "m0", TQ_BANK);
  }
}
__device__ __forceinline__ int tq_pace_lead(TqPace& pc) {   // slowest OTHER tap's position in the word last returned to v13
  unsigned w;
  asm volatile("v_readfirstlane_b32 %0, v13" : "=s"(w) : : TQ_BANK);
  // (a member's own field lags its cursor -- by the atomics in flight, and by a whole clip head right after the cursor has skipped
  //  one -- and a member that holds does not publish: a lead that includes the own field can wait for itself for ever)
  const int me = pc.shift >= 0 ? pc.shift / 10 : -1;
  int lead = TQ_PACE_INF;
  if (me != 0) lead = min(lead, (int)(w & 1023u));
  if (pc.ntaps > 1 && me != 1) lead = min(lead, (int)((w >> 10) & 1023u));
  if (pc.ntaps > 2 && me != 2) lead = min(lead, (int)((w >> 20) & 1023u));
  pc.dbg_word = (int)w;
  return lead;
}
__device__ __forceinline__ void tq_pace_hold(TqPace& pc, int pos_slabs, int& window) {
  int lead = tq_pace_lead(pc);
  if (pos_slabs - lead <= window) return;
  ++pc.dbg_holds;
  // slow path (this wave is waiting anyway; vmcnt(0) only makes the counted waits that follow stricter): first make the own
  // position visible -- the members this one waits for may be waiting for it --, then poll (with the atomic, not a load: a load,
  // even sc1, is served by this XCD's L2)
  tq_pace_add<true>(pc, pos_slabs);
  lead = tq_pace_lead(pc);
  int spins = 0;
  while (pos_slabs - lead > window) {
    ++pc.dbg_spins;
    __builtin_amdgcn_s_sleep(8);
    tq_pace_add<true>(pc, pos_slabs);
    lead = tq_pace_lead(pc);
    if (++spins > 4096) {   // ~10 ms: give up pacing, never hang
      window = 0x7fffffff; ++pc.dbg_timeouts; pc.dbg_pos = pos_slabs; pc.dbg_tword = pc.dbg_word; pc.dbg_tpub = pc.pub;
      break;
    }
  }
}
template <int KIND>
__device__ __forceinline__ void tq_rq_next_run(TqState<KIND>& st, const TqArgs& p, const TqJob* jp, int wave, int b, int t_start, int run) {
  using Dr = TqDer<KIND>;
  const TqJobS jb = tq_load_job(jp);   // slow path: descriptors and row offsets of clip b from the job record
#pragma unroll
  for (int j = 0; j < TqState<KIND>::PMAX; ++j) {
    const int pc = wave + TQ_NW * j;
    const int sub = pc < Dr::NPP ? 0 : (pc < Dr::NPP + Dr::NPQ ? 1 : 2);
    const unsigned sb = tq_sel3(sub, jb.sbP, jb.sbQ0, jb.sbQ1);
    const int valid = tq_sel3(sub, jb.m_valid, jb.n0_valid, jb.n1_valid);
    char* base = const_cast<char*>(tq_sel3(sub, jb.P, jb.Q0, jb.Q1));
    const unsigned nrec = (valid > 0 && base != nullptr && b < p.B) ? (unsigned)(p.T - 1) * sb + (unsigned)valid * 2 : 0u;
    // raw buffer descriptor over [clip base, + nrec): stride 0, offsets checked against num_records (out of range: zeros)
    const unsigned long long a = (unsigned long long)(base + (int64_t)b * p.T * (int64_t)sb);
    tq_i32x4 d;
    d.x = (int)(unsigned)a; d.y = (int)((unsigned)(a >> 32) & 0xffffu); d.z = (int)nrec; d.w = 0x00020000;
    st.srd[j] = d;
    st.ro[j] = (unsigned)((t_start + (sub == 0 ? 0 : jb.shift)) * (int)sb);
    st.inc[j] = 16u * sb;
  }
  st.rq_total -= run;
  st.rq_run = run;
}
template <int KIND>
__device__ __forceinline__ void tq_rq_boundary(TqState<KIND>& st, const TqArgs& p, const TqJob* jp, int wave) {
  if (st.rq_total > 0) {   // the run of the request cursor has ended: the next clip ...
    ++st.rq_b;
    st.rq_pos += 2 * p.spc - st.per_clip;   // the skipped slabs at the head of the next clip
    tq_rq_next_run<KIND>(st, p, jp, wave, st.rq_b, p.spc * 32 - (st.per_clip >> 1) * 32, min(st.rq_total, st.per_clip));
  } else {                 // ... or the dead state behind the segment's end: every lane out of range (zeros into the ring), forever
#pragma unroll
    for (int j = 0; j < TqState<KIND>::PMAX; ++j) { st.ro[j] = TQ_DEAD; st.inc[j] = 0; }
    st.rq_run = 0x7fffffff;
    st.pos_inc = 0;
  }
}
// timing-only ablations (tools/ablate_tq.sh; results are wrong when any bit is set): 1 no requests, 2 no MFMA / fragment streams
#ifndef WAE_TQ_ABL
#define WAE_TQ_ABL 0
#endif
// pieces [J0, min(J1, NPW)) of the half-slab under the request cursor; PACE: the wave's last operation is the pacing atomic
template <int KIND, int NPW, bool PACE, int J0 = 0, int J1 = 3>
__device__ __forceinline__ void tq_request_some(TqState<KIND>& st, TqPace& pc) {
  if constexpr (WAE_TQ_ABL & 1) return;
  constexpr int NR = PACE ? NPW - 1 : NPW;   // real requests
  if constexpr (J0 <= 0 && 0 < J1 && NR > 0) tq_request<0>(st.srd[0], st.ro[0], st.dma_dst);
  if constexpr (J0 <= 1 && 1 < J1 && NR > 1) tq_request<1>(st.srd[1], st.ro[1], st.dma_dst + TQ_NW * 1024);
  if constexpr (J0 <= 2 && 2 < J1 && NR > 2) tq_request<2>(st.srd[2], st.ro[2], st.dma_dst + 2 * TQ_NW * 1024);
  if constexpr (PACE && J0 <= NPW - 1 && NPW - 1 < J1) tq_pace_step(pc, st.rq_pos);
}
template <int KIND>
__device__ __forceinline__ void tq_step_slots(TqState<KIND>& st, unsigned ring_end) {   // behind a half-slab's requests
  constexpr int NS = TqGeo<KIND>::NS, SLOT = TqDer<KIND>::SLOT;
#pragma unroll
  for (int j = 0; j < TqState<KIND>::PMAX; ++j) st.ro[j] += st.inc[j];
  st.dma_dst = st.dma_dst + SLOT >= ring_end ? st.dma_dst - (NS - 1) * SLOT : st.dma_dst + SLOT;
  st.rq_pos += st.pos_inc;
}

// NPW: pieces of this wave per half-slab; MODE 0: no valid tile (requests and barriers only), 1: contracts, 2: contracts + column
// sums.  One instantiation per combination: no wave-uniform branch around the asm streams inside the loop.
template <typename E, int KIND, int NPW, int MODE, bool PACE = false>
__device__ __forceinline__ void tq_run(TqState<KIND>& st, TqPace& pc, const TqArgs& p, const TqJob* jp, unsigned ring_end, int wave, int lane, int nh) {
  using G = TqGeo<KIND>;
  using Dr = TqDer<KIND>;
  constexpr bool F16 = ET<E>::DT == WAE_F16;
  constexpr int PP = G::PP, QP = G::QP, NS = G::NS, SLOT = Dr::SLOT, D = Dr::D;
  constexpr unsigned ONES = F16 ? 0x3c003c00u : 0x3f803f80u;
  constexpr bool active = MODE > 0 && !(WAE_TQ_ABL & 2), bias_wave = MODE == 2;
  float bs0 = 0.f, bs1 = 0.f, bs2 = 0.f, bs3 = 0.f;   // MODE 2: column sums of Q0 (per lane: 8 of the 16 k of every half-slab)
  // ---- prologue: half-slabs 0 .. D-1 requested, the first one visible, its fragments requested -------------------------------
  if constexpr (PACE) tq_pace_init_ret();
#pragma unroll 1
  for (int h = 0; h < D; ++h) {
    tq_request_some<KIND, NPW, PACE>(st, pc);
    tq_step_slots<KIND>(st, ring_end);
    if (--st.rq_run == 0) tq_rq_boundary<KIND>(st, p, jp, wave);
  }
  tq_wait_vm<(D - 1) * NPW>();
  __builtin_amdgcn_s_barrier();
  if constexpr (active) {
    if constexpr (KIND == TQ_COND) tq_read_all_cond<PP, QP>();
    else tq_read_all_2x4<PP, QP>();
  }
  tq_step_read_slot(SLOT);
  st.rd_slot = 1;   // the fragments of the NEXT half-slab are read from slot 1
  int n_left = nh;
#ifdef WAE_TQ_STAMPS
  long long k_wait = 0, k_bar = 0, k_body = 0, k_iter = 0;
#endif
  // The hot loop runs over CHUNKS: stretches of half-slabs in which neither cursor meets a clip boundary.  What happens at a
  // boundary (descriptors of the next clip from the job table, the ones operand of the next clip) sits between the chunks; the
  // pipeline state -- fragments and pieces in flight -- is untouched by it (a table read there drains vmcnt: stricter, never wrong).
#pragma unroll 1
  for (;;) {
    int chunk = min(n_left, st.rq_run);
    if constexpr (KIND == TQ_COND) chunk = min(chunk, st.cc_run);
    n_left -= chunk; st.rq_run -= chunk;
    if constexpr (KIND == TQ_COND) st.cc_run -= chunk;
#pragma unroll 1
    for (; chunk > 0; --chunk) {
#ifdef WAE_TQ_STAMPS
      const long long s0 = tq_clock();
#endif
      tq_wait_vm<(D - 2) * NPW>();   // this wave's pieces of the next half-slab have landed; the younger D - 2 stay in flight
      if constexpr (PACE) tq_pace_hold(pc, st.rq_pos >> 1, st.window);
#ifdef WAE_TQ_STAMPS
      const long long s1 = tq_clock();
#endif
      __builtin_amdgcn_s_barrier();
#ifdef WAE_TQ_STAMPS
      const long long s2 = tq_clock();
#endif
      if constexpr (KIND == TQ_COND) {
        if constexpr (active) tq_c1<F16, QP>();
        tq_request_some<KIND, NPW, PACE>(st, pc);
        if constexpr (active) tq_c2<F16, PP>();
      } else {
#if defined(WAE_TQ_REQ) && WAE_TQ_REQ == 1      // experiment: every request behind the wave's last MFMA
        if constexpr (active) tq_s1<F16>();
        if constexpr (active) tq_s2<F16, PP>();
        if constexpr (active) tq_s3<F16, QP, bias_wave>(bs0, bs1);
        if constexpr (active) tq_s4<F16, PP, QP, bias_wave>(bs2, bs3);
        tq_request_some<KIND, NPW, PACE>(st, pc);
#elif defined(WAE_TQ_REQ) && WAE_TQ_REQ == 2    // experiment: every request in front of the wave's first MFMA
        tq_request_some<KIND, NPW, PACE>(st, pc);
        if constexpr (active) tq_s1<F16>();
        if constexpr (active) tq_s2<F16, PP>();
        if constexpr (active) tq_s3<F16, QP, bias_wave>(bs0, bs1);
        if constexpr (active) tq_s4<F16, PP, QP, bias_wave>(bs2, bs3);
#else
        if constexpr (active) tq_s1<F16>();
        tq_request_some<KIND, NPW, PACE, 0, 1>(st, pc);
        if constexpr (active) tq_s2<F16, PP>();
        tq_request_some<KIND, NPW, PACE, 1, 2>(st, pc);
        if constexpr (active) tq_s3<F16, QP, bias_wave>(bs0, bs1);
        tq_request_some<KIND, NPW, PACE, 2, 3>(st, pc);
        if constexpr (active) tq_s4<F16, PP, QP, bias_wave>(bs2, bs3);
#endif
      }
      tq_step_slots<KIND>(st, ring_end);
      const bool wrap = st.rd_slot + 1 == NS;
      st.rd_slot = wrap ? 0 : st.rd_slot + 1;
      tq_step_read_slot(wrap ? -(NS - 1) * SLOT : SLOT);
#ifdef WAE_TQ_STAMPS
      asm volatile("s_nop 0" ::: "memory");
      const long long s3 = tq_clock();
      k_wait += s1 - s0; k_bar += s2 - s1; k_body += s3 - s2; ++k_iter;
#endif
    }
    if (n_left == 0) break;
    if (st.rq_run == 0) tq_rq_boundary<KIND>(st, p, jp, wave);
    if constexpr (KIND == TQ_COND) {
      if (st.cc_run == 0) {   // the next half-slab belongs to another clip: its sums go to that clip's column
        ++st.cc_b;
        st.cc_run = st.per_clip;
        tq_set_ones((lane & 31) == st.cc_b ? ONES : 0u);
      }
    }
  }
#ifdef WAE_TQ_STAMPS
  if (p.stamps && lane == 0) {   // [workgroup][wave 0..15][wait, barrier, body, half-slabs]; wave 15's row: [life ticks, life 10 ns, -, segments]
    long long* o = p.stamps + ((long long)blockIdx.x * 16 + wave) * 4;
    o[0] += k_wait; o[1] += k_bar; o[2] += k_body; o[3] += k_iter;
    if (wave == 0) p.stamps[((long long)blockIdx.x * 16 + 15) * 4 + 3] += 1;
  }
#endif
  if constexpr (bias_wave) {   // Cb[n] += alpha * (column sums of Q0): this wave's four N-tiles, both lane halves (8 k each)
    const TqJobS jb = tq_load_job(jp);
    const int col = (wave / 3) * 128 + (lane & 31);
    if (col < jb.n0_valid) atomicAdd(jb.Cb + col, jb.alpha * bs0);
    if (col + 32 < jb.n0_valid) atomicAdd(jb.Cb + col + 32, jb.alpha * bs1);
    if (col + 64 < jb.n0_valid) atomicAdd(jb.Cb + col + 64, jb.alpha * bs2);
    if (col + 96 < jb.n0_valid) atomicAdd(jb.Cb + col + 96, jb.alpha * bs3);
  }
}

template <int K, int Q>
__device__ __forceinline__ void tq_flush_quad(float* C, int ldc, int row0, int col, int m_valid, float alpha, int hh) {
  float t[4];
  tq_get_quad<K, Q>(t);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + r + 8 * Q + 4 * hh;
    if (row < m_valid) atomicAdd(C + row * ldc + col, alpha * t[r]);
  }
}
// one tile at a time, four registers at a time: the compiler has sixteen
template <int K>
__device__ __forceinline__ void tq_flush_tile(float* C, int ldc, int row0, int col, bool col_ok, int m_valid, float alpha, int hh) {
  if (!col_ok || !(row0 < m_valid)) return;
  tq_flush_quad<K, 0>(C, ldc, row0, col, m_valid, alpha, hh);
  tq_flush_quad<K, 1>(C, ldc, row0, col, m_valid, alpha, hh);
  tq_flush_quad<K, 2>(C, ldc, row0, col, m_valid, alpha, hh);
  tq_flush_quad<K, 3>(C, ldc, row0, col, m_valid, alpha, hh);
}

template <typename E, int KIND>
__device__ __forceinline__ void tq_segment(const TqArgs& p, const TqJob* jp, const int slab_begin, const int slab_end, char* smem,
                                           const unsigned lds0, const int wave, TqPace& pc, const int pos_base) {
  using G = TqGeo<KIND>;
  using Dr = TqDer<KIND>;
  constexpr bool F16 = ET<E>::DT == WAE_F16;
  constexpr int PP = G::PP, QP = G::QP, NPP = Dr::NPP, NPQ = Dr::NPQ;
  constexpr int PMAX = G::PIECES_MAX;
  constexpr unsigned ONES = F16 ? 0x3c003c00u : 0x3f803f80u;

  const int lane = threadIdx.x & 63;
  const int npw = (Dr::NPIECES - wave + TQ_NW - 1) / TQ_NW;   // pieces of this wave (PMAX or PMAX - 1)
  int wm, wn;
  if constexpr (KIND == TQ_TAPS) { wm = (wave % 6) * 64; wn = (wave / 6) * 128; }
  else if constexpr (KIND == TQ_OUTSKIP) { wm = (wave % 3) * 64; wn = (wave / 3) * 128; }
  else { wm = wave * 32; wn = 0; }
  const int qsub = (KIND == TQ_OUTSKIP && wn >= 256) ? 1 : 0;
  const int wnl = wn - 256 * qsub;

  // ---- setup from the job record (its scalars die here; the flush and the clip boundaries read the record again) ----------------
  TqState<KIND> st;
  unsigned vb[3] = {TQ_OOB, TQ_OOB, TQ_OOB};   // per lane and piece: row * stride + column bytes inside a half-slab, or TQ_OOB
  bool active, bias_wave;
  int nh;                     // useful half-slabs of this (segment, job)
  int first_b, first_t, first_run;
  {
    const TqJobS jb = tq_load_job(jp);
    // (COND: a wave's third tile -- the per-clip sums of its 32 P columns -- is always wanted: n0_valid = 0 makes a column-sum job)
    active = wm < jb.m_valid && (KIND == TQ_COND || wnl < tq_sel3(qsub, jb.n0_valid, jb.n1_valid, 0));
#ifdef WAE_TQ_NOBIAS   // timing experiment: no column sums (the out bias gradient is then missing)
    bias_wave = false;
#else
    bias_wave = KIND == TQ_OUTSKIP && jb.Cb != nullptr && wm == 0 && qsub == 0 && active;
#endif
#pragma unroll
    for (int j = 0; j < PMAX; ++j) {
      const int pc = wave + TQ_NW * j;
      const int sub = pc < NPP ? 0 : (pc < NPP + NPQ ? 1 : 2);
      const int u = (pc - (sub == 0 ? 0 : (sub == 1 ? NPP : NPP + NPQ))) * 64 + lane;
      const int units = sub == 0 ? Dr::PU : Dr::QU;
      const int row = u / units, col = u - row * units;
      const int valid = tq_sel3(sub, jb.m_valid, jb.n0_valid, jb.n1_valid);
      const unsigned sb = tq_sel3(sub, jb.sbP, jb.sbQ0, jb.sbQ1);
      vb[j] = (col * 8 < valid && pc < Dr::NPIECES) ? (unsigned)row * sb + (unsigned)col * 16 : TQ_OOB;
    }
    const int u_lo = max(0, (-jb.shift) >> 5);     // (shift <= 0 on every job of this path; host check)
    const int u_hi = p.spc;
    st.per_clip = 2 * (u_hi - u_lo);
    // useful 32-row slabs among the linear slab numbers [0, x)
    const int be = slab_end / p.spc, re = slab_end - be * p.spc, bb = slab_begin / p.spc, rb = slab_begin - bb * p.spc;
    const int ce = be * (u_hi - u_lo) + min(max(re - u_lo, 0), u_hi - u_lo), cb = bb * (u_hi - u_lo) + min(max(rb - u_lo, 0), u_hi - u_lo);
    nh = 2 * (ce - cb);
    first_b = bb;
    int r = rb;
    if (r < u_lo) r = u_lo;
    if (r >= u_hi) { ++first_b; r = u_lo; }
    first_t = r * 32;
    first_run = min(nh, 2 * (u_hi - r));
  }

  __syncthreads();   // every wave is done with the previous segment's ring
  if (nh <= 0) return;   // (workgroup-uniform) nothing useful in this segment for this job

  st.rq_b = first_b; st.rq_total = nh; st.rq_run = 0;
  tq_rq_next_run<KIND>(st, p, jp, wave, first_b, first_t, first_run);
  st.dma_dst = lds0 + (unsigned)wave * 1024u;
  const unsigned ring_end = lds0 + (unsigned)(G::NS * Dr::SLOT) + (unsigned)wave * 1024u;
  tq_zero_acc();
  {   // transposed-read lane geometry (csrc/gemm_tn.hip tn_load_frags); the addresses run over the ring with the slot being READ
    const int hh2 = lane >> 5, grp = (lane >> 4) & 1, q4 = (lane & 15) >> 2, pq = lane & 3;
    const unsigned lane_row = 8 * hh2 + q4, lane_col = (16 * grp + 4 * pq) * 2;
    const unsigned ap = lds0 + lane_row * PP + lane_col + wm * 2;
    const unsigned bp = lds0 + (NPP + qsub * NPQ) * 1024 + lane_row * QP + lane_col + wnl * 2;
    tq_set_lane_consts(vb[0], vb[1], vb[2], ap, bp);
  }
  st.rd_slot = 0;
  st.cc_b = first_b; st.cc_run = first_run;   // COND: the clip whose column the ones operand feeds
  if constexpr (KIND == TQ_COND) tq_set_ones((lane & 31) == st.cc_b ? ONES : 0u);

  st.rq_pos = pos_base + 2 * (first_b * p.spc + (first_t >> 5) - slab_begin);
  st.pos_inc = 1;
  st.window = KIND == TQ_COND ? p.window_cond : p.window;
  // the wave that paces its workgroup against the team: wave 11, which in TAPS / COND has one piece fewer than PMAX -- the pacing
  // atomic fills the free slot of the PMAX variant's operation count
  const bool pacer = KIND != TQ_OUTSKIP && PMAX == 2 && pc.word != nullptr && wave == TQ_NW - 1 && npw == PMAX - 1;
  if (pacer) {
    if constexpr (KIND != TQ_OUTSKIP) {
      if (!active) tq_run<E, KIND, PMAX, 0, true>(st, pc, p, jp, ring_end, wave, lane, nh);
      else tq_run<E, KIND, PMAX, 1, true>(st, pc, p, jp, ring_end, wave, lane, nh);
    }
  } else if (npw == PMAX) {
    if (!active) tq_run<E, KIND, PMAX, 0>(st, pc, p, jp, ring_end, wave, lane, nh);
    else if (KIND == TQ_OUTSKIP && bias_wave) tq_run<E, KIND, PMAX, (KIND == TQ_OUTSKIP ? 2 : 1)>(st, pc, p, jp, ring_end, wave, lane, nh);
    else tq_run<E, KIND, PMAX, 1>(st, pc, p, jp, ring_end, wave, lane, nh);
  } else {
    if (!active) tq_run<E, KIND, PMAX - 1, 0>(st, pc, p, jp, ring_end, wave, lane, nh);
    else if (KIND == TQ_OUTSKIP && bias_wave) tq_run<E, KIND, PMAX - 1, (KIND == TQ_OUTSKIP ? 2 : 1)>(st, pc, p, jp, ring_end, wave, lane, nh);
    else tq_run<E, KIND, PMAX - 1, 1>(st, pc, p, jp, ring_end, wave, lane, nh);
  }

  // the fragments requested for the half-slab behind the last one, and the zero-fill requests behind the segment's end
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  // the MFMAs above are opaque to the compiler's hazard recogniser: cover the MFMA-result -> VALU-read wait states here
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  // ---- C += alpha * acc   (lane = column n, registers = rows m); fp32 atomics: other workgroups own other time ranges -----------
  if (!active) return;
  const TqJobS jb = tq_load_job(jp);
  int lne = threadIdx.x & 63;
  asm volatile("" : "+v"(lne));   // the output addresses are formed here, not hoisted above the loop
  const int nl = lne & 31, hh = lne >> 5;
  if constexpr (KIND == TQ_COND) {
    tq_flush_tile<0>(jb.C0, jb.ldc0, wm, nl, nl < jb.n0_valid, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<1>(jb.C0, jb.ldc0, wm, 32 + nl, 32 + nl < jb.n0_valid, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<2>(jb.C0, jb.ldc0, wm, jb.ones_col + nl, nl < p.B, jb.m_valid, jb.alpha, hh);
  } else {
    float* Cq = tq_sel3(qsub, jb.C0, jb.C1, (float*)nullptr);
    const int ldc = tq_sel3(qsub, jb.ldc0, jb.ldc1, 0);
    const int nv = tq_sel3(qsub, jb.n0_valid, jb.n1_valid, 0);
    tq_flush_tile<0>(Cq, ldc, wm, wnl + nl, wnl + nl < nv, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<1>(Cq, ldc, wm, wnl + 32 + nl, wnl + 32 + nl < nv, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<2>(Cq, ldc, wm, wnl + 64 + nl, wnl + 64 + nl < nv, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<3>(Cq, ldc, wm, wnl + 96 + nl, wnl + 96 + nl < nv, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<4>(Cq, ldc, wm + 32, wnl + nl, wnl + nl < nv, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<5>(Cq, ldc, wm + 32, wnl + 32 + nl, wnl + 32 + nl < nv, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<6>(Cq, ldc, wm + 32, wnl + 64 + nl, wnl + 64 + nl < nv, jb.m_valid, jb.alpha, hh);
    tq_flush_tile<7>(Cq, ldc, wm + 32, wnl + 96 + nl, wnl + 96 + nl < nv, jb.m_valid, jb.alpha, hh);
  }
}


#endif   // __HIP_DEVICE_COMPILE__

template <typename E>
__global__ void __launch_bounds__(TQ_NW * 64, 1) __attribute__((amdgpu_num_vgpr(5)))   // 10 compiler registers (the attribute counts in units of two on gfx950) + the bank
gemm_tn_static_kernel(TqArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  // Teams.  Workgroups are dealt round-robin over the 8 XCDs in launch order (speed only, never correctness): workgroup b runs on
  // XCD b & 7 as that XCD's (b >> 3)-th workgroup.  A team's members share operands through their XCD's L2, so a team is built from
  // workgroups of ONE XCD: per XCD floor(32 / team_size) whole teams; the XCDs' leftover workgroups (2 each at team_size 5) form
  // the last few teams across XCDs -- they do their share of the work, without the sharing.
  int team, member;
  bool one_xcd = false;   // the team's members share an L2
  if ((gridDim.x & 7) == 0) {
    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int whole = per_xcd / p.team_size, rest = per_xcd - whole * p.team_size;
    if (slot < whole * p.team_size) {
      team = xcd * whole + slot / p.team_size;
      member = slot % p.team_size;
      one_xcd = true;
    } else {
      const int left = xcd * rest + (slot - whole * p.team_size);
      team = 8 * whole + left / p.team_size;
      member = left % p.team_size;
    }
  } else {
    team = blockIdx.x / p.team_size;
    member = blockIdx.x - team * p.team_size;
  }
  if (team >= p.nteams) return;
  long long k_w0 = 0;   // diagnostics: start of this workgroup's life (100 MHz ticks), read only when the caller passes `stamps`
  if (p.stamps) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(k_w0) : : "memory");
#ifdef WAE_TQ_STAMPS
  const long long k_t0 = tq_clock();
#endif
  const int seg_b = p.team_seg[team], seg_e = p.team_seg[team + 1];
  TqPace pc;
  pc.word = (p.pace != nullptr && p.window > 0 && one_xcd && member <= p.ntaps && p.ntaps <= 3) ? p.pace + team : nullptr;
  pc.shift = member < p.ntaps ? 10 * member : -1; pc.pub = 0; pc.ntaps = p.ntaps; pc.next = 0;
  pc.dump = lds0 + (unsigned)p.dump_off;
  pc.dbg_holds = pc.dbg_spins = pc.dbg_timeouts = pc.dbg_pos = pc.dbg_word = pc.dbg_tword = pc.dbg_tpub = 0;
  int pos_base = 0;   // positions of the team's list of half-slabs: segments back to back, 2 per 32-row slab
#pragma unroll 1
  for (int si = seg_b; si < seg_e; ++si) {
    const int* q = (const int*)(p.segs + si);
    const int job0 = tq_ldw(q, 0), slab_begin = tq_ldw(q, 1), slab_end = tq_ldw(q, 2);
    const TqJob* jp = p.jobs + job0 + member;
    const int* jq = (const int*)jp;
    const int m_valid = tq_ldw(jq, 22), kind = tq_ldw(jq, 27);
    if (m_valid <= 0) { pos_base += 2 * (slab_end - slab_begin); continue; }   // null job
    if (kind == TQ_TAPS) tq_segment<E, TQ_TAPS>(p, jp, slab_begin, slab_end, smem, lds0, wave, pc, pos_base);
    else if (kind == TQ_COND) tq_segment<E, TQ_COND>(p, jp, slab_begin, slab_end, smem, lds0, wave, pc, pos_base);
    else tq_segment<E, TQ_OUTSKIP>(p, jp, slab_begin, slab_end, smem, lds0, wave, pc, pos_base);
    pos_base += 2 * (slab_end - slab_begin);
  }
  // this member is done: its field goes to "infinitely far ahead" (nobody waits for it any more)
  if (pc.word != nullptr && threadIdx.x == (TQ_NW - 1) * 64) {
    if (pc.shift >= 0) __hip_atomic_fetch_add(pc.word, (unsigned)(TQ_PACE_INF - pc.pub) << pc.shift, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifndef WAE_TQ_STAMPS
    if (p.stamps) {   // pacing diagnostics of this workgroup: [holds, spins, time-outs, position / published at the last time-out, word then]
      long long* o = p.stamps + (long long)blockIdx.x * 8;
      o[0] = pc.dbg_holds; o[1] = pc.dbg_spins; o[2] = pc.dbg_timeouts; o[3] = ((long long)pc.dbg_pos << 32) | (unsigned)pc.dbg_tpub; o[4] = (unsigned)pc.dbg_tword;
    }
#endif
  }
#ifndef WAE_TQ_STAMPS
  if (p.stamps && threadIdx.x == 0) {   // [.., team, member, life in 10-ns ticks] of EVERY workgroup (tools/pace_debug.py)
    long long* o = p.stamps + (long long)blockIdx.x * 8;
    long long w1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1) : : "memory");
    o[5] = team; o[6] = member; o[7] = w1 - k_w0;
  }
#endif
#ifdef WAE_TQ_STAMPS
  if (p.stamps && threadIdx.x == 0) {
    long long* o = p.stamps + ((long long)blockIdx.x * 16 + 15) * 4;
    long long w1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1) : : "memory");
    o[0] = tq_clock() - k_t0; o[1] = w1 - k_w0;
  }
#endif
#endif   // __HIP_DEVICE_COMPILE__
}

extern "C" int wae_gemm_tn_static(int32_t dtype, const wae_tq_job* jobs_dev, const wae_ts_seg* segs_dev, const int32_t* team_seg_dev,
                                  int32_t nteams, int32_t team_size, int32_t nwg, int32_t B, int32_t T, int64_t* stamps, uint32_t* pace,
                                  int32_t window, int32_t window_cond, int32_t ntaps, int64_t max_clip_bytes, void* stream) {
  WAE_REQUIRE(dtype == WAE_BF16 || dtype == WAE_F16, "gemm_tn_static: 16-bit operands only");
  // the zero fill outside a clip is the buffer descriptor's range check: offsets of rows before / behind a clip must land beyond
  // num_records and below the TQ_OOB / TQ_DEAD markers, which holds while every operand clip (reach x row bytes) stays below 2^30
  WAE_REQUIRE(max_clip_bytes > 0 && max_clip_bytes < (1ll << 30),
              "gemm_tn_static: an operand clip of 2^30 bytes or more (or max_clip_bytes not stated): use wae_gemm_tn_stream");
  WAE_REQUIRE(jobs_dev && segs_dev && team_seg_dev && nteams > 0 && team_size > 0 && nwg >= nteams * team_size && B > 0 && T > 0,
              "gemm_tn_static: bad arguments");
  WAE_REQUIRE(B <= 32, "gemm_tn_static: at most 32 clips per launch (one ones column per clip in one tile)");
  static_assert(sizeof(wae_tq_job) == sizeof(TqJob), "wae_tq_job and TqJob must have the same layout");
  static_assert(sizeof(wae_ts_seg) == sizeof(TqSeg), "wae_ts_seg and TqSeg must have the same layout");
  TqArgs a;
  a.jobs = (const TqJob*)jobs_dev;
  a.segs = (const TqSeg*)segs_dev;
  a.team_seg = team_seg_dev;
  a.nteams = nteams; a.team_size = team_size;
  a.B = B; a.T = T; a.spc = (T + 31) / 32;
  a.stamps = (long long*)stamps;
  a.pace = window > 0 ? (unsigned*)pace : nullptr;
  a.window = window;
  a.window_cond = window_cond;
  a.ntaps = ntaps;
  constexpr size_t lds_taps = (size_t)TqGeo<TQ_TAPS>::NS * TqDer<TQ_TAPS>::SLOT, lds_cond = (size_t)TqGeo<TQ_COND>::NS * TqDer<TQ_COND>::SLOT,
                   lds_os = (size_t)TqGeo<TQ_OUTSKIP>::NS * TqDer<TQ_OUTSKIP>::SLOT;
  constexpr size_t lds_ring = lds_taps > lds_os ? (lds_taps > lds_cond ? lds_taps : lds_cond) : (lds_os > lds_cond ? lds_os : lds_cond);
  constexpr size_t lds = lds_ring + 1024;   // + the dump area of the pacing wave's filler requests
  static_assert(lds <= 160 * 1024, "LDS");
  a.dump_off = (int)lds_ring;
  auto go = [&](auto kernel) -> int {
    static WaeLdsCache lds_cache;
    if (int rc = wae_ensure_lds((const void*)kernel, lds_cache, lds, "gemm_tn_static"); rc != WAE_OK) return rc;
    hipLaunchKernelGGL(kernel, dim3(nwg), dim3(TQ_NW * 64), lds, as_stream(stream), a);
    return WAE_OK;
  };
  int rc = dtype == WAE_F16 ? go(gemm_tn_static_kernel<f16>) : go(gemm_tn_static_kernel<__bf16>);
  if (rc != WAE_OK) return rc;
  return wae_check_launch("gemm_tn_static");
}
