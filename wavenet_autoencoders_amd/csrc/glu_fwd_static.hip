// glu_fwd_static: the fused ResidualConv1dGLU layer (reference: modules.py:115-163) for the 16-bit geometries the benchmarked
// configurations run (BASELINE C2 / hps/vqwae.json / C5), with the WHOLE chunk schedule resolved at compile time.
//
// Same arithmetic, same weight stream, same fragment layouts and the same MFMA order as glu_fwd_kernel (glu_fwd.hip; results are
// bit-identical, tests/test_gpu_parity.py) -- what differs is the instruction stream around the MFMAs:
//   * Rp, Ccp, Hp and the tap count are template constants: ring slots, LDS offsets, every `s_waitcnt vmcnt(n)` and the position of
//     every request inside a chunk's MFMA stream are immediates.  The generic kernel formed them at run time at the top of every
//     chunk (~100 scalar instructions and 18 branches per chunk).
//   * The activation operand is requested with `buffer_load_dwordx4` through a per-clip buffer descriptor: a row before t = 0
//     (the causal pad, conv.py / modules.py:131 `padding=(k-1)*d` + the `[:, :, :T]` crop) or past the clip's end is out of range and
//     the hardware returns zeros -- no clamp, no zero-fill selects, three per-lane row offsets for the whole kernel.
//   * The A-fragment reads run as ONE stream over all chunks of a pass (CONT): the first reads of chunk q + 1 are issued under the
//     last MFMAs of chunk q, so a chunk no longer starts with a cold LDS round trip after its barrier.  That needs the weights of
//     chunk q + 1 visible one barrier early; the schedule below carries the proof obligations as static_asserts.
//   * WAE_GLU_SAVE_Z / WAE_GLU_NO_OUT are template parameters: training and inference launches are different kernel symbols
//     (rocprof kernel stats and PMC passes separate them).
//
// Ring discipline (NSLOT slots, weights D chunks ahead, a workgroup barrier at the top of every BE-th chunk):
//   visibility  a wave reads chunk c's slot only after a barrier that every wave passed with its own DMA pieces of chunk c landed.
//               Between the barriers at Q and Q + BE a wave reads chunks Q .. Q + BE (the last one by read-ahead), so the wait in
//               front of the barrier at Q covers DMA(<= Q + BE): needs D >= BE + 1 (those pieces were issued before chunk Q).
//   reuse       DMA(c + D), issued inside chunk c, overwrites the slot of chunk c + D - NSLOT; between the barriers at Q and Q + BE
//               waves sit in chunks Q .. Q + BE: needs (BE - 1) + D - NSLOT < 0, i.e. D <= NSLOT - BE.
// The counted waits come out of a constexpr replay of the issue order (allowed_at), not out of hand-written formulas.
#include "glu_fwd.hpp"

typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifndef WAE_GS_PD
#define WAE_GS_PD 4        // A-fragment reads in flight per wave in GEMM 1
#endif
#ifndef WAE_GS_PRIO
#define WAE_GS_PRIO 0      // 0: none; 1: static s_setprio 1 for waves 4-7; 2: the two waves of a SIMD swap priority every chunk
#endif
#ifndef WAE_GS_CONT
#define WAE_GS_CONT 1
#endif
#ifndef WAE_GS_SLOTS6
#define WAE_GS_SLOTS6 5    // ring slots for 24-KiB chunks (C2)
#endif
#ifndef WAE_GS_BE
#define WAE_GS_BE 1
#endif
// timing-only ablations (tools/glu_ab.py on variant builds; results are wrong when any bit is set):
// 1 no in-stream weight DMA, 2 no in-stream activation requests, 4 no epilogue stores, 8 no gate transcendentals, 16 no MFMAs,
// 32 no A-fragment reads
#ifndef WAE_GS_ABL
#define WAE_GS_ABL 0
#endif
#ifndef WAE_GS_BVAR
#define WAE_GS_BVAR 0      // timing-only shapes of the activation requests: 1 one segment per row, 2 full lines (8 rows per request), 3 half
#endif

namespace {

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for<I + 1, N>(f);
  }
}

template <int OFF, typename F>
__device__ __forceinline__ void bload_async(F& dst, unsigned voff, i32x4 rsrc) {
  static_assert(sizeof(F) == 16 && OFF >= 0 && OFF < 4096, "one 16-byte fragment, 12-bit offset");
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "+v"(dst) : "v"(voff), "s"(rsrc), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void wait_vm() {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}
template <int OFF, typename frag>
__device__ __forceinline__ void lds_read_at(frag& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// raw buffer descriptor over [base, base + bytes): stride 0, offsets checked against num_records (an out-of-range load returns 0)
__device__ __forceinline__ i32x4 make_srd(const char* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

// ---- the schedule -----------------------------------------------------------------------------------------------------------
// Issue order of one wave's VMEM operations (stores excluded: they only make a counted wait stricter):
//   prologue   DMA(0) .. DMA(pre), B(0), B(1), DMA(pre + 1) .. DMA(D)          pre = chunks that must be visible at barrier 0
//   chunk c    [DMA(c + D) if c >= 1]  [B(c + 2)]                                (GEMM 1: c < NQ1T; pieces first, spread over the MFMAs)
//   chunk c    RES(c)  [DMA(c + D)]                                              (GEMM 2: c >= NQ1T; RES retired inside the chunk)
// allowed_at(Q): how many of the youngest operations may still be outstanding at the top of chunk Q.
template <int NQ1T, int NQT, int D, int BE, int CONT, int PPW, int NBL, int NRES>
struct Sched {
  static constexpr bool meet(int Q) { return Q >= NQ1T || Q % BE == 0; }
  static constexpr int vis(int Q) {   // last chunk whose weights must be visible to every wave after the barrier at Q
    int v = Q >= NQ1T ? Q : Q + BE - (CONT ? 0 : 1);
    return v < NQT - 1 ? v : NQT - 1;
  }
  static constexpr int pre() { int v = vis(0); return v < D ? v : D; }
  static constexpr bool has_dma(int c) { return c >= 1 && c + D < NQT; }
  static constexpr bool has_b(int c) { return c < NQ1T && c + 2 < NQ1T; }
  static constexpr int allowed_at(int Q) {
    int n = 0, last = 0;
    // prologue
    for (int c = 0; c <= pre() && c < NQT; ++c) { n += PPW; if (meet(Q) && c <= vis(Q)) last = n; }
    for (int c = 0; c < 2 && c < NQ1T; ++c) { n += NBL; if (c == Q) last = n; }
    for (int c = pre() + 1; c <= D && c < NQT; ++c) { n += PPW; if (meet(Q) && c <= vis(Q)) last = n; }
    for (int c = 0; c < Q; ++c) {
      if (c < NQ1T) {
        if (has_dma(c)) { n += PPW; if (meet(Q) && c + D <= vis(Q)) last = n; }
        if (has_b(c)) { n += NBL; if (c + 2 == Q) last = n; }
      } else {
        n += NRES; last = n;   // retired inside chunk c
        if (has_dma(c)) { n += PPW; if (c + D <= vis(Q)) last = n; }
      }
    }
    return n - last;
  }
};

}  // namespace

// (outside the anonymous namespace, and two differently NAMED entry points below: rocprofv3's kernel names then start with
// `glu_fwd_static_kernel<` for inference launches and `glu_fwd_static_z_kernel<` for training launches (z saved) -- its demangler
// garbles the template arguments of these symbols, so the name itself has to tell the launch kinds apart in kernel stats and PMC passes)
template <typename E, int NP, int NPHP, bool ONEP, int CPR, int NCC, int KT, bool SAVE_Z, bool NO_OUT, int NSLOT, int D, int BE, bool CONT, int PD>
__device__ __forceinline__ void glu_fwd_static_body(const GluArgs& p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  static_assert(sizeof(E) == 2 && T_::CK == 64 && T_::KBU == 2, "16-bit storage only");
  constexpr int NW = 8;
  // NPHP: gate-channel tiles per pass of the PACKED weight stream (packing.py: glu_pass_tiles).  ONEP: the kernel nevertheless runs
  // GEMM 1 in ONE pass over all NP tile pairs (2 NP accumulator tiles, 192 registers at Hp = 192) on chunks of KB = 2 k-blocks
  // (K = 32): the activation operand is then requested once instead of once per pass -- half the requests, which is what takes the
  // CU's texture-address path out of saturation (a request of 32 rows x 32 bytes costs 64 quad accesses, and
  // with four of them per wave and chunk on top of the weight pieces the path was the co-limiter of the chunk loop).  A chunk is
  // then the (k-block pair, all passes) cut of two packed chunks; its LDS image is [packed pass][k-block][tile], filled by pieces
  // with per-wave source offsets.
  constexpr int NPH = ONEP ? NP : NPHP, KB = ONEP ? 2 : 4, PP = NP / NPHP, NMP = 2 * NPHP;
  constexpr int NPASS = NP / NPH, NM = 2 * NPH, CHB = NM * KB * 1024, PCHB = NMP * 4 * 1024, ES = 2;
  constexpr int NQC = KT * CPR, PQ1 = NQC + NCC, NQ1 = PQ1 * (4 / KB), NQ1T = NPASS * NQ1;
  constexpr int NKB = NP * 2, MT2 = PCHB / (NKB * 1024);
  static_assert(NP % NPHP == 0 && MT2 >= 1 && MT2 * NKB * 1024 == PCHB && PCHB == CHB, "GEMM-2 chunk must equal GEMM-1 chunk");
  static_assert(NQ1T == PP * PQ1, "GEMM-2 chunks keep their packed index");
  constexpr int NQ2 = NO_OUT ? 0 : (CPR * 2) / MT2, NQT = NQ1T + NQ2;
  constexpr int ROWX = CPR * 128, ROWC = NCC * 128, RP = CPR * 64, HP = NP * 32;
  constexpr int PITCH = 128, STG = 32 * PITCH;
  constexpr int PPW = CHB / NW / 1024, NBL = KB, NSTEP = KB * NM, NOPS = PPW + NBL, SP = NSTEP / NOPS;
  static_assert(!ONEP || NMP % PPW == 0, "a wave's pieces stay inside one (pass, k-block) run of the packed chunk");
  constexpr int NRES = 2 * MT2;
  // global store instructions of stage_store_tiles<E, NT, PITCH = 128> for a full 32-row tile group: passes of two tiles (4) and one (2)
  constexpr int ST_PASS = (NPH / 2) * 4 + (NPH % 2) * 2;                                  // one stage_store_tiles<E, NPH>
  constexpr int GHS = (NPH >= 6 && NPH % 2 == 0) ? NPH / 2 : NPH;                           // the u store goes in NPH / GHS groups
  constexpr int ST_U = (NPH / GHS) * ((GHS / 2) * 4 + (GHS % 2) * 2);
  constexpr int NST_PASS = (SAVE_Z ? 2 * ST_PASS : 0) + ST_U;
  static_assert(NOPS * SP <= NSTEP, "not enough MFMA steps to carry a chunk's VMEM issue");
  static_assert(D >= BE + (CONT ? 1 : 0) && D <= NSLOT - BE, "ring discipline (see the header comment)");
  static_assert(2 * CHB <= 65536, "two slots per ds_read base register");
  static_assert(!CONT || NSTEP >= PD, "read-ahead reaches at most one chunk ahead");
  using SC = Sched<NQ1T, NQT, D, BE, CONT ? 1 : 0, PPW, NBL, NRES>;

  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef WAE_GLU_STAMPS
  unsigned long long st_[16] = {};
  if (p.stamps) st_[14] = __builtin_amdgcn_s_memrealtime();
#define GS_STAMP(i)                                  \
  do {                                               \
    if (p.stamps) {                                  \
      __builtin_amdgcn_sched_barrier(0);             \
      st_[i] = __builtin_amdgcn_s_memtime();         \
      __builtin_amdgcn_sched_barrier(0);             \
    }                                                \
  } while (0)
  unsigned long long acc_wait = 0, acc_bar = 0, acc_g1 = 0;
#define GS_TICK(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define GS_ACC(a, d) a += (d)
#else
#define GS_STAMP(i) do { } while (0)
#define GS_TICK(v) do { } while (0)
#define GS_ACC(a, d) do { } while (0)
#endif
  GS_STAMP(0);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  constexpr int TW = NW * 32;
  const int tiles_per_b = (p.T + TW - 1) / TW;
  const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int b = tile_id / tiles_per_b;
  const int t0w = (tile_id % tiles_per_b) * TW + wave * 32;
  const int t = t0w + n;

  const unsigned clip_x = (unsigned)p.T * ROWX, clip_c = (unsigned)p.T * ROWC;
  const i32x4 srd_xc = make_srd(p.x_conv + (int64_t)b * clip_x, clip_x);   // convolution operand (modules.py:127-131)
  const i32x4 srd_xr = make_srd(p.x_in + (int64_t)b * clip_x, clip_x);     // residual path (modules.py:126,161)
  const i32x4 srd_c = make_srd(p.c_up + (int64_t)b * clip_c, clip_c);
  // per-lane row offsets: tap j reads row t - (KT-1-j) d; a negative row wraps to an offset >= 2^31 > num_records -> zeros
  unsigned voff_tap[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) voff_tap[j] = (unsigned)((t - (KT - 1 - j) * p.dilation) * ROWX + h * 16);
  const unsigned voff_c = (unsigned)(t * ROWC + h * 16);

  char* ring_end = smem + NSLOT * CHB;
  char* stg = ring_end + wave * STG;
  float* bias_lds = (float*)(ring_end + NW * STG);
  float* zb_lds = bias_lds + RP;
  // ---- request helpers (all indices compile-time) -------------------------------------------------------------------------
  constexpr int per_wave = CHB / NW;
  const unsigned lane_off = (unsigned)(wave * per_wave + lane * 16);
  char* lds_wave = smem + wave * per_wave;
  // ONEP: piece L = wave * PPW + k of a GEMM-1 chunk is block (packed pass L / (KB NMP), k-block (L % (KB NMP)) / NMP, tile L % NMP)
  [[maybe_unused]] const int L0 = wave * PPW;
  [[maybe_unused]] const unsigned onep_src = (unsigned)(((L0 / (KB * NMP)) * PQ1) * PCHB + (((L0 % (KB * NMP)) / NMP) * NMP + L0 % NMP) * 1024 + lane * 16);
  auto dma_piece_of = [&](auto cc, auto kc) {   // piece k of weight chunk c -> its ring slot
    constexpr int c = decltype(cc)::value, k = decltype(kc)::value;
    if constexpr (ONEP && c < NQ1T) {
      constexpr int q = c / 2, hk = c % 2;
      const char* sbase = p.w + ((int64_t)q * PCHB + hk * (KB * NMP * 1024) + k * 1024);
      dma_piece(sbase + onep_src, lds_wave + (c % NSLOT) * CHB + k * 1024);
    } else {
      const char* sbase = p.w + ((int64_t)c * CHB + k * 1024);
      dma_piece(sbase + lane_off, lds_wave + (c % NSLOT) * CHB + k * 1024);
    }
  };
  auto dma_whole = [&](auto cc) {
    static_for<0, PPW>([&](auto kc) { dma_piece_of(cc, kc); });
  };
  frag S[3][KB] = {};   // activation fragments of chunks c, c + 1, c + 2 (group c % 3); defined: the asm loads are read-write
  auto b_piece = [&](auto cc, auto blkc) {   // fragment blk of activation chunk c
    constexpr int c = decltype(cc)::value;
    constexpr int q = (c % NQ1) / (4 / KB), blk = ((c % NQ1) % (4 / KB)) * KB + decltype(blkc)::value, bi = decltype(blkc)::value;
#if WAE_GS_BVAR == 3
    if constexpr (blk >= 2) return;
#endif
#if WAE_GS_BVAR == 1
    constexpr int boff = 0;
#else
    constexpr int boff = blk * 32;
#endif
    if constexpr (q < NQC) {
      constexpr int cblk = q / KT, tap = q % KT;   // taps of one column block back to back (packing.py: glu_w1_map)
#if WAE_GS_BVAR == 2
      // timing only: the same bytes as full 128-byte lines, 8 rows per request
      const unsigned vo = (unsigned)((t0w + (lane >> 3) + 8 * blk - (KT - 1 - tap) * p.dilation) * ROWX + (lane & 7) * 16);
      bload_async<cblk * 128>(S[c % 3][bi], vo, srd_xc);
#elif WAE_GS_BVAR == 4
      // timing only: the request shape of a 16x16x32 operand fragment -- 16 rows x 64 bytes
      const unsigned vo = (unsigned)((t0w + 16 * (blk & 1) + (lane & 15) - (KT - 1 - tap) * p.dilation) * ROWX + (lane >> 4) * 16);
      bload_async<cblk * 128 + (blk >> 1) * 64>(S[c % 3][bi], vo, srd_xc);
#else
      bload_async<cblk * 128 + boff>(S[c % 3][bi], voff_tap[tap], srd_xc);
#endif
    } else {
      bload_async<(q - NQC) * 128 + boff>(S[c % 3][bi], voff_c, srd_c);
    }
  };
  auto b_whole = [&](auto cc) {
    static_for<0, KB>([&](auto blkc) { b_piece(cc, blkc); });
  };

  // ---- prologue -------------------------------------------------------------------------------------------------------------
  // out bias and this clip's zb -> LDS (read back with ds_read: keeps accumulator inits off vmcnt).  Their loads go out FIRST, the
  // requests chunk 0 needs right behind them, and only then the table values are written to LDS: one cold round trip per launch
  // carries both (hipcc's own wait for the table data cannot see the asm requests behind it, so it can only be stricter than
  // needed).  The rest of the weight burst follows the writes: every CU is in its prologue at once, and whatever is queued in front
  // of chunk 0's operands delays the first MFMA.
  static_assert(RP <= NW * 256 && 2 * HP <= NW * 256, "one 16-byte table piece per thread");
  const bool has_tb = NQ2 > 0 && (int)threadIdx.x * 4 < RP, has_tz = (int)threadIdx.x * 4 < 2 * HP;
  f32x4 tb = {}, tz = {};
  if (has_tb) tb = *(const f32x4*)(p.bias_out + threadIdx.x * 4);
  if (has_tz) tz = *(const f32x4*)(p.zb + (int64_t)b * p.zb_stride + threadIdx.x * 4);
  static_for<0, SC::pre() + 1>([&](auto cc) { dma_whole(cc); });
  b_whole(IntC<0>{});
  if constexpr (NQ1T > 1) b_whole(IntC<1>{});
  if (has_tb) *(f32x4*)(bias_lds + threadIdx.x * 4) = tb;
  if (has_tz) *(f32x4*)(zb_lds + threadIdx.x * 4) = tz;
  static_for<SC::pre() + 1, (D < NQT - 1 ? D : NQT - 1) + 1>([&](auto cc) { dma_whole(cc); });
  GS_STAMP(1);

#if WAE_GS_PRIO == 1
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif

  // ds_read offsets are 16 bits: one base register per pair of ring slots
  unsigned a_base[(NSLOT + 1) / 2];
#pragma unroll
  for (int i = 0; i < (NSLOT + 1) / 2; ++i)
    a_base[i] = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + lane * 16 + i * 2 * CHB;
  auto a_read = [&](auto cc, auto ic, frag& dst) {   // A fragment block I = (k-block jb, accumulator tile a) of weight chunk c
    constexpr int c = decltype(cc)::value, I = decltype(ic)::value, slot = c % NSLOT;
    constexpr int jb = I / NM, a = I % NM;
    // accumulator tiles: [tanh tiles 0 .. NPH) | sigmoid tiles 0 .. NPH); packed pass of a tile = tile / NPHP
    constexpr int tt = a < NPH ? a : a - NPH;
    constexpr int L = ONEP ? (tt / NPHP) * (KB * NMP) + jb * NMP + (a < NPH ? tt % NPHP : NPHP + tt % NPHP) : I;
    lds_read_at<(slot & 1) * CHB + L * 1024>(dst, a_base[slot / 2]);
  };
  const bool full_rows = __builtin_amdgcn_readfirstlane(p.T - t0w) >= 32;   // wave-uniform
  auto chunk_top = [&](auto qc) {
    constexpr int Q = decltype(qc)::value;
    GS_TICK(tk0);
    // first chunk of a later pass: the previous pass's epilogue stores are the youngest operations of a wave that owns a full
    // 32-row group (their count is static); a wave at the clip's tail issued fewer, so it takes the store-blind (stricter) count
    if constexpr (Q > 0 && Q < NQ1T && Q % NQ1 == 0) {
      if (full_rows) wait_vm<SC::allowed_at(Q) + NST_PASS>();
      else wait_vm<SC::allowed_at(Q)>();
    } else {
      wait_vm<SC::allowed_at(Q)>();
    }
    GS_TICK(tk1);
    // a bare s_barrier: __syncthreads() is fence + barrier and the fence drains vmcnt, i.e. the whole prefetch queue
    if constexpr (SC::meet(Q)) __builtin_amdgcn_s_barrier();
    GS_TICK(tk2);
    GS_ACC(acc_wait, tk1 - tk0);
    GS_ACC(acc_bar, tk2 - tk1);
#if WAE_GS_PRIO == 2
    if constexpr (Q < NQ1T) {
      if (((wave >> 2) ^ Q) & 1) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
#endif
  };

  const unsigned voff_res = (unsigned)(t * ROWX + h * 16);
  constexpr int NT2 = NQ2 * MT2;                 // output tiles (Rp / 32)
  constexpr bool GROUPED = NQ2 > 0 && NT2 <= 8 && NQ2 <= NSLOT && NT2 % 4 == 0;
  frag res[8];   // GROUPED: residual x[t] as operand-shaped 16-byte fragments, four output tiles at a time
  frag uf[NKB];
  frag a_init_ = {};
  static_for<0, NPASS>([&](auto psc) {
    constexpr int ps = decltype(psc)::value;
    // ---- accumulators start from zb = conv bias + hoisted global conditioning ---------------------------------------------
    f32x16 acc[NM];
    // zb / bias tables: the writers' ds_writes are retired by lgkmcnt(0), the barrier of the pass's first chunk top publishes them
    // (__syncthreads() here would be fence + barrier, and the fence drains vmcnt: the whole prologue prefetch, once per launch)
    if constexpr (ps == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    static_assert(ps != 0 || SC::meet(0), "chunk 0 meets");
    chunk_top(IntC<ps * NQ1>{});
#pragma unroll
    for (int m = 0; m < NM; ++m) init_rows(acc[m], zb_lds + (m < NPH ? 32 * (ps * NPH + m) : HP + 32 * (ps * NPH + m - NPH)), h);

    // ---- GEMM 1, pass ps: one instruction stream over NQ1 chunks ----------------------------------------------------------
    GS_TICK(tg0);
    frag a[PD];
    if constexpr (WAE_GS_ABL & 32) {
#pragma unroll
      for (int i = 0; i < PD; ++i) a[i] = a_init_;
    }
    constexpr int NG = NQ1 * NSTEP;   // MFMA steps of the pass
    static_for<0, NQ1>([&](auto qc) {
      constexpr int q = decltype(qc)::value, Q = ps * NQ1 + q;
      if constexpr (q > 0) chunk_top(IntC<Q>{});
      // B(Q) has landed (counted wait): every use comes after this point
      if constexpr (KB == 4) asm volatile("" : "+v"(S[Q % 3][0]), "+v"(S[Q % 3][1]), "+v"(S[Q % 3][2]), "+v"(S[Q % 3][3]));
      else asm volatile("" : "+v"(S[Q % 3][0]), "+v"(S[Q % 3][1]));
      if constexpr (q == 0 || !CONT) {
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(WAE_GS_ABL & 32)) static_for<0, PD>([&](auto ic) { a_read(IntC<Q>{}, ic, a[decltype(ic)::value]); });
      }
      static_for<0, NSTEP>([&](auto ic) {
        constexpr int I = decltype(ic)::value, G = q * NSTEP + I;
        constexpr int AI = (CONT ? G : I) % PD;
        constexpr int remaining = CONT ? NG - 1 - G : NSTEP - 1 - I;   // reads issued after this block's
        if constexpr (!(WAE_GS_ABL & 32)) lds_wait<(remaining < PD - 1 ? remaining : PD - 1)>(a[AI]);
        if constexpr (!(WAE_GS_ABL & 16)) mma32(acc[I % NM], a[AI], S[Q % 3][I / NM]);
        if constexpr (remaining >= PD && !(WAE_GS_ABL & 32)) {
          constexpr int G2 = G + PD;
          a_read(IntC<ps * NQ1 + G2 / NSTEP>{}, IntC<G2 % NSTEP>{}, a[AI]);
        }
        if constexpr (I % SP == 0 && I / SP < NOPS) {   // this chunk's requests, spread over its MFMAs: pieces first
          constexpr int k = I / SP;
          if constexpr (k < PPW) {
            if constexpr (SC::has_dma(Q) && !(WAE_GS_ABL & 1)) dma_piece_of(IntC<Q + D>{}, IntC<k>{});
          } else {
            if constexpr (SC::has_b(Q) && !(WAE_GS_ABL & 2)) b_piece(IntC<Q + 2>{}, IntC<k - PPW>{});
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    GS_TICK(tg1);
    GS_ACC(acc_g1, tg1 - tg0);
    GS_STAMP(2 + 2 * ps);

    if constexpr (GROUPED && ps == NPASS - 1) {
      // residual rows of the first four output tiles (L2 hits: tap k-1 of GEMM 1 read the same bytes), requested AHEAD of this
      // epilogue's stores: the wait in front of GEMM 2 then leaves every store in flight
      static_for<0, 8>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        const frag zf = {};
        res[f] = zf;
        bload_async<f * 32>(res[f], voff_res, srd_xr);
      });
    }
    const int rows_valid = (WAE_GS_ABL & 4) ? 0 : min(max(p.T - t0w, 0), 32);
    const int64_t row0 = (int64_t)b * p.T + t0w;
    // ---- optional z save (training): rows of 2Hp elements, a-half then b-half ----------------------------------------------
    if constexpr (SAVE_Z) {
      if (rows_valid > 0) {
        char* zr = p.z_save + (row0 * (2 * HP) + ps * NPH * 32) * ES;
        stage_store_tiles<E, NPH, PITCH>(stg, &acc[0], zr, (int64_t)2 * HP * ES, rows_valid, lane);
        stage_store_tiles<E, NPH, PITCH>(stg, &acc[NPH], zr + (int64_t)HP * ES, (int64_t)2 * HP * ES, rows_valid, lane);
      }
    }
    // ---- gate: u = tanh(a) * sigmoid(b); stored once for the head's skip GEMM, and converted in place to the operand
    //      fragments of GEMM 2 ------------------------------------------------------------------------------------------------
    // (six tile pairs -- the one-pass form -- go in two halves, each followed by the store of its three u tiles: the stored tiles'
    //  registers are free before the next half's temporaries arrive)
    constexpr int GH = (NPH >= 6 && NPH % 2 == 0) ? NPH / 2 : NPH;
#pragma unroll
    for (int gh = 0; gh < NPH / GH; ++gh) {
#pragma unroll
      for (int pr = gh * GH; pr < (gh + 1) * GH; ++pr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float av = acc[pr][r], g = acc[NPH + pr][r];
          // tanh(a)*sigmoid(g) = (1-ea) / ((1+ea)(1+eg)), ea = e^-2a, eg = e^-g.  a is clamped from below so that ea stays
          // finite (tanh(-15) == -1 in fp32); eg = inf gives rcp(inf) = 0, the correct limit.
          const f32x2 sc = {-2.885390081777927f, -1.4426950408889634f};
          float amax;
          asm("v_max_f32 %0, %1, %2" : "=v"(amax) : "v"(av), "v"(-15.0f));
          f32x2 ag = {amax, g};
          ag = ag * sc;
          const float ea = (WAE_GS_ABL & 8) ? ag.x : __builtin_amdgcn_exp2f(ag.x);
          const float eg = (WAE_GS_ABL & 8) ? ag.y : __builtin_amdgcn_exp2f(ag.y);
          const f32x2 one = {1.0f, 1.0f};
          const f32x2 e2 = {ea, eg};
          const f32x2 d = e2 + one;
          acc[pr][r] = (1.0f - ea) * ((WAE_GS_ABL & 8) ? d.x * d.y : fast_rcp(d.x * d.y));
        }
        if constexpr (NQ2 > 0) {   // (the last layer's launch has no second GEMM: x' is dead, wavenet.py:205-207)
          frag tmp[2];
          acc_to_frags(acc[pr], tmp);
          uf[(ps * NPH + pr) * 2] = tmp[0];
          uf[(ps * NPH + pr) * 2 + 1] = tmp[1];
        }
        __builtin_amdgcn_sched_barrier(0);  // one tile at a time: keeps the gate's temporaries from piling up
      }
      if (rows_valid > 0) {
        char* ur = p.u_out + (row0 * p.u_stride + (ps * NPH + gh * GH) * 32) * ES;
        stage_store_tiles<E, GH, PITCH>(stg, &acc[gh * GH], ur, p.u_stride * ES, rows_valid, lane);
      }
    }
    GS_STAMP(3 + 2 * ps);
  });

  // ---- GEMM 2 + residual epilogue ---------------------------------------------------------------------------------------------
  if constexpr (GROUPED) {
    // All of GEMM 2 as one phase: every output tile stays in registers (NT2 <= 8), the MFMAs of all NQ2 chunks run back to back and
    // ONE residual epilogue follows -- the per-chunk form below alternates 24 MFMAs per wave with a two-tile epilogue NQ2 times with
    // every wave of the workgroup in the same phase (stamps: 4800 clocks per chunk for 1536 of matrix pipe).
    // Visibility: chunks NQ1T .. FU - 1 (FU = first chunk GEMM 1's stream has not requested) are retired by the drain + barrier
    // below; chunks >= FU are requested right after it -- their slots held chunks < NQ1T, which every wave has left -- and meet
    // at one more barrier in front of their MFMAs.
    constexpr int FU = (NQ1T - 1 + D + 1) < NQT ? (NQ1T - 1 + D + 1) : NQT;   // chunks < FU were requested by GEMM 1 (chunk c requests c + D)
    static_assert(NQT - FU <= NSLOT - (FU - NQ1T), "late GEMM-2 chunks must fit the slots GEMM 1 has left");
    // chunks < FU and their pieces are older than the last pass's epilogue stores: those stay in flight (static count for a full
    // 32-row group; a tail wave drains)
    if (full_rows) wait_vm<NST_PASS>();
    else wait_vm<0>();
    GS_STAMP(9);
    __builtin_amdgcn_s_barrier();
    GS_STAMP(10);
    static_for<FU, NQT>([&](auto cc) { dma_whole(cc); });
    constexpr int NLATE = (NQT - FU) * PPW;
    f32x16 y[NT2];
#pragma unroll
    for (int mt = 0; mt < NT2; ++mt) init_rows(y[mt], bias_lds + 32 * mt, h);
    NoFiller nf;
    static_for<0, NQ2>([&](auto q2c) {
      constexpr int q2 = decltype(q2c)::value, Q = NQ1T + q2;
      if constexpr (Q == FU) {   // the late chunks: this wave's pieces landed, then meet
        wait_vm<0>();              // (its pieces are the youngest requests of the wave)
        __builtin_amdgcn_s_barrier();
      }
      const char* buf = smem + (Q % NSLOT) * CHB + lane * 16;
      gemm_chunk_fill<MT2 * NKB, MT2, NKB, true, 4>(buf, uf, *(f32x16(*)[MT2]) & y[q2 * MT2], nf);
    });
    (void)NLATE;
    const int rows_valid = (WAE_GS_ABL & 4) ? 0 : min(max(p.T - t0w, 0), 32);
    const f32x2 rs = {0.70710678118654752440f, 0.70710678118654752440f};
    static_for<0, NT2 / 4>([&](auto hc) {
      constexpr int hf = decltype(hc)::value;
      // the residual fragments of this half are older than the previous half's 8 store instructions: those stay in flight
      if constexpr (hf == 0) {
        GS_STAMP(11);
        if constexpr (FU == NQT) {   // no late chunk has drained the queue: the last pass's stores may still be in flight
          if (full_rows) wait_vm<NST_PASS>();
          else wait_vm<0>();
        }
      } else {
        if (full_rows) wait_vm<8>();
        else wait_vm<0>();
      }
#pragma unroll
      for (int f = 0; f < 8; ++f) asm volatile("" : "+v"(res[f]));
      residual_to_acc_layout(res);
      // x' = (y + x) * sqrt(.5) in the accumulator layout
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 r4 = residual_piece<E>(res, mt, g);
          f32x16& yt = y[4 * hf + mt];
          f32x2 lo = {yt[4 * g + 0], yt[4 * g + 1]}, hi = {yt[4 * g + 2], yt[4 * g + 3]};
          const f32x2 rlo = {r4.x, r4.y}, rhi = {r4.z, r4.w};
          lo = (lo + rlo) * rs;
          hi = (hi + rhi) * rs;
          yt[4 * g + 0] = lo.x; yt[4 * g + 1] = lo.y; yt[4 * g + 2] = hi.x; yt[4 * g + 3] = hi.y;
        }
      }
      if constexpr (hf + 1 < NT2 / 4)   // the next four tiles' residual rows travel under this half's stores
        static_for<0, 8>([&](auto fc) { bload_async<(hf + 1) * 256 + decltype(fc)::value * 32>(res[decltype(fc)::value], voff_res, srd_xr); });
      if (rows_valid > 0) {
        char* orow = p.x_out + ((int64_t)b * p.T + t0w) * ROWX + (int64_t)hf * 128 * ES;
        stage_store_tiles<E, 4, PITCH>(stg, &y[4 * hf], orow, ROWX, rows_valid, lane);
      }
    });
  } else {
  // Each chunk = MT2 M-tiles (MT2*32 output channels = one staging pass per time row) against all of u.
  static_for<0, NQ2>([&](auto q2c) {
    constexpr int q2 = decltype(q2c)::value, Q = NQ1T + q2;
    constexpr bool dma_now = Q + D < NQT;
    // q2 == 0: drain (the pass epilogues' stores are uncounted); later: DMA(Q) has landed once only the younger chunks' residual
    // requests and pieces are outstanding
    if constexpr (q2 == 0) wait_vm<0>();
    else wait_vm<SC::allowed_at(Q)>();
    __builtin_amdgcn_s_barrier();
    const char* buf = smem + (Q % NSLOT) * CHB + lane * 16;
    constexpr int gm0 = q2 * MT2;
    // residual x[t] for this chunk's channels, as operand-shaped 16-byte fragments (L2 hits: tap k-1 of GEMM 1 read the same
    // bytes); requested FIRST (loads retire in order: the wait for them below then leaves this chunk's DMA pieces in flight)
    frag res[NRES] = {};
    static_for<0, NRES>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      bload_async<gm0 * 32 * ES + f * 32>(res[f], voff_res, srd_xr);
    });
    if constexpr (dma_now) dma_whole(IntC<Q + D>{});
    f32x16 y[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) init_rows(y[mt], bias_lds + 32 * (gm0 + mt), h);
    NoFiller nf;
    gemm_chunk_fill<MT2 * NKB, MT2, NKB, true, 8>(buf, uf, y, nf);
    // x' = (y + x) * sqrt(.5) in the accumulator layout
    const f32x2 rs = {0.70710678118654752440f, 0.70710678118654752440f};
    wait_vm<(dma_now ? PPW : 0)>();
#pragma unroll
    for (int f = 0; f < NRES; ++f) asm volatile("" : "+v"(res[f]));
    residual_to_acc_layout(res);
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 r4 = residual_piece<E>(res, mt, g);
        f32x2 lo = {y[mt][4 * g + 0], y[mt][4 * g + 1]}, hi = {y[mt][4 * g + 2], y[mt][4 * g + 3]};
        const f32x2 rlo = {r4.x, r4.y}, rhi = {r4.z, r4.w};
        lo = (lo + rlo) * rs;
        hi = (hi + rhi) * rs;
        y[mt][4 * g + 0] = lo.x; y[mt][4 * g + 1] = lo.y; y[mt][4 * g + 2] = hi.x; y[mt][4 * g + 3] = hi.y;
      }
    }
    const int rows_valid = (WAE_GS_ABL & 4) ? 0 : min(max(p.T - t0w, 0), 32);
    if (rows_valid > 0) {
      char* orow = p.x_out + ((int64_t)b * p.T + t0w) * ROWX + (int64_t)gm0 * 32 * ES;
      stage_store_tiles<E, MT2, PITCH>(stg, y, orow, ROWX, rows_valid, lane);
    }
  });
  }
  GS_STAMP(8);
#ifdef WAE_GLU_STAMPS
  if (p.stamps && lane == 0) {   // every wave: its own sums over the GEMM-1 chunk tops
    unsigned long long* o = p.stamps + (size_t)blockIdx.x * 64 + 16 + wave * 4;
    o[0] = acc_wait; o[1] = acc_bar; o[2] = acc_g1; o[3] = __builtin_amdgcn_s_memtime() - st_[0];
  }
  if (p.stamps && threadIdx.x == 0) {
    st_[15] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) p.stamps[(size_t)blockIdx.x * 64 + i] = st_[i];
  }
#endif
}

template <typename E, int NP, int NPHP, bool ONEP, int CPR, int NCC, int KT, bool NO_OUT, int NSLOT, int D, int BE, bool CONT, int PD>
__global__ void __launch_bounds__(512, 1) glu_fwd_static_kernel(GluArgs p) {     // inference launch: no z
  glu_fwd_static_body<E, NP, NPHP, ONEP, CPR, NCC, KT, false, NO_OUT, NSLOT, D, BE, CONT, PD>(p);
}
template <typename E, int NP, int NPHP, bool ONEP, int CPR, int NCC, int KT, bool NO_OUT, int NSLOT, int D, int BE, bool CONT, int PD>
__global__ void __launch_bounds__(512, 1) glu_fwd_static_z_kernel(GluArgs p) {   // training launch: also stores the pre-activations z
  glu_fwd_static_body<E, NP, NPHP, ONEP, CPR, NCC, KT, true, NO_OUT, NSLOT, D, BE, CONT, PD>(p);
}

namespace {

#ifndef WAE_GS_ONEPASS
#define WAE_GS_ONEPASS 1
#endif
template <typename E, int NP, int NPHP, int CPR, int NCC, int KT, bool SAVE_Z, bool NO_OUT>
int launch_static(const GluArgs& a, hipStream_t st) {
  constexpr int CHB = 2 * NPHP * 4 * 1024;
  // Hp = 192 (two packed passes of 24-KiB chunks): ONE pass over 24-KiB chunks of K = 32
  constexpr bool ONEP = WAE_GS_ONEPASS && NP == 6 && NPHP == 3;
  // 24-KiB chunks: 5 slots, weights four chunks ahead (every GEMM-2 chunk is then requested inside GEMM 1); 32-KiB chunks: 3 slots, two ahead
  constexpr int NSLOT = CHB <= 24 * 1024 ? WAE_GS_SLOTS6 : 3;
  constexpr int BE = WAE_GS_BE;
  constexpr int D = NSLOT - BE;
  constexpr bool CONT = WAE_GS_CONT && D >= BE + 1;
  constexpr int RP = CPR * 64, HP = NP * 32;
  void (*kern)(GluArgs);
  if constexpr (SAVE_Z) kern = glu_fwd_static_z_kernel<E, NP, NPHP, ONEP, CPR, NCC, KT, NO_OUT, NSLOT, D, BE, CONT, WAE_GS_PD>;
  else kern = glu_fwd_static_kernel<E, NP, NPHP, ONEP, CPR, NCC, KT, NO_OUT, NSLOT, D, BE, CONT, WAE_GS_PD>;
  const size_t lds = (size_t)NSLOT * CHB + 8 * 4096 + (size_t)(RP + 2 * HP) * 4;
  static_assert((size_t)NSLOT * CHB + 8 * 4096 + (size_t)(RP + 2 * HP) * 4 <= 160 * 1024, "LDS budget");
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)kern, lds_cache, lds, "glu_fwd_static"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(a.B * tiles), dim3(512), lds, st, a);
  return wae_check_launch("glu_fwd_static");
}

template <typename E, int NP, int NPH, int CPR, int NCC, int KT>
int launch_static_flags(const GluArgs& a, hipStream_t st) {
  const bool sz = a.flags & WAE_GLU_SAVE_Z, no = a.flags & WAE_GLU_NO_OUT;
  if (sz) return no ? launch_static<E, NP, NPH, CPR, NCC, KT, true, true>(a, st) : launch_static<E, NP, NPH, CPR, NCC, KT, true, false>(a, st);
  return no ? launch_static<E, NP, NPH, CPR, NCC, KT, false, true>(a, st) : launch_static<E, NP, NPH, CPR, NCC, KT, false, false>(a, st);
}

template <typename E>
int dispatch_static(const GluArgs& a, hipStream_t st, bool* handled) {
  *handled = true;
  if (a.ktaps == 3 && a.Ccp == 64) {
    if (a.Rp == 256 && a.Hp == 192) return launch_static_flags<E, 6, 3, 4, 1, 3>(a, st);   // BASELINE C2 (inae dims, G = 368)
    if (a.Rp == 256 && a.Hp == 128) return launch_static_flags<E, 4, 4, 4, 1, 3>(a, st);   // hps/vqwae.json (C1 / C3 / C4)
#ifndef WAE_GS_NO_WIDE
    if (a.Rp == 512 && a.Hp == 256) return launch_static_flags<E, 8, 4, 8, 1, 3>(a, st);   // C5 (48 layers x 512)
#endif
  }
  *handled = false;
  return WAE_OK;
}

}  // namespace

int wae_glu_static_launch(const GluArgs& a, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (!wae_is16(dtype) || !a.c_up) return WAE_OK;
  // the buffer descriptors address a clip with 32-bit offsets; a negative row must wrap beyond num_records
  if ((int64_t)a.T * a.Rp * 2 >= (int64_t)1 << 31) return WAE_OK;
  if (dtype == WAE_BF16) return dispatch_static<__bf16>(a, st, handled);
  return dispatch_static<f16>(a, st, handled);
}
