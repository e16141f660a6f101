// wae_glu_bwd_fused: the backward data path across one layer boundary in ONE launch (autograd of modules.py:115-163):
//
//   dx_l-hat[t]  = sqrt(.5) * ( dx_{l+1}-hat[t] + sum_tap W1_l,tap^T dz_l[t + (k-1-tap) d_l] )        (K_X of layer l)
//   du_{l-1}[t]  = W_out_{l-1}^T dx_l-hat[t] + W_skip_{l-1}^T dskip[t]                                 (K_U of layer l-1)
//   dz_{l-1}[t]  = gate'(z_{l-1}[t]) * du_{l-1}[t]
//
// ("-hat" = the stored gradient already carries the layer's sqrt(.5).)  The unfused path runs these as two wae_gemm_tm
// launches (RESIDUAL, then GATE_BWD) and re-reads dx_l-hat from HBM in between.  Here the 32x32 accumulator tiles of the
// first GEMM (column = time on the lane, rows = residual channels in the registers) are scaled, stored once (the weight
// gradient of conv1x1_out and the next boundary need them) and converted in place into the B operand of the second GEMM
// (cdna_hip_programming.md section 3, "An accumulator tile as the next MFMA's operand") -- the same chaining as
// csrc/glu_fwd.hip, run backwards.  Decomposition as in csrc/gemm_tm.hip: 128 time steps per workgroup, one wave per
// 32 time columns owning all rows, weights in A-fragment order through a double-buffered LDS ring.
#include "glu_bwd.hpp"

template <typename E, int NTX, int NTU>
__global__ void __launch_bounds__(256, 1) glu_bwd_fused_kernel(GbArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  constexpr int ES = sizeof(E);
  constexpr int KBU = T_::KBU;
  constexpr int CHX = NTX * 4 * 1024;            // GEMM-A chunk (= ring slot)
  constexpr int NKB = NTX * KBU;                 // 16-byte k-blocks of GEMM B1 (K = Rp)
  constexpr int MTB = (CHX / (NKB * 1024) >= 2 && NTU % 2 == 0) ? 2 : 1;   // M-tiles per B1 chunk
  constexpr int CHB1 = MTB * NKB * 1024;
  constexpr int CHB2 = NTU * 4 * 1024;
  static_assert(CHB1 <= CHX && CHB2 <= CHX, "ring slots are sized for the first GEMM's chunks");
  constexpr int Z2 = 2 * NTU * 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int tiles_per_b = (p.T + 127) >> 7;
  const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int b = tile_id / tiles_per_b;
  const int t0w = (tile_id % tiles_per_b) * 128 + wave * 32;
  const int t = t0w + n;
  const bool tvalid = t < p.T;
  const int rows_valid = min(max(p.T - t0w, 0), 32);
  char* stg = smem + 2 * CHX + wave * STG_BYTES;
  const int64_t row0 = (int64_t)b * p.T + t0w;

  const int cpt = Z2 / T_::CK;                 // chunks per tap
  const int nqa = p.ktaps * cpt;
  constexpr int nqb1 = NTU / MTB;
  const int nqb2 = p.Sp / T_::CK;
  const int nq_total = nqa + nqb1 + nqb2;
  auto chunk_src = [&](int qi, int& bytes) -> const char* {
    if (qi < nqa) { bytes = CHX; return p.w_x + (int64_t)qi * CHX; }
    if (qi < nqa + nqb1) { bytes = CHB1; return p.w_uo + (int64_t)(qi - nqa) * CHB1; }
    bytes = CHB2;
    return p.w_us + (int64_t)(qi - nqa - nqb1) * CHB2;
  };
  auto dma = [&](int qi) {
    int bytes;
    const char* src = chunk_src(qi, bytes);
    dma_chunk(src, smem + (qi & 1) * CHX, bytes, wave, lane);
  };

  frag Bn[4], Bc[4];
  auto load_B = [&](const char* base, int64_t stride_e, int col_chunk, int shift, frag (&Bf)[4]) {
    const int ts = t + shift;
    const bool ok = tvalid && ts >= 0 && ts < p.T;
    const char* src = base + (((int64_t)b * p.T + (ok ? ts : 0)) * stride_e) * ES + col_chunk * 128 + h * 16;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      if (ok) {
        Bf[blk] = *(const frag*)(src + blk * 32);
      } else {
        frag zf = {};
        Bf[blk] = zf;
      }
    }
  };
  auto load_B_a = [&](int q, frag (&Bf)[4]) {   // GEMM A: dz_l rows shifted forward in time (transpose of the causal conv)
    const int tap = q / cpt;
    load_B(p.dz, p.dz_stride, q - tap * cpt, (p.ktaps - 1 - tap) * p.dilation, Bf);
  };

  // ---- GEMM A: acc_x = sum_tap W1_tap^T dz_l ----------------------------------------------------------------------
  f32x16 accx[NTX];
#pragma unroll
  for (int m = 0; m < NTX; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[m][r] = 0.f;
  constexpr int NPX = StagePasses<NTX, E>::N;
  f32x4 pre_x[NPX][8];   // residual rows dx_{l+1}-hat: fetched now, they arrive under the MFMAs
  if (rows_valid > 0) stage_fetch_tiles<E, NTX>(pre_x, p.g_next + row0 * NTX * 32 * ES, (int64_t)NTX * 32 * ES, rows_valid, lane);
  dma(0);
  load_B_a(0, Bn);
  for (int q = 0; q < nqa; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
    dma(q + 1);                                   // nq_total > nqa: there is always a next chunk
    if (q + 1 < nqa) load_B_a(q + 1, Bn);
    gemm_chunk<4 * NTX, NTX, 4>(smem + (q & 1) * CHX + lane * 16, Bc, accx);
  }

  // ---- epilogue A: dx_l-hat = alpha * (acc + residual); stored once, kept as the operand of GEMM B1 ---------------------
  frag xf[NKB];
  {
    f32x16 res[NTX];
    if (rows_valid > 0) stage_unpack_tiles<E, NTX>(stg, res, pre_x, lane);
#pragma unroll
    for (int m = 0; m < NTX; ++m) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accx[m][r] = p.alpha * (accx[m][r] + (rows_valid > 0 ? res[m][r] : 0.f));
      frag tmp[KBU];
      acc_to_frags(accx[m], tmp);
#pragma unroll
      for (int s = 0; s < KBU; ++s) xf[m * KBU + s] = tmp[s];
    }
    if (rows_valid > 0) stage_store_tiles<E, NTX>(stg, accx, p.g_out + row0 * NTX * 32 * ES, (int64_t)NTX * 32 * ES, rows_valid, lane);
  }

  // ---- GEMM B: du = W_out^T dx_l-hat (operand from registers) + W_skip^T dskip (operand from memory) -------------------
  f32x16 accu[NTU];
#pragma unroll
  for (int m = 0; m < NTU; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accu[m][r] = 0.f;
  constexpr int NPU = StagePasses<NTU, E>::N;
  f32x4 pre_a[NPU][8], pre_b[NPU][8];   // saved pre-activations z_{l-1} (tanh half, sigmoid half)
  if (rows_valid > 0) {
    const char* zrow = p.z_prev + row0 * Z2 * ES;
    stage_fetch_tiles<E, NTU>(pre_a, zrow, (int64_t)Z2 * ES, rows_valid, lane);
    stage_fetch_tiles<E, NTU>(pre_b, zrow + (int64_t)NTU * 32 * ES, (int64_t)Z2 * ES, rows_valid, lane);
  }
#pragma unroll
  for (int q1 = 0; q1 < nqb1; ++q1) {
    const int qi = nqa + q1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    dma(qi + 1);                                  // nqb2 >= 1
    if (q1 + 1 == nqb1) load_B(p.dskip, p.Sp, 0, 0, Bn);
    gemm_chunk<MTB * NKB, MTB, NKB, true>(smem + (qi & 1) * CHX + lane * 16, xf, *(f32x16(*)[MTB]) & accu[q1 * MTB]);
  }
  for (int q2 = 0; q2 < nqb2; ++q2) {
    const int qi = nqa + nqb1 + q2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
    if (qi + 1 < nq_total) {
      dma(qi + 1);
      load_B(p.dskip, p.Sp, q2 + 1, 0, Bn);
    }
    gemm_chunk<4 * NTU, NTU, 4>(smem + (qi & 1) * CHX + lane * 16, Bc, accu);
  }
  if (rows_valid <= 0) return;

  // ---- epilogue B: gate backward (modules.py:154: u = tanh(a) * sigmoid(b)):  da = du s (1 - th^2),  db = du th s (1 - s) ------
  f32x16 za[NTU], zg[NTU];
  stage_unpack_tiles<E, NTU>(stg, za, pre_a, lane);
  stage_unpack_tiles<E, NTU>(stg, zg, pre_b, lane);
#pragma unroll
  for (int m = 0; m < NTU; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float th, sg;
      if constexpr (sizeof(E) == 4) {
        th = tanhf(za[m][r]);
        sg = 1.0f / (1.0f + expf(-zg[m][r]));
      } else {
        const float ea = __builtin_amdgcn_exp2f(fmaxf(za[m][r], -15.0f) * -2.885390081777927f);
        th = (1.0f - ea) * fast_rcp(1.0f + ea);
        sg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(zg[m][r] * -1.4426950408889634f));
      }
      const float du = accu[m][r];
      za[m][r] = du * sg * (1.0f - th * th);
      zg[m][r] = du * th * sg * (1.0f - sg);
    }
  }
  char* orow = p.dz_prev + row0 * p.dz_stride * ES;
  stage_store_tiles<E, NTU>(stg, za, orow, p.dz_stride * ES, rows_valid, lane);
  stage_store_tiles<E, NTU>(stg, zg, orow + (int64_t)NTU * 32 * ES, p.dz_stride * ES, rows_valid, lane);
}


// ---- 16-bit storage: the round-5 form of the same launch ---------------------------------------------------------------------------
// The per-layer backward launches are bound by what a CU takes in -- a residual workgroup pulls 864 KB through L2 in ~20 us whether it
// has the CU to itself or shares it (profiles/EXPERIMENT_LOG.md, round 5) -- and the gate launch by HBM bytes (4.5 TB/s).  Fusing K_X of
// layer l with K_U of layer l-1 removes the re-read of dx_l-hat (32.8 of the gate launch's 164 MB at C2) and one launch boundary per
// layer.  The round-1 kernel above did that on one workgroup per CU with plain loads and a drain per chunk and lost (118 against 110 us);
// this one is built like csrc/gemm_tm.hip's two-workgroups-per-CU shape:
//   * 4 waves x 32 time columns, TWO workgroups per CU (64 KiB of LDS each: a two-slot weight ring that doubles as staging area);
//   * phase A (K_X): the interleaved tap order and packed weights of wae_gemm_tm mode 1 (same accumulation order: dx-hat is bitwise
//     what the two-launch path stores), operand fragments two chunks ahead by asm loads under counted waits;
//   * epilogue A walks the tiles two at a time (residual rows of the next pair fetched under this pair's math), stores dx-hat once and
//     keeps it, rounded exactly as stored, as the B operand of phase B1 (W_out^T, k in accumulator-row order);
//   * phase B2 (W_skip^T dS) streams dS like phase A streams dz; the saved pre-activations arrive under B1 / B2;
//   * epilogue B = the pairwise gate-derivative epilogue of wae_gemm_tm mode 2.
// One chunk sequence runs through all three GEMMs (the weight ring never drains between them).
// vmcnt(0) that carries the three operand groups as read-write operands: the redundant requests at the end of a chunk loop are never
// consumed, so to the compiler their destinations are dead the moment they are issued -- it hoisted the epilogue's address arithmetic
// into those registers above a bare s_waitcnt, and the loads, landing afterwards, turned the addresses into wild pointers
// (tools/check_asm_regs.py finds it; three of the four instantiations had it).  With the groups as operands they live until the wait.
template <typename F>
__device__ __forceinline__ void drain_groups(F (&a)[4], F (&b)[4], F (&c)[4]) {
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(c[0]), "+v"(c[1]),
                 "+v"(c[2]), "+v"(c[3])
               :
               : "memory");
}

#ifdef WAE_GBP_STAMPS
// diagnostic build (tools/stamps_gbp.py): per workgroup 8 x u64 of wave 0: [0] prologue, [1] phase A, [2] epilogue A, [3] B1, [4] B2,
// [5] epilogue B (clocks, s_memtime), [6] life in 10-ns ticks (s_memrealtime), [7] 1
static unsigned long long* g_gbp_stamps = nullptr;
extern "C" void wae_debug_set_gbp_stamps(unsigned long long* dev_buf) { g_gbp_stamps = dev_buf; }
#define GBP_TICK(v) __builtin_amdgcn_sched_barrier(0); const unsigned long long v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#else
#define GBP_TICK(v) do { } while (0)
#endif

template <typename E, int NTX, int NTU, bool FOLD = false>
__global__ void __launch_bounds__(256, 2) glu_bwd_pair_kernel(GbArgs p) {
  static_assert(sizeof(E) == 2 && NTX % 2 == 0 && NTU % 2 == 0, "16-bit storage, pairwise epilogues");
#ifdef WAE_GBP_STAMPS
  const unsigned long long gw0 = __builtin_amdgcn_s_memrealtime();
#endif
  GBP_TICK(g0);
  using T_ = ET<E>;
  using frag = typename T_::frag;
  constexpr int ES = sizeof(E);
  constexpr int KBU = T_::KBU;                   // 2
  constexpr int CHX = NTX * 4 * 1024;            // phase-A chunk = ring slot
  constexpr int NKB = NTX * KBU;                 // 16-byte k-blocks of GEMM B1 (K = Rp)
  constexpr int MTB = (CHX / (NKB * 1024) >= 2) ? 2 : 1;
  constexpr int CHB1 = MTB * NKB * 1024;
  constexpr int CHB2 = NTU * 4 * 1024;
  static_assert(CHB1 <= CHX && CHB2 <= CHX && NTU % MTB == 0, "ring slots are sized for the first GEMM's chunks");
  constexpr int Z2 = 2 * NTU * 32;
  constexpr int STGB = 4096;                     // per-wave staging tile (row pitch 128 B)
  static_assert(4 * STGB <= CHX, "the staging tiles of the four waves fit one ring slot");
  constexpr int CHC = 2 * 4 * 1024;              // FOLD: a column block of the dc weights (Ccp = 64: two tiles x four k-blocks)
  constexpr int SLOT = CHX + (FOLD ? CHC : 0);   // ring slot: a phase-A chunk [+ the dc block that rides on shift-0 chunks]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int tiles_per_b = (p.T + 127) >> 7;
  const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int b = tile_id / tiles_per_b;
  const int t0w = (tile_id % tiles_per_b) * 128 + wave * 32;
  const int t = t0w + n;
  const bool tvalid = t < p.T;
  const int rows_valid = min(max(p.T - t0w, 0), 32);
  const int64_t row0 = (int64_t)b * p.T + t0w;

  const int cpt = Z2 / T_::CK;                   // column blocks per tap
  const int nqa = p.ktaps * cpt;
  constexpr int nqb1 = NTU / MTB;
  const int nqb2 = p.Sp / T_::CK;
  const int nq_total = nqa + nqb1 + nqb2;
  auto dma = [&](int qi) {                       // chunk qi of the whole sequence -> ring slot qi & 1
#ifdef WAE_GBP_NOW
    return;                                      // timing-only: no weight DMA at all
#endif
    char* dst = smem + (qi & 1) * SLOT;
    qi = min(qi, nq_total - 1);                  // (past the end: the last chunk again, into the slot nobody reads -- constant counts)
    const char* src;
    int bytes;
    if constexpr (FOLD) {                        // (three taps: chunk qi = column block qi / 3 of tap qi % 3; tap 2 has shift 0)
      if (qi < nqa && qi % 3 == 2) dma_chunk(p.w_c + (int64_t)(qi / 3) * CHC, dst + CHX, CHC, wave, lane);
    }
    if (qi < nqa) { bytes = CHX; src = p.w_x + (int64_t)qi * CHX; }
    else if (qi < nqa + nqb1) { bytes = CHB1; src = p.w_uo + (int64_t)(qi - nqa) * CHB1; }
    else { bytes = CHB2; src = p.w_us + (int64_t)(qi - nqa - nqb1) * CHB2; }
#ifdef WAE_GBP_HALFW
    bytes /= 2;                                  // timing-only: half the weight bytes per chunk (what a 256-column workgroup would pull)
#endif
    dma_chunk(src, dst, bytes, wave, lane);
  };

  // operand fragments by asm loads, two chunks ahead (csrc/gemm_tm.hip: TM_ASM_B); clamped address, zero fill at use
  auto a_addr = [&](int q, bool& ok) -> const char* {      // phase A, chunk q = column block q / ktaps of tap q % ktaps
    q = min(q, nqa - 1);
    const int tap = q % p.ktaps, cb = q / p.ktaps;
    const int ts = t + (p.ktaps - 1 - tap) * p.dilation;
    ok = tvalid && ts < p.T;
    return p.dz + (((int64_t)b * p.T + (ok ? ts : 0)) * p.dz_stride) * ES + cb * 128 + h * 16;
  };
  auto s_addr = [&](int q) -> const char* {                // phase B2, chunk q of dS
    q = min(q, nqb2 - 1);
    return p.dskip + (((int64_t)b * p.T + (tvalid ? t : 0)) * p.Sp) * ES + q * 128 + h * 16;
  };
  auto request = [&](frag (&G)[4], const char* src) {
#ifdef WAE_GBP_NOB
    return;                                      // timing-only: no operand requests (dz taps, dS)
#endif
#ifdef WAE_GBP_LINEB
    // timing-only: the same bytes as whole 128-byte lines (8 rows per request) -- what LDS-staged pieces would ask the memory system for
    src += ((lane >> 3) - (lane & 31)) * (int64_t)(p.dz_stride * ES) + ((lane & 7) - (lane >> 5)) * 16;
    gload_async<0>(G[0], src); gload_async<0>(G[1], src + 8 * p.dz_stride * ES); gload_async<0>(G[2], src + 16 * p.dz_stride * ES);
    gload_async<0>(G[3], src + 24 * p.dz_stride * ES);
    return;
#endif
    gload_async<0>(G[0], src); gload_async<32>(G[1], src); gload_async<64>(G[2], src); gload_async<96>(G[3], src);
  };
  auto zero_unless = [&](frag (&G)[4], bool ok) {
    const frag z = {};
#pragma unroll
    for (int i = 0; i < 4; ++i) G[i] = ok ? G[i] : z;
  };

  // ---- phase A: acc_x = sum_tap W1_tap^T dz_l ------------------------------------------------------------------------------------
  f32x16 accx[NTX];
#pragma unroll
  for (int m = 0; m < NTX; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[m][r] = 0.f;
  f32x4 fa[8], fb[8];
  frag G0[4] = {}, G1[4] = {}, G2[4] = {};
  bool k0, k1, k2;
  dma(0);
  request(G0, a_addr(0, k0));
  request(G1, a_addr(1, k1));
  [[maybe_unused]] f32x16 accd[2];               // FOLD: this tile's Wc_l^T dz_l (64 conditioning channels)
  if constexpr (FOLD) {
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) accd[m][r] = 0.f;
  }
  auto step_a = [&](auto shift0, int q, frag (&Gc)[4], bool kc, frag (&Gl)[4], bool& kl) {
    wait_vmcnt_frags<4>(Gc);
    __builtin_amdgcn_s_barrier();     // chunk q visible; every wave is past its reads of chunk q-1, whose slot is refilled now
    zero_unless(Gc, kc);
    dma(q + 1);                       // (q + 1 == nqa: the first chunk of phase B1)
    request(Gl, a_addr(q + 2, kl));
    gemm_chunk<4 * NTX, NTX, 4>(smem + (q & 1) * SLOT + lane * 16, Gc, accx);
    if constexpr (FOLD && decltype(shift0)::value != 0)      // the shift-0 tap's fragments are dz_l[t]: the dc block rides on them
      gemm_chunk<8, 2, 4>(smem + (q & 1) * SLOT + CHX + lane * 16, Gc, accd);
  };
  GBP_TICK(g1);
  for (int q = 0; q < nqa; q += 3) {
    step_a(IntC<0>{}, q, G0, k0, G2, k2);
    if (q + 1 < nqa) step_a(IntC<0>{}, q + 1, G1, k1, G0, k0);
    if (q + 2 < nqa) step_a(IntC<1>{}, q + 2, G2, k2, G1, k1);      // (FOLD: three taps, q + 2 is the shift-0 tap)
  }
  drain_groups(G0, G1, G2);                          // chunk nqa (B1's first) has landed; the redundant operand requests are retired
  __builtin_amdgcn_s_barrier();                      // every wave has left slot (nqa - 1) & 1: it is the staging area of epilogue A
  GBP_TICK(g2);

  // ---- epilogue A: dx_l-hat = alpha * (acc + residual), stored once, kept as the operand of GEMM B1 ------------------------------
  // The residual rows (dx_{l+1}-hat of this tile) arrive as operand-shaped 16-byte fragments -- lane (n, h): row n, bytes
  // [32 f + 16 h, +16) -- ALL 2 NTX of them requested at once (the three operand groups of phase A are dead: their registers), and
  // one half-swap puts them into the accumulator layout (csrc/wae_common.hpp: residual_to_acc_layout, what the layer kernel does with
  // its residual): ONE exposed round trip for the whole tile and no LDS staging on the way in.  The pairwise form fetched pair
  // i + 1 under the math of pair i, i.e. paid a round trip per pair (stamps: 16.5 k clocks for this epilogue).
  frag xf[NKB];
  {
    // the lane id, laundered: hipcc otherwise forms the staging passes' per-lane addresses at the top of the kernel, carries them
    // through phase A in registers it does not have, and reloads them from scratch once per tile pair -- a scratch reload waits for
    // EVERY outstanding request, the previous pair's stores included (stamps: 27.6 k clocks against 9.7 k in the two-launch kernel)
    int le = lane;
    asm volatile("" : "+v"(le));
    const int ne = le & 31, he = le >> 5;
    frag rs[2 * NTX];
    // (rows at or behind the clip's end read its last row: never stored)
    const char* rp = p.g_next + ((int64_t)b * p.T + min(t0w + ne, p.T - 1)) * (int64_t)(NTX * 32 * ES) + he * 16;
    char* stg = smem + ((nqa - 1) & 1) * SLOT + wave * STGB;
    if constexpr (FOLD) {
      // the first half of the residual fragments and the running dc sum travel together; the dc tiles leave first (their 32 registers
      // are what the second half of the fragments lands in)
      [[maybe_unused]] f32x4 w0[4], w1[4];
      const bool add = (p.dc_mode & 1) != 0;
      if (add && rows_valid > 0) {
        rmw_fetch(w0, p.dc_acc + row0 * 64, 0, rows_valid, le);
        rmw_fetch(w1, p.dc_acc + row0 * 64, 1, rows_valid, le);
      }
#pragma unroll
      for (int f = 0; f < NTX; ++f) rs[f] = *(const frag*)(rp + f * 32);
      __builtin_amdgcn_sched_barrier(0);
      if (rows_valid > 0) {
        char* o16 = (p.dc_mode & 2) ? p.dc_out + row0 * 64 * ES : nullptr;
        stage_rmw_tile<E>(stg, accd[0], w0, add, p.dc_acc + row0 * 64, o16, 0, rows_valid, le);
        stage_rmw_tile<E>(stg, accd[1], w1, add, p.dc_acc + row0 * 64, o16, 1, rows_valid, le);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = NTX; f < 2 * NTX; ++f) rs[f] = *(const frag*)(rp + f * 32);
    } else {
#pragma unroll
      for (int f = 0; f < 2 * NTX; ++f) rs[f] = *(const frag*)(rp + f * 32);
    }
    __builtin_amdgcn_sched_barrier(0);
    residual_to_acc_layout(rs);
    char* orow = p.g_out + row0 * NTX * 32 * ES;
#pragma unroll
    for (int pr = 0; pr < NTX / 2; ++pr) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 r4 = residual_piece<E>(rs, 2 * pr + i, g);
          f32x16& a = accx[2 * pr + i];
          a[4 * g] = p.alpha * (a[4 * g] + r4.x); a[4 * g + 1] = p.alpha * (a[4 * g + 1] + r4.y);
          a[4 * g + 2] = p.alpha * (a[4 * g + 2] + r4.z); a[4 * g + 3] = p.alpha * (a[4 * g + 3] + r4.w);
        }
        frag tmp[KBU];
        acc_to_frags(accx[2 * pr + i], tmp);
#pragma unroll
        for (int s = 0; s < KBU; ++s) xf[(2 * pr + i) * KBU + s] = tmp[s];
      }
      if (rows_valid > 0) stage_store_pass<E, 2, 128>(stg, &accx[2 * pr], orow + pr * 64 * ES, (int64_t)NTX * 32 * ES, rows_valid, le);
    }
  }
  if (p.last) {                                      // layer 0: nothing below to gate (the B1 chunk requested by the last step lands first)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }

  GBP_TICK(g3);
  // ---- GEMM B: du = W_out^T dx_l-hat (operand from registers) + W_skip^T dskip (operand from memory) -------------------------------
  f32x16 accu[NTU];
#pragma unroll
  for (int m = 0; m < NTU; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accu[m][r] = 0.f;
  const char* zrow = p.z_prev + row0 * Z2 * ES;
#pragma unroll
  for (int q1 = 0; q1 < nqb1; ++q1) {
    const int qi = nqa + q1;
    if (q1 > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (chunk nqa landed before epilogue A)
    __builtin_amdgcn_s_barrier();                                   // q1 == 0: every wave is done staging in the slot refilled now
    dma(qi + 1);
    if (q1 == 0) {    // dS of the first two B2 chunks: they arrive under B1
      // (fresh values: an asm request reads its destination ("+v"), which would keep the three groups of phase A alive through
      //  epilogue A -- 48 registers the tile pairs there do not have)
      const frag zf = {};
#pragma unroll
      for (int i = 0; i < 4; ++i) { G0[i] = zf; G1[i] = zf; }
      request(G0, s_addr(0));
      request(G1, s_addr(1));
    }
    gemm_chunk<MTB * NKB, MTB, NKB, true>(smem + (qi & 1) * SLOT + lane * 16, xf, *(f32x16(*)[MTB]) & accu[q1 * MTB]);
  }
  // B2: VMEM order per step j: [DMA(next)][dS(j + 2)]; at the top of step j >= 1 only dS(j + 1) may be outstanding
  // (the first step is peeled at compile time: a run-time branch around an asm request or its counted wait is what lets hipcc park a
  //  fragment's register elsewhere between the request and its wait -- csrc/wae_common.hpp: gload_async)
  auto step_s = [&](auto first, int j, frag (&Gc)[4], frag (&Gl)[4]) {
    constexpr bool FIRST = decltype(first)::value != 0;
    const int qi = nqa + nqb1 + j;
    if constexpr (FIRST) asm volatile("s_waitcnt vmcnt(0)" : "+v"(Gc[0]), "+v"(Gc[1]), "+v"(Gc[2]), "+v"(Gc[3]));
    else wait_vmcnt_frags<4>(Gc);
    __builtin_amdgcn_s_barrier();
    zero_unless(Gc, tvalid);
    if constexpr (FIRST) {
      // the saved pre-activations of the first tile pair (xf is dead: their registers are free now); they arrive under B2.  Older than
      // every request of the steps below: the counted waits only get stricter.
      if (rows_valid > 0) {
        stage_fetch_pass<E, 2>(fa, zrow, (int64_t)Z2 * ES, rows_valid, lane);
        stage_fetch_pass<E, 2>(fb, zrow + (int64_t)NTU * 32 * ES, (int64_t)Z2 * ES, rows_valid, lane);
      }
    }
    dma(qi + 1);
    if constexpr (FIRST) {
      const frag zf = {};
#pragma unroll
      for (int i = 0; i < 4; ++i) Gl[i] = zf;
    }
    request(Gl, s_addr(j + 2));
    gemm_chunk<4 * NTU, NTU, 4>(smem + (qi & 1) * SLOT + lane * 16, Gc, accu);
  };
  GBP_TICK(g4);
  step_s(IntC<1>{}, 0, G0, G2);
  for (int j = 1; j < nqb2; j += 3) {
    step_s(IntC<0>{}, j, G1, G0);
    if (j + 1 < nqb2) step_s(IntC<0>{}, j + 1, G2, G1);
    if (j + 2 < nqb2) step_s(IntC<0>{}, j + 2, G0, G2);
  }
  drain_groups(G0, G1, G2);                          // the redundant tail requests must not outlive the ring (or their registers)
  __syncthreads();                                   // every wave is done with the weight ring: it becomes the staging area
  GBP_TICK(g5);
  if (rows_valid <= 0) return;

  // ---- epilogue B: gate backward (modules.py:154: u = tanh(a) * sigmoid(b)):  da = du s (1 - th^2),  db = du th s (1 - s) ------------
  {
    int le = lane;                                   // (laundered like epilogue A's)
    asm volatile("" : "+v"(le));
    char* stg = smem + wave * STGB;
    char* orow = p.dz_prev + row0 * p.dz_stride * ES;
#pragma unroll
    for (int pr = 0; pr < NTU / 2; ++pr) {
      f32x16 za[2], zg[2];
      stage_unpack_pass<E, 2, 128>(stg, za, fa, le);
      stage_unpack_pass<E, 2, 128>(stg, zg, fb, le);
      if (pr + 1 < NTU / 2) {
        stage_fetch_pass<E, 2>(fa, zrow + (pr + 1) * 64 * ES, (int64_t)Z2 * ES, rows_valid, le);
        stage_fetch_pass<E, 2>(fb, zrow + ((int64_t)NTU * 32 + (pr + 1) * 64) * ES, (int64_t)Z2 * ES, rows_valid, le);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ea = __builtin_amdgcn_exp2f(fmaxf(za[i][r], -15.0f) * -2.885390081777927f);
          const float th = (1.0f - ea) * fast_rcp(1.0f + ea);
          const float sg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(zg[i][r] * -1.4426950408889634f));
          const float du = accu[2 * pr + i][r];
          za[i][r] = du * sg * (1.0f - th * th);
          zg[i][r] = du * th * sg * (1.0f - sg);
        }
      stage_store_pass<E, 2, 128>(stg, za, orow + pr * 64 * ES, p.dz_stride * ES, rows_valid, le);
      stage_store_pass<E, 2, 128>(stg, zg, orow + ((int64_t)NTU * 32 + pr * 64) * ES, p.dz_stride * ES, rows_valid, le);
    }
  }
#ifdef WAE_GBP_STAMPS
  GBP_TICK(g6);
  if (p.stamps && threadIdx.x == 0) {
    unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
    o[0] = g1 - g0; o[1] = g2 - g1; o[2] = g3 - g2; o[3] = g4 - g3; o[4] = g5 - g4; o[5] = g6 - g5;
    o[6] = __builtin_amdgcn_s_memrealtime() - gw0; o[7] = 1;
  }
#endif
}

template <typename E, int NTX, int NTU>
static int launch_gb_pair(const GbArgs& a, hipStream_t st) {
  constexpr int CHX = NTX * 4 * 1024;
  const int tiles = (a.T + 127) / 128;
  if (a.w_c) {                                       // dc folded in: 8 KiB more per ring slot
    const size_t lds = 2 * (CHX + 8192);
    static WaeLdsCache lds_cache_f;
    if (int rc = wae_ensure_lds((const void*)glu_bwd_pair_kernel<E, NTX, NTU, true>, lds_cache_f, lds, "glu_bwd_pair"); rc != WAE_OK) return rc;
    hipLaunchKernelGGL((glu_bwd_pair_kernel<E, NTX, NTU, true>), dim3(a.B * tiles), dim3(256), lds, st, a);
    return wae_check_launch("glu_bwd_pair");
  }
  const size_t lds = 2 * CHX;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)glu_bwd_pair_kernel<E, NTX, NTU>, lds_cache, lds, "glu_bwd_pair"); rc != WAE_OK) return rc;
  hipLaunchKernelGGL((glu_bwd_pair_kernel<E, NTX, NTU>), dim3(a.B * tiles), dim3(256), lds, st, a);
  return wae_check_launch("glu_bwd_pair");
}

template <typename E, int NTX, int NTU>
static int launch_gb(const GbArgs& a, hipStream_t st) {
  constexpr int CHX = NTX * 4 * 1024;
  const size_t lds = 2 * CHX + 4 * STG_BYTES;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)glu_bwd_fused_kernel<E, NTX, NTU>, lds_cache, lds, "glu_bwd_fused"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 127) / 128;
  hipLaunchKernelGGL((glu_bwd_fused_kernel<E, NTX, NTU>), dim3(a.B * tiles), dim3(256), lds, st, a);
  return wae_check_launch("glu_bwd_fused");
}

template <typename E>
static int dispatch_gb(int ntx, int ntu, const GbArgs& a, hipStream_t st) {
  // 16-bit storage: the two-workgroups-per-CU form (w_x in the INTERLEAVED chunk order of wae_gemm_tm mode 1)
  if constexpr (sizeof(E) == 2) {
    if (ntx == 8 && ntu == 6) return launch_gb_pair<E, 8, 6>(a, st);
    if (ntx == 8 && ntu == 4) return launch_gb_pair<E, 8, 4>(a, st);
    if (ntx == 4 && ntu == 2) return launch_gb_pair<E, 4, 2>(a, st);
    if (ntx == 4 && ntu == 4) return launch_gb_pair<E, 4, 4>(a, st);
    wae_set_error("glu_bwd_fused: no 16-bit instance for Rp=%d, Hp=%d (use the two wae_gemm_tm launches)", ntx * 32, ntu * 32);
    return WAE_EUNSUPPORTED;
  } else {
    // fp32 (the round-1 kernel): (Rp/32, Hp/32) pairs of the shipped presets and the test configurations
    if (ntx == 8 && ntu == 6) return launch_gb<E, 8, 6>(a, st);
    if (ntx == 8 && ntu == 4) return launch_gb<E, 8, 4>(a, st);
    if (ntx == 4 && ntu == 1) return launch_gb<E, 4, 1>(a, st);
    if (ntx == 4 && ntu == 2) return launch_gb<E, 4, 2>(a, st);
    if (ntx == 4 && ntu == 3) return launch_gb<E, 4, 3>(a, st);
    if (ntx == 4 && ntu == 4) return launch_gb<E, 4, 4>(a, st);
    wae_set_error("glu_bwd_fused: no instance for Rp=%d, Hp=%d (use the two wae_gemm_tm launches)", ntx * 32, ntu * 32);
    return WAE_EUNSUPPORTED;
  }
}

extern "C" int wae_glu_bwd_fused_supported(int32_t Rp, int32_t Hp) {
  const int x = Rp / 32, u = Hp / 32;
  return (x == 8 && (u == 6 || u == 4)) || (x == 4 && u >= 1 && u <= 4);
}
// ... and for 16-bit storage (the pairwise epilogues need even tile counts)
extern "C" int wae_glu_bwd_fused_supported16(int32_t Rp, int32_t Hp) {
  const int x = Rp / 32, u = Hp / 32;
  return (x == 8 && (u == 6 || u == 4)) || (x == 4 && (u == 2 || u == 4));
}

static int glu_bwd_fused_impl(const wae_glu_bwd_desc* d, const void* dz, int64_t dz_stride, const void* g_next, void* g_out,
                              const void* dskip, const void* z_prev, void* dz_prev, const void* w_x, const void* w_uo,
                              const void* w_us, const void* w_c, float* dc_acc, void* dc_out, int dc_mode, int last, void* stream) {
  WAE_REQUIRE(d && dz && g_next && g_out && dskip && z_prev && dz_prev && w_x && w_uo && w_us, "glu_bwd_fused: null pointer argument");
  WAE_REQUIRE(wae_dtype_ok(d->dtype), "glu_bwd_fused: bad dtype");
  const int ck = wae_is16(d->dtype) ? 64 : 32;
  WAE_REQUIRE(d->B > 0 && d->T > 0 && d->Rp % 128 == 0 && d->Hp % 32 == 0 && d->Sp % ck == 0 && d->Sp > 0 && d->ktaps >= 1 &&
                  d->dilation >= 1 && (2 * d->Hp) % ck == 0,
              "glu_bwd_fused: bad sizes");
  WAE_REQUIRE(!w_c || (wae_is16(d->dtype) && d->ktaps == 3 && dc_acc && (!(dc_mode & 2) || dc_out)),
              "glu_bwd_fused_dc: the folded dc needs 16-bit storage, three taps, dc_acc (and dc_out with mode bit 1)");
  WAE_REQUIRE(!last || w_c, "glu_bwd_fused_dc: last = 1 (layer 0) exists for the folded-dc form only");
  GbArgs a;
  a.dz = (const char*)dz; a.g_next = (const char*)g_next; a.g_out = (char*)g_out; a.dskip = (const char*)dskip;
  a.z_prev = (const char*)z_prev; a.dz_prev = (char*)dz_prev; a.w_x = (const char*)w_x; a.w_uo = (const char*)w_uo;
  a.w_us = (const char*)w_us; a.dz_stride = dz_stride; a.alpha = d->alpha; a.B = d->B; a.T = d->T; a.Sp = d->Sp;
  a.ktaps = d->ktaps; a.dilation = d->dilation;
  a.w_c = (const char*)w_c; a.dc_acc = dc_acc; a.dc_out = (char*)dc_out; a.dc_mode = dc_mode; a.last = last;
  a.stamps = nullptr;
#ifdef WAE_GBP_STAMPS
  a.stamps = g_gbp_stamps;
#endif
  hipStream_t st = as_stream(stream);
  if (!(dc_mode & 4)) {     // (dc_mode bit 2: keep the 4-wave kernel -- the A/B and parity handle of tests/ and tools/)
    bool handled = false;
    const int rc = wae_glu_bwd8_launch(a, d->dtype, d->Rp / 32, d->Hp / 32, st, &handled);
    if (rc != WAE_OK || handled) return rc;
  }
  a.dc_mode &= 3;
  if (d->dtype == WAE_BF16) return dispatch_gb<__bf16>(d->Rp / 32, d->Hp / 32, a, st);
  if (d->dtype == WAE_F16) return dispatch_gb<f16>(d->Rp / 32, d->Hp / 32, a, st);
  return dispatch_gb<float>(d->Rp / 32, d->Hp / 32, a, st);
}

extern "C" int wae_glu_bwd_fused(const wae_glu_bwd_desc* d, const void* dz, int64_t dz_stride, const void* g_next, void* g_out,
                                 const void* dskip, const void* z_prev, void* dz_prev, const void* w_x, const void* w_uo,
                                 const void* w_us, void* stream) {
  return glu_bwd_fused_impl(d, dz, dz_stride, g_next, g_out, dskip, z_prev, dz_prev, w_x, w_uo, w_us, nullptr, nullptr, nullptr, 0, 0, stream);
}

extern "C" int wae_glu_bwd_fused_dc(const wae_glu_bwd_desc* d, const void* dz, int64_t dz_stride, const void* g_next, void* g_out,
                                    const void* dskip, const void* z_prev, void* dz_prev, const void* w_x, const void* w_uo,
                                    const void* w_us, const void* w_c, float* dc_acc, void* dc_out, int32_t dc_mode, int32_t last,
                                    void* stream) {
  WAE_REQUIRE(w_c, "glu_bwd_fused_dc: w_c is null (use wae_glu_bwd_fused)");
  return glu_bwd_fused_impl(d, dz, dz_stride, g_next, g_out, dskip, z_prev, dz_prev, w_x, w_uo, w_us, w_c, dc_acc, dc_out, dc_mode, last, stream);
}
