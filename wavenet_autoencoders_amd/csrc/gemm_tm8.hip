// gemm_tm8: the time-major GEMM of csrc/gemm_tm.hip for ONE source whose K dimension is long and whose operand streams from HBM --
// the head's skip contraction  h0 = relu(sqrt(1/L) (sum_l b_skip_l + [W_skip_0 .. W_skip_{L-1}] [u_0; ..; u_{L-1}]))
// (wavenet.py:204-209; K = L * Hp = 4608 at C2: 590 MB of u per launch) -- on the schedule of the fused layer kernel
// (csrc/glu_fwd_static.hip) instead of the two-slot ring of gemm_tm_kernel:
//   * 8 waves x 32 time columns, ONE workgroup per CU: the packed weights enter the CU once per 256 columns (gemm_tm: per 128);
//   * the weights stream through an 8-slot ring of K = 32 HALF-chunks (2 k-blocks x M/32 tiles x 1 KiB), requested D = 7 half-chunks
//     ahead by LDS-DMA; the operand fragments (buffer_load_dwordx4 through a per-clip descriptor: rows past the clip's end read as
//     zeros) are requested NGA = 6 half-chunks ahead into 8 register groups.  gemm_tm_kernel keeps ONE weight chunk and two operand
//     chunks in flight per workgroup; its stamps (profiles/EXPERIMENT_LOG.md, "Stamps of the skip contraction") show every wave
//     standing 10-30 % of a chunk at the chunk-top barrier, waiting for the slowest wave's pieces;
//   * the chunk loop is a run-time loop over bodies of 8 half-chunks: ring slot, operand group, every `s_waitcnt vmcnt(n)` and every
//     ds_read offset are immediates; the A-fragment reads run as one stream over the 8 half-chunks of a body.
// Same packed weight stream, same fragment layouts and the same accumulation order as gemm_tm_kernel: results are bit-identical
// (tests/test_gpu_wide.py, tests/test_gpu_small_kernels.py).
//
// Issue order of one wave's VMEM operations (what the counted waits are derived from):
//   virtual half-chunk c = -D .. nh-1 issues   [DMA(c + D): PPW pieces]  [B(c + NGA): KB fragments]     (negative targets skipped)
//   top of half-chunk c needs   B(c)  (issued in c - NGA, last in its list)   and, where the workgroup meets, DMA(c + BE)
//   (issued in c + BE - D, first in its list; one barrier EARLY because the A reads run ahead into the next half-chunk).
#include "gemm_tm.hpp"

#ifndef WAE_TM8_D
#define WAE_TM8_D 7
#endif
#ifndef WAE_TM8_NGA
#define WAE_TM8_NGA 6
#endif
#ifndef WAE_TM8_BE
#define WAE_TM8_BE 1
#endif
#ifndef WAE_TM8_PD
#define WAE_TM8_PD 4
#endif
// timing-only ablations (tools/time_tm8.py on variant builds; results are wrong when any bit is set):
// 1 no operand requests, 2 no weight DMA, 4 no MFMAs, 8 no A-fragment reads, 16 no barriers
#ifndef WAE_TM8_ABL
#define WAE_TM8_ABL 0
#endif
// cache policy of the operand requests (timing experiments: "sc1", "nt", "sc0 sc1")
#ifndef WAE_TM8_BPOL
#define WAE_TM8_BPOL ""
#endif
// timing-only request shape of the operand: 2 = the same bytes as full 128-byte lines, 8 rows per request (results are wrong)
#ifndef WAE_TM8_BVAR
#define WAE_TM8_BVAR 0
#endif

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for8(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for8<I + 1, N>(f);
  }
}

template <int OFF, typename F>
__device__ __forceinline__ void bload_soff(F& dst, unsigned voff, i32x4 rsrc, unsigned soff) {
  static_assert(sizeof(F) == 16 && OFF >= 0 && OFF < 4096, "one 16-byte fragment, 12-bit offset");
  if constexpr (!(WAE_TM8_ABL & 1))
    asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:%4 " WAE_TM8_BPOL : "+v"(dst) : "v"(voff), "s"(rsrc), "s"(soff), "n"(OFF));
}
template <int CNT>
__device__ __forceinline__ void wait_vm8() {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}
template <int OFF, typename frag>
__device__ __forceinline__ void lds_read_off(frag& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ i32x4 make_srd8(const char* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

}  // namespace

template <typename E, int NT, int MODE, int D, int NGA, int BE, int PD>
__global__ void __launch_bounds__(512, 1) gemm_tm8_kernel(TmArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  static_assert(sizeof(E) == 2 && T_::CK == 64, "16-bit storage only");
  static_assert(MODE == TM_BIAS_RELU, "modes: 3");
  constexpr int NW = 8, KB = 2, NM = NT, ES = 2;
  constexpr int HCB = NM * KB * 1024;          // one half-chunk of packed weights = one ring slot
  constexpr int NSLOT = 8, BODY = 8, NGB = 8;  // ring slots, half-chunks per loop body, operand register groups
  constexpr int NSTEP = KB * NM;               // MFMAs per wave and half-chunk
  constexpr int PPW = HCB / NW / 1024;         // LDS-DMA pieces per wave and half-chunk
  constexpr int NOPS = PPW + KB, SP = NSTEP / NOPS;
  static_assert(PPW >= 1 && PPW * NW * 1024 == HCB, "whole pieces per wave");
  static_assert(NOPS * SP <= NSTEP, "not enough MFMA steps to carry a half-chunk's VMEM issue");
  static_assert(BE == 1 || BE == 2, "the workgroup meets at every, or at every second, half-chunk");
  // ring discipline (csrc/glu_fwd_static.hip): visibility one barrier early (D >= BE + 1), slot reuse (D <= NSLOT - BE);
  // operand groups: B(c + NGA) lands in the group of half-chunk c + NGA - NGB, which must be consumed: NGA < NGB
  static_assert(D >= BE + 1 && D <= NSLOT - BE && NGA >= 1 && NGA < NGB && NGA <= D, "ring discipline");
  static_assert(NSLOT * HCB <= 128 * 1024 && NW * STG_BYTES <= NSLOT * HCB, "LDS budget; the ring doubles as staging area");
  // counted waits: operations issued after B(c) / after the first piece of DMA(c + BE)
  constexpr int AFTER_B = (NGA - 1) * NOPS;
  constexpr int AFTER_DMA = KB + (D - BE - 1) * NOPS;            // (after its LAST piece)
  constexpr int ALLOW_MEET = AFTER_B < AFTER_DMA ? AFTER_B : AFTER_DMA;
  static_assert(AFTER_B < 64 && ALLOW_MEET >= 0, "vmcnt range");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  constexpr int TW = NW * 32;
  const int tiles_per_b = (p.T + TW - 1) / TW;
  const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int b = tile_id / tiles_per_b;
  const int t0w = (tile_id % tiles_per_b) * TW + wave * 32;
  const int t = t0w + n;
  const int rows_valid = min(max(p.T - t0w, 0), 32);

  const int nh = 2 * (p.src_cols[0] / T_::CK);         // half-chunks (host: a multiple of BODY)
  const int nit = nh / BODY;
  const unsigned row_bytes = (unsigned)(p.src_stride[0] * ES);
  const unsigned clip_bytes = (unsigned)p.T * row_bytes;
  const i32x4 srd = make_srd8(p.src[0] + (int64_t)b * clip_bytes, clip_bytes);
  const unsigned voff = (unsigned)t * row_bytes + h * 16;       // t >= T: beyond num_records -> zeros

  float* bias_lds = (float*)(smem + NSLOT * HCB);
  const unsigned lane_off = (unsigned)(wave * PPW * 1024 + lane * 16);
  char* lds_wave = smem + wave * PPW * 1024;
  const char* wbase = p.w;

  // weights of half-chunk `cc` (clamped: past the end the last one again, into a slot nobody reads -- the counts stay constant)
  auto dma_piece8 = [&](int cc, auto slotc, auto kc) {
    constexpr int slot = decltype(slotc)::value, k = decltype(kc)::value;
    const int cs = min(cc, nh - 1);
    const char* sb = wbase + (int64_t)cs * HCB;
    if constexpr (!(WAE_TM8_ABL & 2))
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sb + lane_off),
                                     (__attribute__((address_space(3))) void*)(lds_wave + slot * HCB), 16, k * 1024, 0);
  };
  frag S[NGB][KB] = {};
  // operand fragment f of the half-chunk that sits `rel` half-chunks behind the body's first one (soff = the body's column offset)
  auto b_frag = [&](auto relc, auto fc, unsigned soff) {
    constexpr int rel = decltype(relc)::value, f = decltype(fc)::value;
#if WAE_TM8_BVAR == 3 || WAE_TM8_BVAR == 4
    // timing only: u as if stored layer-major, [layer][b][t][192] (rows 384 B apart; chunk q = layer q / 3, columns (q % 3) * 64)
    const unsigned q = soff / 128 + rel / 2, l = q / 3;
    const unsigned lb = (unsigned)p.B * p.T * 384u;
    const unsigned so = l * lb + (unsigned)b * p.T * 384u + (q - 3 * l) * 128 + (rel % 2) * 64 + f * 32;
    const i32x4 srdw = make_srd8(p.src[0], 0x7fffffffu);
#if WAE_TM8_BVAR == 3
    const unsigned vo = (unsigned)t * 384u + h * 16;
    bload_soff<0>(S[rel % NGB][f], vo, srdw, so);
#else
    constexpr int r = (rel % 2) * 2 + f;
    const unsigned vo = (unsigned)(t0w + 8 * r + (lane >> 3)) * 384u + (lane & 7) * 16;
    bload_soff<0>(S[rel % NGB][f], vo, srdw, so - ((rel % 2) * 64 + f * 32));
#endif
#elif WAE_TM8_BVAR == 2
    constexpr int r = (rel % 2) * 2 + f;
    const unsigned vo = (unsigned)(t0w + 8 * r + (lane >> 3)) * row_bytes + (lane & 7) * 16;
    bload_soff<(rel / 2) * 128>(S[rel % NGB][f], vo, srd, soff);
#else
    bload_soff<rel * 64 + f * 32>(S[rel % NGB][f], voff, srd, soff);
#endif
  };

  // ---- prologue: the bias table first, then virtual half-chunks -D .. -1 in the steady-state order ---------------------------------
  // (the table travels by asm requests like everything else: hipcc cannot count the asm requests behind a load of its own and would
  //  drain them all -- the whole prologue prefetch -- in front of the first use of the table)
  f32x4 tb = {};
  const unsigned tb_off = lane * 16;
  if (wave == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(tb) : "v"(tb_off), "s"(p.aux));
  static_for8<0, D>([&](auto vc) {
    constexpr int v = decltype(vc)::value - D;        // virtual half-chunk index, negative
    static_for8<0, PPW>([&](auto kc) { dma_piece8(v + D, IntC<(v + D) % NSLOT>{}, kc); });
    if constexpr (v + NGA >= 0) static_for8<0, KB>([&](auto fc) { b_frag(IntC<v + NGA>{}, fc, 0u); });
  });
  constexpr int PRO_OPS = D * PPW + NGA * KB;
  static_assert(PRO_OPS < 64, "vmcnt range");
  if (wave == 0) {
    const unsigned tw = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)bias_lds + lane * 16;
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tb) : "n"(PRO_OPS));
    asm volatile("ds_write_b128 %0, %1" ::"v"(tw), "v"(tb));
  }

  unsigned a_base[2];
  a_base[0] = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
  a_base[1] = a_base[0] + 4 * HCB;
  static_assert(4 * HCB <= 65536, "four slots per ds_read base register");
  auto a_read = [&](auto jc, auto ic, frag& dst) {     // A fragment block I of the half-chunk in slot j
    constexpr int j = decltype(jc)::value, I = decltype(ic)::value;
    lds_read_off<(j % 4) * HCB + I * 1024>(dst, a_base[j / 4]);
  };

  f32x16 acc[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the table writes are retired; the first barrier publishes them
  // accumulators start from the bias table, read back by asm ds_reads: a compiler-visible LDS read behind LDS-DMA pieces makes hipcc
  // drain vmcnt -- the whole prologue prefetch -- in front of it
  const unsigned tab_addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)bias_lds + h * 16;
  auto init_tile = [&](auto mc) {
    constexpr int m = decltype(mc)::value;
    f32x4 v0, v1, v2, v3;
    lds_read_off<m * 128 + 0>(v0, tab_addr);
    lds_read_off<m * 128 + 32>(v1, tab_addr);
    lds_read_off<m * 128 + 64>(v2, tab_addr);
    lds_read_off<m * 128 + 96>(v3, tab_addr);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    acc[m][0] = v0.x; acc[m][1] = v0.y; acc[m][2] = v0.z; acc[m][3] = v0.w;
    acc[m][4] = v1.x; acc[m][5] = v1.y; acc[m][6] = v1.z; acc[m][7] = v1.w;
    acc[m][8] = v2.x; acc[m][9] = v2.y; acc[m][10] = v2.z; acc[m][11] = v2.w;
    acc[m][12] = v3.x; acc[m][13] = v3.y; acc[m][14] = v3.z; acc[m][15] = v3.w;
  };
  constexpr int NG = BODY * NSTEP;
  for (int it = 0; it < nit; ++it) {
    const int c0 = it * BODY;
    const unsigned soff = (unsigned)it * (BODY * 64);
    frag a[PD];
    static_for8<0, BODY>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      constexpr bool meet = (j % BE) == 0;
      wait_vm8<meet ? ALLOW_MEET : AFTER_B>();
      if constexpr (meet && !(WAE_TM8_ABL & 16)) __builtin_amdgcn_s_barrier();
      asm volatile("" : "+v"(S[j][0]), "+v"(S[j][1]));       // B(c) has landed: every use comes after this point
      if constexpr (j == 0) {
        if (it == 0) static_for8<0, NM>(init_tile);      // rows 8g + 4h + j of tile m <- table[32 m + 8 g + 4 h + j]
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(WAE_TM8_ABL & 8)) static_for8<0, PD>([&](auto ic) { a_read(IntC<0>{}, ic, a[decltype(ic)::value]); });
        else static_for8<0, PD>([&](auto ic) { a[decltype(ic)::value] = S[0][0]; });
      }
      static_for8<0, NSTEP>([&](auto ic) {
        constexpr int I = decltype(ic)::value, G = j * NSTEP + I, AI = G % PD;
        constexpr int remaining = NG - 1 - G;
        if constexpr (!(WAE_TM8_ABL & 8)) lds_wait<(remaining < PD - 1 ? remaining : PD - 1)>(a[AI]);
        if constexpr (!(WAE_TM8_ABL & 4)) mma32(acc[I % NM], a[AI], S[j][I / NM]);
        if constexpr (remaining >= PD && !(WAE_TM8_ABL & 8)) {
          constexpr int G2 = G + PD;
          a_read(IntC<G2 / NSTEP>{}, IntC<G2 % NSTEP>{}, a[AI]);
        }
        if constexpr (I % SP == 0 && I / SP < NOPS) {        // this half-chunk's requests, spread over its MFMAs: pieces first
          constexpr int k = I / SP;
          if constexpr (k < PPW) dma_piece8(c0 + j + D, IntC<(j + D) % NSLOT>{}, IntC<k>{});
          else b_frag(IntC<j + NGA>{}, IntC<k - PPW>{}, soff);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  // the redundant tail requests must not outlive the ring -- nor their registers (csrc/gemm_tm.hip: the drain carries the groups)
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(S[0][0]), "+v"(S[0][1]), "+v"(S[1][0]), "+v"(S[1][1]), "+v"(S[2][0]), "+v"(S[2][1]), "+v"(S[3][0]), "+v"(S[3][1]),
                 "+v"(S[4][0]), "+v"(S[4][1]), "+v"(S[5][0]), "+v"(S[5][1]), "+v"(S[6][0]), "+v"(S[6][1]), "+v"(S[7][0]), "+v"(S[7][1])
               :
               : "memory");
  __syncthreads();      // every wave is done with the weight ring: it becomes the staging area
  if (rows_valid <= 0) return;

  char* stg = smem + wave * STG_BYTES;
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = fmaxf(acc[m][r] * p.alpha, 0.f);
  char* orow = p.out + ((int64_t)b * p.T + t0w) * p.out_stride * ES;
  stage_store_tiles<E, NT>(stg, acc, orow, p.out_stride * ES, rows_valid, lane);
}

namespace {

template <typename E, int NT, int MODE>
int launch_tm8(const TmArgs& a, hipStream_t st) {
  constexpr int HCB = NT * 2 * 1024;
  auto kern = gemm_tm8_kernel<E, NT, MODE, WAE_TM8_D, WAE_TM8_NGA, WAE_TM8_BE, WAE_TM8_PD>;
  const size_t lds = (size_t)8 * HCB + (size_t)NT * 32 * 4;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)kern, lds_cache, lds, "gemm_tm8"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(a.B * tiles), dim3(512), lds, st, a);
  return wae_check_launch("gemm_tm8");
}

}  // namespace

int wae_gemm_tm8_launch(const TmArgs& a, int dtype, int M, hipStream_t st, bool* handled) {
  *handled = false;
  if (!wae_is16(dtype) || a.nsrc != 1 || a.src_shift[0] != 0 || a.mode != TM_BIAS_RELU || M != 256) return WAE_OK;
  if ((a.flags & WAE_TM_ONE_WG) || a.stamps) return WAE_OK;               // A/B switch and diagnostic builds: the generic kernel
  const int nq = a.src_cols[0] / 64;
  if (nq < 4 || nq % 4 != 0) return WAE_OK;                                // bodies of 8 half-chunks
  if ((int64_t)a.T * a.src_stride[0] * 2 >= (int64_t)1 << 31) return WAE_OK;   // 32-bit offsets inside a clip's descriptor
  *handled = true;
  if (dtype == WAE_BF16) return launch_tm8<__bf16, 8, TM_BIAS_RELU>(a, st);
  return launch_tm8<f16, 8, TM_BIAS_RELU>(a, st);
}
