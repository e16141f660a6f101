// gemm_tm8: the time-major GEMM of csrc/gemm_tm.hip for ONE source whose K dimension is long and whose operand streams from HBM --
// the head's skip contraction  h0 = relu(sqrt(1/L) (sum_l b_skip_l + [W_skip_0 .. W_skip_{L-1}] [u_0; ..; u_{L-1}]))
// (wavenet.py:204-209; K = L * Hp = 4608 at C2: 590 MB of u per launch).
//
// gemm_tm_kernel runs it on 4 waves x 128 columns with a two-slot weight ring and MFMA-operand-shaped requests (32 rows x 32 bytes per
// wave instruction) issued by the computing waves: 196 us at C2, 3.2 TB/s.  What round 6 measured on the way to this kernel (timing-only
// ablations, profiles/r06_tm8_experiments.txt): the compute side alone (MFMAs + A-fragment reads) takes 83 us, the weight re-reads
// alone 31 us (19 TB/s out of L2), the operand requests alone 132 us in fragment shape and 117 us as whole row pieces -- and the three do
// not overlap while the waves that request are the waves that compute (an LDS-DMA piece costs its wave 60-185 clocks of issue time
// among MFMAs, and a wave issues in order).  gemm_tm8s_kernel therefore:
//   * 8 CONSUMER waves x 32 time columns (ds_read + MFMA, nothing else) + 4 LOADER waves (requests, nothing else), one workgroup per
//     CU: the packed weights enter the CU once per 256 columns (gemm_tm: per 128); three waves per SIMD, so 168 registers;
//   * BOTH operands travel by LDS-DMA (`buffer_load_dwordx4 ... lds`), 8 one-KiB pieces per loader and K = 32 half-chunk: loaders 8-9
//     the weights (4-slot ring, 3 half-chunks ahead), loaders 10-11 the activation rows (16 rows x 64 B per piece through the clip's
//     buffer descriptor -- a row past the clip's end lands as zeros; 4 tiles, 4 half-chunks ahead).  One stream per loader: vmcnt
//     retires in order, and a wave that mixed the two would hold the deep operand stream back to the depth of the shallow weight stream;
//   * operand tile: 32 rows x 64 B per consumer wave, its 16-byte columns XOR-ed with (row >> 2) & 3 on the GLOBAL side (the LDS side
//     of an LDS-DMA is linear): the fragment reads (row n, column 2 f + h) are conflict-free; the fragments of half-chunk c + 1 are read
//     into registers during half-chunk c, the A fragments run PD = 4 reads ahead of the MFMAs under counted lgkmcnt waits;
//   * one workgroup barrier per half-chunk: a loader passes it once its own pieces through X(c + 1) have landed (counted vmcnt), a
//     consumer once it has left half-chunk c - 1; behind it the loaders refill the slot / tile that half-chunk c - 1 / c occupied.
// Same packed weight stream, same fragment layouts and the same accumulation order as gemm_tm_kernel: results are bit-identical
// (tests/test_gpu_tm8.py, 12 launches per case).  C2: 170 us (-14 %), hps/vqwae.json (K = 2560): 57 us against 86 (-34 %).
#include "gemm_tm.hpp"

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void static_for8(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    static_for8<I + 1, N>(f);
  }
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

template <typename E, int NT, int MODE, int PD>
__global__ void __launch_bounds__(768, 1) gemm_tm8s_kernel(TmArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  static_assert(sizeof(E) == 2 && T_::CK == 64 && MODE == TM_BIAS_RELU && NT == 8, "16-bit storage, M = 256, mode 3");
  constexpr int NW = 8, KB = 2, NM = NT, ES = 2;
  constexpr int HCB = NM * KB * 1024;
  constexpr int NSW = 4, DW = 3, NTB = 4, DB = 4;
  constexpr int TILEB = NW * 2048;
  constexpr int BODY = 4, NSTEP = KB * NM, NG = BODY * NSTEP;
  constexpr int ALLOW_W = (DW - 2) * 8, ALLOW_B = (DB - 2) * 8;
  constexpr int RING = NSW * HCB, TILES = NTB * TILEB;
  static_assert(RING + TILES + 1024 <= 160 * 1024 && NW * STG_BYTES <= RING && 2 + PD < NSTEP, "LDS budget; the ring doubles as staging area");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int TW = NW * 32;
  const int tiles_per_b = (p.T + TW - 1) / TW;
  // (the slices of one time tile sit next to each other in the XCD-contiguous launch order: they run on one XCD at the same time, and
  //  the second one's operand requests are hits in that XCD's L2.  As blockIdx.y the slices ran a whole grid apart: C5's skip
  //  contraction -- 2 GB of u, two slices -- 1 112 us; adjacent 943 us, tools/time_tm8.py)
  const int lin_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int tile_id = lin_id / p.nslices;
  const int b = tile_id / tiles_per_b;
  const int t0 = (tile_id % tiles_per_b) * TW;
  const int nh = 2 * (p.src_cols[0] / T_::CK);
  const int nit = nh / BODY;
  const unsigned row_bytes = (unsigned)(p.src_stride[0] * ES);
  const unsigned clip_bytes = (unsigned)p.T * row_bytes;
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  float* bias_lds = (float*)(smem + RING + TILES);
  // outputs wider than 256 rows (the wide head, BASELINE config C5): a slice of 8 tiles per workgroup; the weight stream is
  // [slice][chunk] (packing.py: first_gemm_map), bias and output columns follow
  const int slice = lin_id % p.nslices;
  const char* wslice = p.w + (int64_t)slice * nh * HCB;
  const int64_t col0 = (int64_t)slice * NT * 32;

  if (wave >= NW) {
    // ================= loader waves: 8, 9 the weights (KiB 8 lq .. +8 of every half-chunk), 10, 11 the operand (consumer waves 4 lq .. +4)
    const int lw = wave - NW, lq = lw & 1;
    const bool wl = lw < 2;
    const i32x4 srd_b = make_srd8(p.src[0] + (int64_t)b * clip_bytes, clip_bytes);
    const i32x4 srd_w = make_srd8(wslice, (unsigned)min((int64_t)nh * HCB, (int64_t)0x7fffffff));
    i32x4 srd;
    srd.x = wl ? srd_w.x : srd_b.x; srd.y = wl ? srd_w.y : srd_b.y; srd.z = wl ? srd_w.z : srd_b.z; srd.w = srd_w.w;
    unsigned voff[8], lds_piece[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int cw = 4 * lq + (k >> 1), r0 = 16 * (k & 1);
      const unsigned vb = (unsigned)(t0 + 32 * cw + r0 + (lane >> 2)) * row_bytes + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);
      const unsigned vw = (unsigned)((8 * lq + k) * 1024 + lane * 16);
      voff[k] = wl ? vw : vb;
      lds_piece[k] = wl ? (unsigned)((8 * lq + k) * 1024) : (unsigned)(cw * 2048 + r0 * 64);
    }
    const unsigned role_base = lds0 + (wl ? 0u : (unsigned)RING);
    const unsigned role_step = wl ? (unsigned)HCB : (unsigned)TILEB;
    const unsigned role_cols = wl ? (unsigned)HCB : 64u;
    const int role_d = wl ? DW : DB;
    auto piece = [&](int cc, auto jwc, auto jbc, auto kc) {
      constexpr int jw = decltype(jwc)::value, jb = decltype(jbc)::value, k = decltype(kc)::value;
      const unsigned soff = (unsigned)min(cc, nh - 1) * role_cols;
      const unsigned m0v = role_base + (wl ? (unsigned)jw : (unsigned)jb) * role_step + lds_piece[k];
      const unsigned vk = voff[k];
      const i32x4 sr = srd;
      asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(vk), "s"(sr), "s"(soff) : "m0");
    };
    f32x4 tb = {};
    const unsigned tb_off = lane * 16;
    const char* bias_g = p.aux + col0 * 4;
    if (lw == 0) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(tb) : "v"(tb_off), "s"(bias_g));
    static_for8<0, DB>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      if (c < role_d) static_for8<0, 8>([&](auto kc) { piece(c, IntC<c % NSW>{}, IntC<c % NTB>{}, kc); });
    });
    if (lw == 0) {
      const unsigned tw = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)bias_lds + lane * 16;
      asm volatile("s_waitcnt vmcnt(%1)" : "+v"(tb) : "n"(DW * 8));
      asm volatile("ds_write_b128 %0, %1" ::"v"(tw), "v"(tb));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // half-chunk "-1": X(0) has landed -- the consumers read the first operand fragments (and the bias table) between this barrier and
    // the next, BEFORE any loader may overwrite tile 0 with the operand of half-chunk DB
    if (wl) wait_vm8<(DW - 1) * 8>(); else wait_vm8<(DB - 1) * 8>();
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < nit; ++it) {
      const int c0 = it * BODY;
      static_for8<0, BODY>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if (wl) wait_vm8<ALLOW_W>(); else wait_vm8<ALLOW_B>();       // this wave's pieces through X(c + 1) have landed
        __builtin_amdgcn_s_barrier();                                 // ... every loader's; every consumer has left half-chunk c - 1
        static_for8<0, 8>([&](auto kc) { piece(c0 + j + role_d, IntC<(j + DW) % NSW>{}, IntC<(j + DB) % NTB>{}, kc); });
      });
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return;
  }

  // ================= consumer waves ===========================================================================================================
  const int n = lane & 31, h = lane >> 5;
  const int t0w = t0 + wave * 32;
  const int rows_valid = min(max(p.T - t0w, 0), 32);
  unsigned a_base = lds0 + lane * 16;
  auto a_read = [&](auto jc, auto ic, frag& dst) {
    constexpr int j = decltype(jc)::value, I = decltype(ic)::value;
    lds_read_off<(j % NSW) * HCB + I * 1024>(dst, a_base);
  };
  unsigned b_addr[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) b_addr[f] = lds0 + RING + wave * 2048 + n * 64 + (((2 * f + h) ^ ((n >> 2) & 3)) << 4);
  auto b_read = [&](auto tlc, frag (&dst)[2]) {
    constexpr int tl = decltype(tlc)::value;
    lds_read_off<tl * TILEB>(dst[0], b_addr[0]);
    lds_read_off<tl * TILEB>(dst[1], b_addr[1]);
  };
  f32x16 acc[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
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
  frag Bf[2][2];
  __builtin_amdgcn_s_barrier();          // half-chunk "-1" (the loaders' comment): X(0) and the bias table are visible
  static_for8<0, NM>(init_tile);
  b_read(IntC<0>{}, Bf[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bf[0][0]), "+v"(Bf[0][1]));
  for (int it = 0; it < nit; ++it) {
    frag a[PD];
    static_for8<0, BODY>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      __builtin_amdgcn_s_barrier();
      if constexpr (j == 0) {
        __builtin_amdgcn_sched_barrier(0);
        static_for8<0, PD>([&](auto ic) { a_read(IntC<0>{}, ic, a[decltype(ic)::value]); });
      }
      asm volatile("" : "+v"(Bf[j % 2][0]), "+v"(Bf[j % 2][1]));
      static_for8<0, NSTEP>([&](auto ic) {
        constexpr int I = decltype(ic)::value, G = j * NSTEP + I, AI = G % PD;
        constexpr int remaining = NG - 1 - G;
        constexpr int younger_a = remaining < PD - 1 ? remaining : PD - 1;
        constexpr int extra = (I >= 2 && I <= 1 + PD) ? 2 : 0;          // (the two operand-fragment reads of step 1 go out right after A(G_1 + PD): younger than every A read up to that one, i.e. outstanding at the waits of steps 2 .. 1 + PD)
        lds_wait<younger_a + extra>(a[AI]);
        mma32(acc[I % NM], a[AI], Bf[j % 2][I / NM]);
        if constexpr (remaining >= PD) {
          constexpr int G2 = G + PD;
          a_read(IntC<G2 / NSTEP>{}, IntC<G2 % NSTEP>{}, a[AI]);
        }
        if constexpr (I == 1) b_read(IntC<(j + 1) % NTB>{}, Bf[(j + 1) % 2]);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  __syncthreads();      // every wave is done with the weight ring (the loaders have drained their requests): it becomes the staging area
  if (rows_valid <= 0) return;
  char* stg = smem + wave * STG_BYTES;
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = fmaxf(acc[m][r] * p.alpha, 0.f);
  char* orow = p.out + (((int64_t)b * p.T + t0w) * p.out_stride + col0) * ES;
  stage_store_tiles<E, NT>(stg, acc, orow, p.out_stride * ES, rows_valid, lane);
}

namespace {

// what a loader wave of gemm_tm8x_kernel needs to place piece k of X(cbase + J) (by value: a capturing lambda of this size ended up
// with its captures in scratch memory in one instantiation, and a reload with s_waitcnt vmcnt(0) in front of every request)
struct Tm8xLoader {
  unsigned lds0, rb0, rb1, swz, lane16;
  int rowl, wq, nsrc, nq0, nh;
  bool wl;
  const char* wslice;
  i32x4 srd0, srd1;
};

// piece K of X(c), c = cbase + J0 with cbase a multiple of 4 and J0 a compile-time constant: ring slot, tile and k-block half of c follow
// from J0 (the bodies are 4 half-chunks long)
template <int NT, int MODE, int J0, int K>
__device__ __forceinline__ void tm8x_issue(const Tm8xLoader& L, int cbase, int sh0, int sh1, int sh2) {
  constexpr int J = J0;
  // (the row shifts by value: as fields of L the select below becomes an indexed load, and the whole of L stays in scratch memory)
  constexpr int NM = NT, HCB = NM * 2 * 1024, NSW = 4, NTB = 4, TILEB = 8 * 2048, RING = NSW * HCB, PPW = NM / 2;
  constexpr int slot = J % NSW, tile = J % NTB, hk = J % 2;
  const int c = cbase + J;
  if (c >= L.nh) return;
  if (L.wl) {
    if constexpr (K < PPW) {
      const char* src = L.wslice + ((int64_t)c * HCB + (L.wq * PPW + K) * 1024) + L.lane16;
      const unsigned dst = L.lds0 + slot * HCB + (L.wq * PPW + K) * 1024;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
    }
  } else {
    // (wave-uniform values that hipcc may keep in vector registers: the asm's scalar operands are made scalar explicitly)
    auto sgpr = [](unsigned v) __attribute__((always_inline)) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    const unsigned m0v = sgpr(L.lds0 + RING + tile * TILEB + (2 * L.wq + (K >> 1)) * 2048 + (K & 1) * 1024);
    const int q = c >> 1;
    bool second = false;
    int cb, shift;
    if constexpr (MODE == TM_RESIDUAL) {     // chunk q: tap q % nsrc of column block q / nsrc (TM_INTERLEAVE), every tap the same array
      cb = q / L.nsrc;
      const int tap = q - cb * L.nsrc;
      shift = tap == 0 ? sh0 : (tap == 1 ? sh1 : sh2);
    } else {                                 // the sources one after the other
      second = q >= L.nq0;
      cb = second ? q - L.nq0 : q;
      shift = second ? sh1 : sh0;
    }
    const unsigned vo = (unsigned)(L.rowl + 16 * K + shift) * (second ? L.rb1 : L.rb0) + L.swz;
    const unsigned so = sgpr((unsigned)(cb * 128 + hk * 64));
    i32x4 sr;
    sr.x = (int)sgpr((unsigned)(second ? L.srd1.x : L.srd0.x)); sr.y = (int)sgpr((unsigned)(second ? L.srd1.y : L.srd0.y));
    sr.z = (int)sgpr((unsigned)(second ? L.srd1.z : L.srd0.z)); sr.w = (int)sgpr((unsigned)L.srd0.w);
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(vo), "s"(sr), "s"(so) : "m0");
  }
}

}  // namespace

// ---- the backward sweep's two launches on the same machinery (round 6) --------------------------------------------------------------------
// TM_RESIDUAL with interleaved taps (dx-hat of a layer: K = taps x 2Hp over shifted rows of dz, output rows in slices of 256) and
// TM_GATE_BWD (du -> dz of a layer: K = Rp + Sp over dx-hat and dskip, Hp <= 256 output rows, gate derivative from the saved z): what the
// 16-bit sweep runs where csrc/glu_bwd8.hip has no instantiation (Rp = 512: BASELINE C5), on its top and bottom layer, and with
// WAE_BWD_FUSED=0.  The registers do not allow loader waves of their own here (eight accumulator tiles + the epilogue's rows), so the
// eight waves share the requests as in glu_bwd_pair8_kernel: waves 0-3 the weight pieces (4-slot ring of K = 32 half-chunks, three
// ahead), waves 4-7 the operand pieces (16 rows x 64 bytes through per-clip descriptors into four swizzled tiles, four ahead), issued
// between their MFMAs; one workgroup barrier per half-chunk.  Same packed weight streams, fragment layouts, accumulation order and
// epilogue arithmetic as gemm_tm_kernel: bit-identical results (tests/test_gpu_tm8.py).
template <typename E, int NT, int MODE>
__global__ void __launch_bounds__(512, 1) gemm_tm8x_kernel(TmArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  static_assert(sizeof(E) == 2 && T_::CK == 64 && (MODE == TM_RESIDUAL || MODE == TM_GATE_BWD) && NT % 2 == 0 && NT <= 8, "16-bit storage");
  constexpr int NW = 8, NM = NT, ES = 2, PD = 4;
  constexpr int HCB = NM * 2 * 1024;
  constexpr int NSW = 4, DW = 3, NTB = 4, DB = 4;
  constexpr int TILEB = NW * 2048, RING = NSW * HCB, TILES = NTB * TILEB;
  constexpr int BODY = 4, NSTEP = 2 * NM, NG = BODY * NSTEP, PPW = NM / 2, SP = NSTEP / 4;
  constexpr int STGB = 4096;
  static_assert(RING + TILES <= 160 * 1024 && NW * STGB <= TILES && PPW <= 4 && 4 * SP <= NSTEP && 2 + PD < NSTEP, "LDS budget; four issue points per half-chunk");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  constexpr int TW = NW * 32;
  const int tiles_per_b = (p.T + TW - 1) / TW;
  const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int b = tile_id / tiles_per_b;
  const int t0 = (tile_id % tiles_per_b) * TW;
  const int t0w = t0 + wave * 32;
  const int rows_valid = min(max(p.T - t0w, 0), 32);
  const int slice = blockIdx.y, col0 = slice * NT * 32;
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  const int nsrc = p.nsrc;
  const int nq0 = p.src_cols[0] / 64;
  const int nq = MODE == TM_RESIDUAL ? nsrc * nq0 : nq0 + (nsrc > 1 ? p.src_cols[1] : 0) / 64;
  const int nh = 2 * nq, nit = nh / BODY;
  const char* wslice = p.w + (int64_t)slice * nh * HCB;

  // ---- loader roles ------------------------------------------------------------------------------------------------------------------
  const bool wl = wave < 4;
  const int wq = wave & 3;
  const unsigned lane16 = lane * 16;
  // operand: this loader's piece k = rows 16 (k & 1) .. +16 of consumer wave 2 wq + (k >> 1); lane -> row (lane >> 2), 16-byte column
  // (lane & 3) ^ ((lane >> 4) & 3)   [= col ^ ((row >> 2) & 3): the tile's swizzle, applied on the global side]
  const int rowl = t0 + 64 * wq + (lane >> 2);
  const unsigned swz = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
  // (no run-time index into the kernel argument: that would put a copy of it into scratch memory)
  const bool two = MODE == TM_GATE_BWD && nsrc > 1;
  const char* src1 = two ? p.src[1] : p.src[0];
  const int64_t stride1 = two ? p.src_stride[1] : p.src_stride[0];
  const int cols1 = two ? p.src_cols[1] : p.src_cols[0];
  const unsigned rb0 = (unsigned)(p.src_stride[0] * ES), rb1 = (unsigned)(stride1 * ES);
  // (a source may be a column slice of a wider array: the clip's descriptor ends with the last row's own columns)
  const i32x4 srd0 = make_srd8(p.src[0] + (int64_t)b * p.T * rb0, (unsigned)p.T * rb0 - (unsigned)(p.src_stride[0] - p.src_cols[0]) * ES);
  const i32x4 srd1 = make_srd8(src1 + (int64_t)b * p.T * rb1, (unsigned)p.T * rb1 - (unsigned)(stride1 - cols1) * ES);
  const int sh0 = p.src_shift[0], sh1 = p.src_shift[1], sh2 = p.src_shift[2];
  const Tm8xLoader ld{lds0, rb0, rb1, swz, lane16, rowl, wq, nsrc, nq0, nh, wl, wslice, srd0, srd1};
  auto issue = [&](int cbase, auto jc, auto kc) __attribute__((always_inline)) {      // cbase: a multiple of 4
    tm8x_issue<NT, MODE, decltype(jc)::value, decltype(kc)::value>(ld, cbase, sh0, sh1, sh2);
  };
  auto issue_all = [&](int cbase, auto jc) __attribute__((always_inline)) { static_for8<0, 4>([&](auto kc) { issue(cbase, jc, kc); }); };
  // a loader's counted wait at the top of half-chunk c: everything through X(c + 1) has landed, X(c + 2) .. X(c + D - 1) may be in flight.
  // At the end of the stream issue() drops X(>= nh): the same count would then leave pieces of X(c + 1) in flight -- the last D - 1
  // half-chunks drain instead (tail).
  auto top = [&](bool tail) __attribute__((always_inline)) {
    if (tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (wl) wait_vm8<(DW - 2) * PPW>();
    else wait_vm8<(DB - 2) * 4>();
    __builtin_amdgcn_s_barrier();
  };

  // ---- consumer side -----------------------------------------------------------------------------------------------------------------
  unsigned a_base = lds0 + lane16;
  auto a_read = [&](auto jc, auto ic, frag& dst) __attribute__((always_inline)) {
    constexpr int j = decltype(jc)::value, I = decltype(ic)::value;
    lds_read_off<(j % NSW) * HCB + I * 1024>(dst, a_base);
  };
  unsigned b_addr[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) b_addr[f] = lds0 + RING + wave * 2048 + n * 64 + (((2 * f + h) ^ ((n >> 2) & 3)) << 4);
  auto b_read = [&](auto tlc, frag (&dst)[2]) __attribute__((always_inline)) {
    constexpr int tl = decltype(tlc)::value;
    lds_read_off<tl * TILEB>(dst[0], b_addr[0]);
    lds_read_off<tl * TILEB>(dst[1], b_addr[1]);
  };

  // ---- prologue: X(0) .. X(D - 1); half-chunk "-1": X(0) visible, the first operand fragments into registers --------------------------
  static_for8<0, DB>([&](auto jc) {
    constexpr int j = decltype(jc)::value;
    if constexpr (j < DW) { if (wl) issue_all(0, jc); }
    if (!wl) issue_all(0, jc);
  });
  if (wl) wait_vm8<(DW - 1) * PPW>(); else wait_vm8<(DB - 1) * 4>();
  __builtin_amdgcn_s_barrier();
  frag Bf[2][2];
  b_read(IntC<0>{}, Bf[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bf[0][0]), "+v"(Bf[0][1]));

  f32x16 acc[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  for (int it = 0; it < nit; ++it) {
    const int c0 = it * BODY;
    frag a[PD];
    static_for8<0, BODY>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      top(c0 + j + DB > nh);
      asm volatile("" : "+v"(Bf[j % 2][0]), "+v"(Bf[j % 2][1]));
      if constexpr (j == 0) {
        __builtin_amdgcn_sched_barrier(0);
        static_for8<0, PD>([&](auto ic) { a_read(IntC<0>{}, ic, a[decltype(ic)::value]); });
      }
      static_for8<0, NSTEP>([&](auto ic) {
        constexpr int I = decltype(ic)::value, G = j * NSTEP + I, AI = G % PD;
        constexpr int remaining = NG - 1 - G;
        constexpr int younger_a = remaining < PD - 1 ? remaining : PD - 1;
        constexpr int extra = (I >= 2 && I <= 1 + PD) ? 2 : 0;          // the operand-fragment reads of step 1
        lds_wait<younger_a + extra>(a[AI]);
        mma32(acc[I % NM], a[AI], Bf[j % 2][I / NM]);
        if constexpr (remaining >= PD) {
          constexpr int G2 = G + PD;
          a_read(IntC<G2 / NSTEP>{}, IntC<G2 % NSTEP>{}, a[AI]);
        }
        if constexpr (I == 1) b_read(IntC<(j + 1) % NTB>{}, Bf[(j + 1) % 2]);
        if constexpr (I % SP == 0 && I / SP < 4) {
          if (wl) issue(c0, IntC<j + DW>{}, IntC<I / SP>{}); else issue(c0, IntC<j + DB>{}, IntC<I / SP>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();      // every wave is done with ring and tiles: the ring becomes the staging area
  if (rows_valid <= 0) return;
  int le = lane;
  asm volatile("" : "+v"(le));
  char* stg = smem + wave * STGB;
  f32x4 fa[8], fb[8];
  if constexpr (MODE == TM_RESIDUAL) {
    // out = alpha * (acc + res[t])   (csrc/gemm_tm.hip: the paired residual epilogue)
    const char* arow = p.aux + (((int64_t)b * p.T + t0w) * p.aux_stride + col0) * ES;
    char* orow = p.out + (((int64_t)b * p.T + t0w) * p.out_stride + col0) * ES;
    stage_fetch_pass<E, 2>(fa, arow, p.aux_stride * ES, rows_valid, le);
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
      f32x16 res[2];
      stage_unpack_pass<E, 2, 128>(stg, res, fa, le);
      if (pr + 1 < NT / 2) stage_fetch_pass<E, 2>(fa, arow + (pr + 1) * 64 * ES, p.aux_stride * ES, rows_valid, le);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[2 * pr + i][r] = p.alpha * (acc[2 * pr + i][r] + res[i][r]);
      stage_store_pass<E, 2, 128>(stg, &acc[2 * pr], orow + pr * 64 * ES, p.out_stride * ES, rows_valid, le);
    }
  } else {
    // gate backward (modules.py:154: u = tanh(a) * sigmoid(b)):  da = du * s * (1 - th^2),  db = du * th * s * (1 - s)
    const char* zrow = p.aux + ((int64_t)b * p.T + t0w) * p.aux_stride * ES;
    char* orow = p.out + ((int64_t)b * p.T + t0w) * p.out_stride * ES;
    stage_fetch_pass<E, 2>(fa, zrow, p.aux_stride * ES, rows_valid, le);
    stage_fetch_pass<E, 2>(fb, zrow + (int64_t)NT * 32 * ES, p.aux_stride * ES, rows_valid, le);
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
      f32x16 za[2], zg[2];
      stage_unpack_pass<E, 2, 128>(stg, za, fa, le);
      stage_unpack_pass<E, 2, 128>(stg, zg, fb, le);
      if (pr + 1 < NT / 2) {
        stage_fetch_pass<E, 2>(fa, zrow + (pr + 1) * 64 * ES, p.aux_stride * ES, rows_valid, le);
        stage_fetch_pass<E, 2>(fb, zrow + ((int64_t)NT * 32 + (pr + 1) * 64) * ES, p.aux_stride * ES, rows_valid, le);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ea = __builtin_amdgcn_exp2f(fmaxf(za[i][r], -15.0f) * -2.885390081777927f);
          const float th = (1.0f - ea) * fast_rcp(1.0f + ea);
          const float sg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(zg[i][r] * -1.4426950408889634f));
          const float du = acc[2 * pr + i][r];
          za[i][r] = du * sg * (1.0f - th * th);
          zg[i][r] = du * th * sg * (1.0f - sg);
        }
      stage_store_pass<E, 2, 128>(stg, za, orow + pr * 64 * ES, p.out_stride * ES, rows_valid, le);
      stage_store_pass<E, 2, 128>(stg, zg, orow + ((int64_t)NT * 32 + pr * 64) * ES, p.out_stride * ES, rows_valid, le);
    }
  }
}

namespace {

template <typename E, int NT, int MODE>
int launch_tm8s(const TmArgs& a, int nslices, hipStream_t st) {
  auto kern = gemm_tm8s_kernel<E, NT, MODE, 4>;
  const size_t lds = (size_t)4 * NT * 2048 + (size_t)4 * 8 * 2048 + 1024;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)kern, lds_cache, lds, "gemm_tm8s"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 255) / 256;
  TmArgs as = a;
  as.nslices = nslices;
  hipLaunchKernelGGL(kern, dim3(a.B * tiles * nslices), dim3(768), lds, st, as);
  return wae_check_launch("gemm_tm8s");
}

}  // namespace

namespace {

template <typename E, int NT, int MODE>
int launch_tm8x(const TmArgs& a, int nslices, hipStream_t st) {
  auto kern = gemm_tm8x_kernel<E, NT, MODE>;
  const size_t lds = (size_t)4 * NT * 2048 + (size_t)4 * 8 * 2048;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)kern, lds_cache, lds, "gemm_tm8x"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(a.B * tiles, nslices), dim3(512), lds, st, a);
  return wae_check_launch("gemm_tm8x");
}

template <typename E>
int dispatch_tm8x(const TmArgs& a, int M, hipStream_t st, bool* handled) {
  *handled = true;
  if (a.mode == TM_RESIDUAL) return launch_tm8x<E, 8, TM_RESIDUAL>(a, M / 256, st);
  switch (M) {
    case 256: return launch_tm8x<E, 8, TM_GATE_BWD>(a, 1, st);
    case 192: return launch_tm8x<E, 6, TM_GATE_BWD>(a, 1, st);
    case 128: return launch_tm8x<E, 4, TM_GATE_BWD>(a, 1, st);
  }
  *handled = false;
  return WAE_OK;
}

// the backward sweep's residual (interleaved taps of ONE array) and gate launches: shapes gemm_tm8x_kernel covers
bool tm8x_covers(const TmArgs& a, int M) {
  if (!a.aux || a.nsrc < 1) return false;
  int nq = 0;
  if (a.mode == TM_RESIDUAL) {
    if (!a.interleave || a.nsrc > 3 || M % 256 != 0) return false;
    for (int s = 1; s < a.nsrc; ++s)
      if (a.src[s] != a.src[0] || a.src_stride[s] != a.src_stride[0] || a.src_cols[s] != a.src_cols[0]) return false;
    nq = a.nsrc * (a.src_cols[0] / 64);
  } else if (a.mode == TM_GATE_BWD) {
    if (a.interleave || a.nsrc > 2 || (M != 256 && M != 192 && M != 128)) return false;
    for (int s = 0; s < a.nsrc; ++s) nq += a.src_cols[s] / 64;
  } else {
    return false;
  }
  if (nq < 4 || nq % 2 != 0) return false;                                  // bodies of 4 half-chunks
  for (int s = 0; s < a.nsrc; ++s)
    if ((int64_t)a.T * a.src_stride[s] * 2 >= (int64_t)1 << 31) return false;   // 32-bit offsets inside a clip's descriptor
  return true;
}

}  // namespace

int wae_gemm_tm8_launch(const TmArgs& a, int dtype, int M, hipStream_t st, bool* handled) {
  *handled = false;
  if (!wae_is16(dtype) || M <= 0) return WAE_OK;
  if ((a.flags & WAE_TM_ONE_WG) || a.stamps) return WAE_OK;               // A/B switch and diagnostic builds: the generic kernel
  if (tm8x_covers(a, M)) return dtype == WAE_BF16 ? dispatch_tm8x<__bf16>(a, M, st, handled) : dispatch_tm8x<f16>(a, M, st, handled);
  if (a.nsrc != 1 || a.src_shift[0] != 0 || a.mode != TM_BIAS_RELU || M % 256 != 0) return WAE_OK;
  const int nq = a.src_cols[0] / 64;
  if (nq < 4 || nq % 2 != 0) return WAE_OK;                                // bodies of 4 half-chunks
  if ((int64_t)a.T * a.src_stride[0] * 2 >= (int64_t)1 << 31) return WAE_OK;   // 32-bit offsets inside a clip's descriptor
  *handled = true;
  if (dtype == WAE_BF16) return launch_tm8s<__bf16, 8, TM_BIAS_RELU>(a, M / 256, st);
  return launch_tm8s<f16, 8, TM_BIAS_RELU>(a, M / 256, st);
}
