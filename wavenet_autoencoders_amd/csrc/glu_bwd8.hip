// glu_bwd8: the backward pair launch of csrc/glu_bwd.hip -- residual(l) + gate(l-1) + this layer's conditioning gradient (autograd of
// modules.py:115-163) -- on 8 waves x 256 time columns, ONE workgroup per CU, with BOTH operand streams through LDS (round 6).
//
//   dx_l-hat[t]  = sqrt(.5) * ( dx_{l+1}-hat[t] + sum_tap W1_l,tap^T dz_l[t + (2-tap) d_l] )      phase A   (+ dc += Wc_l^T dz_l[t])
//   du_{l-1}[t]  = W_out_{l-1}^T dx_l-hat[t]  (phase B1, operand from registers)  +  W_skip_{l-1}^T dskip[t]  (phase B2)
//   dz_{l-1}[t]  = gate'(z_{l-1}[t]) * du_{l-1}[t]                                                  epilogue B
//
// What the round-5 kernel (glu_bwd_pair_kernel: 4 waves x 128 columns, two workgroups per CU, a two-slot weight ring, MFMA-operand-shaped
// requests two chunks ahead) pays for, measured on timing-only builds with the SAME data (tools/time_pair.py, profiles/
// r06_tm8_experiments.txt): 85-86 us per launch; without its operand requests 62-64; with half / none of its weight DMA 77-81 / 71-73.
// Here, as in csrc/gemm_tm8.hip:
//   * the packed weights enter the CU once per 256 columns (half the L2 -> LDS bytes per column), through a 4-slot ring of K = 32
//     half-chunks (16 KiB + the 4-KiB block of the conditioning weights behind the shift-0 tap's), three half-chunks ahead;
//   * the activation operands (dz_l at the three tap shifts, dS) travel as whole 64-byte row pieces by LDS-DMA through per-clip buffer
//     descriptors -- a row past the clip's end lands as zeros -- into swizzled tiles, four half-chunks ahead; a fragment-shaped request
//     (32 rows x 32 bytes) costs the texture-address path four times as many accesses per byte;
//   * waves 0-3 issue the weight pieces, waves 4-7 the operand pieces (vmcnt retires in order: one stream per wave lets each run at
//     its own depth); all eight compute.  The whole sequence -- 36 + 6 + 8 half-chunks at C2 -- is unrolled: ring slot, tile, every
//     counted wait and every LDS offset are immediates; the waits come out of a constexpr replay of the issue order (Plan).
// Same packed streams, same fragment layouts, same MFMA order per accumulator as glu_bwd_pair_kernel: results are BITWISE its results
// (tests/test_gpu_backward.py::test_pair8_launch_is_bitwise_the_pair_launch).
#include "glu_bwd.hpp"

// timing-only ablations (tools/time_pair.py on variant builds; results are wrong): 1 no operand pieces, 2 no weight pieces, 4 no MFMAs
#ifndef WAE_GB8_ABL
#define WAE_GB8_ABL 0
#endif
#define GB8_MMA(acc, a, b) do { if constexpr (!(WAE_GB8_ABL & 4)) mma32(acc, a, b); } while (0)

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    sfor<I + 1, N>(f);
  }
}
template <int CNT>
__device__ __forceinline__ void wait_vmc() {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}
template <int OFF, typename frag>
__device__ __forceinline__ void lds_rd(frag& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
__device__ __forceinline__ i32x4 srd_of(const char* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

// ---- the plan: everything about half-chunk c that is known at compile time ------------------------------------------------------------
// NTU = Hp / 32 gate tiles (6 at C2, 4 at hps/vqwae.json); SPQ = Sp / 64 chunks of dS; three taps; Rp = 256 (8 output tiles in phase A).
template <int NTU, int SPQ>
struct Plan {
  static constexpr int DW = 3, DB = 4, NSW = 4, NTB = 4;
  static constexpr int CPT = NTU;                         // 64-column blocks per tap (2 Hp / 64)
  static constexpr int HA = 2 * 3 * CPT, HB1 = NTU, HB2 = 2 * SPQ, NH = HA + HB1 + HB2;
  static constexpr int phase(int c) { return c < HA ? 0 : (c < HA + HB1 ? 1 : 2); }
  static constexpr int tap(int c) { return (c / 2) % 3; }
  static constexpr int cb(int c) { return (c / 2) / 3; }
  static constexpr bool fold(int c) { return c >= 0 && c < HA && tap(c) == 2; }
  static constexpr int nsteps(int c) { return phase(c) == 0 ? 16 + (fold(c) ? 4 : 0) : (phase(c) == 1 ? 16 : 2 * NTU); }
  // pieces per loader wave
  static constexpr int nW(int c) { return c < 0 || c >= NH ? 0 : (phase(c) == 0 ? 4 + (fold(c) ? 1 : 0) : (phase(c) == 1 ? 4 : NTU / 2)); }
  static constexpr int nB(int c) { return c < 0 || c >= NH ? 0 : (phase(c) == 1 ? 0 : 4); }
  // a loader's counted wait at the top of half-chunk c: everything through X(c + 1) has landed; X(c + 2) .. X(c + D - 1) may be in flight
  static constexpr int allow_w(int c) { int n = 0; for (int i = c + 2; i <= c + DW - 1; ++i) n += nW(i); return n; }
  static constexpr int allow_b(int c) { int n = 0; for (int i = c + 2; i <= c + DB - 1; ++i) n += nB(i); return n; }
};

}  // namespace

template <typename E, int NTU, int SPQ>
__global__ void __launch_bounds__(512, 1) glu_bwd_pair8_kernel(GbArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  using PL = Plan<NTU, SPQ>;
  static_assert(sizeof(E) == 2 && NTU % 2 == 0, "16-bit storage, pairwise gate epilogue");
  constexpr int ES = 2, NTX = 8, KBU = 2, NKB = NTX * KBU, NW = 8, PD = 4;
  constexpr int DW = PL::DW, DB = PL::DB, NSW = PL::NSW, NTB = PL::NTB;
  constexpr int HA = PL::HA, HB1 = PL::HB1, NH = PL::NH;
  constexpr int SLOT = 16384 + 4096;             // a half-chunk of phase-A weights + the dc block behind the shift-0 tap's
  constexpr int RING = NSW * SLOT, TILEB = NW * 2048, TILES = NTB * TILEB;
  constexpr int Z2 = 2 * NTU * 32;
  constexpr int STGB = 4096;
  static_assert(RING + TILES <= 160 * 1024 && NW * STGB <= TILES && 2 * SLOT + 20480 <= 65536, "LDS budget; staging sits in the tile area");
  static_assert(HB1 * 16384 == NTU * NKB * 1024, "a B1 half-chunk is one output tile x all k-blocks");

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
  const int64_t row0 = (int64_t)b * p.T + t0w;
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  // (Measured and not kept, profiles/r06_pair8_experiments.txt: every second tile started 5 / 10 / 15 / 25 us late -- so that its HBM-only
  //  epilogues meet the other half's GEMM phases -- ends the launch 0 / 2 / 4 / 14 us LATER; a static s_setprio 1 for waves 4-7 and the
  //  two workgroup barriers around epilogue A's staging change nothing.)

  // ---- loader roles ------------------------------------------------------------------------------------------------------------------------
  const bool wl = wave < 4;
  const int wq = wave & 3;
  const unsigned lane16 = lane * 16;
  // operand rows: this loader's piece k = rows 16 (k & 1) .. +16 of consumer wave 2 wq + (k >> 1); lane -> row (lane >> 2), 16-byte column
  // (lane & 3) ^ ((lane >> 4) & 3)   [= col ^ ((row >> 2) & 3): the tile's swizzle]
  const unsigned rb_dz = (unsigned)(p.dz_stride * ES), rb_ds = (unsigned)(p.Sp * ES);
  const i32x4 srd_dz = srd_of(p.dz + (int64_t)b * p.T * rb_dz, (unsigned)p.T * rb_dz - (unsigned)(p.dz_stride - Z2) * ES);
  const i32x4 srd_ds = srd_of(p.dskip + (int64_t)b * p.T * rb_ds, (unsigned)p.T * rb_ds);
  unsigned vdz[4], vds[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = t0 + 32 * (2 * wq + (k >> 1)) + 16 * (k & 1) + (lane >> 2);
    const unsigned swz = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
    vdz[k] = (unsigned)row * rb_dz + swz;
    vds[k] = (unsigned)row * rb_ds + swz;
  }
  const unsigned dshift = (unsigned)p.dilation * rb_dz;      // one tap step along time, in bytes of the dz array
  // piece k of X(c) for this wave's role.  Everything but the wave's quarter (wq) is a compile-time constant.
  auto issue = [&](auto cc, auto kc) {
    constexpr int c = decltype(cc)::value, k = decltype(kc)::value;
    if constexpr (c < NH) {
      constexpr int ph = PL::phase(c), slot = c % NSW, tile = c % NTB;
      if (wl) {
        if constexpr (k < PL::nW(c) && !(WAE_GB8_ABL & 2)) {
          const char* src;
          unsigned dst = lds0 + slot * SLOT;
          if constexpr (ph == 0) {
            constexpr int q = c / 2, hk = c % 2;
            if constexpr (k < 4) { src = p.w_x + ((int64_t)q * 32768 + hk * 16384 + k * 1024) + wq * 4096; dst += k * 1024 + wq * 4096; }
            else { src = p.w_c + ((int64_t)PL::cb(c) * 8192 + hk * 4096) + wq * 1024; dst += 16384 + wq * 1024; }
          } else if constexpr (ph == 1) {
            src = p.w_uo + ((int64_t)(c - HA) * 16384 + k * 1024) + wq * 4096; dst += k * 1024 + wq * 4096;
          } else {
            constexpr int j = c - HA - HB1, q = j / 2, hk = j % 2, ppw = NTU / 2;
            src = p.w_us + ((int64_t)q * (NTU * 4096) + hk * (NTU * 2048) + k * 1024) + wq * (ppw * 1024); dst += k * 1024 + wq * (ppw * 1024);
          }
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane16),
                                           (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
        }
      } else {
        if constexpr (k < PL::nB(c) && !(WAE_GB8_ABL & 1)) {
          const unsigned m0v = lds0 + RING + tile * TILEB + (2 * wq + (k >> 1)) * 2048 + (k & 1) * 1024;
          if constexpr (ph == 0) {
            constexpr int hk = c % 2;
            const unsigned vo = vdz[k] + (unsigned)(2 - PL::tap(c)) * dshift;
            const unsigned so = PL::cb(c) * 128 + hk * 64;
            const i32x4 sr = srd_dz;
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(vo), "s"(sr), "s"(so) : "m0");
          } else {
            constexpr int j = c - HA - HB1;
            const unsigned vo = vds[k];
            const unsigned so = (j / 2) * 128 + (j % 2) * 64;
            const i32x4 sr = srd_ds;
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(vo), "s"(sr), "s"(so) : "m0");
          }
        }
      }
    }
  };
  auto issue_all = [&](auto cc) { sfor<0, 5>([&](auto kc) { issue(cc, kc); }); };
  // top of half-chunk c: this wave's pieces through X(c + 1) have landed; then every wave's, and every wave has left half-chunk c - 1
  auto top = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (wl) wait_vmc<PL::allow_w(c)>(); else wait_vmc<PL::allow_b(c)>();
    __builtin_amdgcn_s_barrier();
  };

  // ---- consumer side -------------------------------------------------------------------------------------------------------------------------
  unsigned a_base[2];
  a_base[0] = lds0 + lane16;
  a_base[1] = a_base[0] + 3 * SLOT;
  auto a_rd = [&](auto slotc, auto offc, frag& dst) {      // 1-KiB block at byte `off` of ring slot `slot`
    constexpr int slot = decltype(slotc)::value, off = decltype(offc)::value;
    lds_rd<(slot % 3) * SLOT + off>(dst, a_base[slot / 3]);
  };
  unsigned b_addr[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) b_addr[f] = lds0 + RING + wave * 2048 + n * 64 + (((2 * f + h) ^ ((n >> 2) & 3)) << 4);
  auto b_rd = [&](auto tlc, frag (&dst)[2]) {
    constexpr int tl = decltype(tlc)::value;
    lds_rd<tl * TILEB>(dst[0], b_addr[0]);
    lds_rd<tl * TILEB>(dst[1], b_addr[1]);
  };

  // ---- prologue: X(0) .. X(D - 1); half-chunk "-1": X(0) visible, the first operand fragments into registers ------------------------------
  sfor<0, DB>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (c < DW) { if (wl) issue_all(cc); }
    if (!wl) issue_all(cc);
  });
  if (wl) wait_vmc<PL::nW(1) + PL::nW(2)>(); else wait_vmc<PL::allow_b(-1)>();
  __builtin_amdgcn_s_barrier();
  frag Bf[2][2];
  b_rd(IntC<0>{}, Bf[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bf[0][0]), "+v"(Bf[0][1]));

  // ---- phase A: acc_x = sum_tap W1_tap^T dz_l  (+ acc_d = Wc_l^T dz_l on the shift-0 tap) ----------------------------------------------------
  f32x16 accx[NTX], accd[2];
#pragma unroll
  for (int m = 0; m < NTX; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accx[m][r] = 0.f;
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[m][r] = 0.f;
  {
    frag a[PD];
    // step (c, i) of phase A: i < 16: block (blk = i / 8, m = i % 8) of the main half-chunk; i >= 16: block (blk, m) = ((i - 16) / 2, (i - 16) % 2) of
    // the dc block.  ahead(c, i, k): the step k positions later, or c = HA when the phase ends first.
    auto step_off = [](int i) constexpr { return i < 16 ? i * 1024 : 16384 + (i - 16) * 1024; };
    sfor<0, HA>([&](auto cc) {
      constexpr int c = decltype(cc)::value, ns = PL::nsteps(c), sp = ns / 5;
      top(cc);
      asm volatile("" : "+v"(Bf[c % 2][0]), "+v"(Bf[c % 2][1]));
      if constexpr (c == 0) {
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, PD>([&](auto ic) { a_rd(IntC<0>{}, IntC<step_off(decltype(ic)::value)>{}, a[decltype(ic)::value]); });
      }
      sfor<0, ns>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        // global position of this step in the phase's read stream
        constexpr int G = [] { int g = 0; for (int cc2 = 0; cc2 < c; ++cc2) g += PL::nsteps(cc2); return g + i; }();
        constexpr int GT = [] { int g = 0; for (int cc2 = 0; cc2 < HA; ++cc2) g += PL::nsteps(cc2); return g; }();
        constexpr int remaining = GT - 1 - G;
        constexpr int younger_a = remaining < PD - 1 ? remaining : PD - 1;
        constexpr int extra = (i >= 2 && i <= 1 + PD && c + 1 < HA) ? 2 : 0;      // the operand-fragment reads of step 1
        lds_wait<younger_a + extra>(a[G % PD]);
        if constexpr (i < 16) GB8_MMA(accx[i % 8], a[G % PD], Bf[c % 2][i / 8]);
        else GB8_MMA(accd[(i - 16) % 2], a[G % PD], Bf[c % 2][(i - 16) / 2]);
        if constexpr (remaining >= PD) {
          // the step PD positions later
          constexpr int c2 = [] { int cc2 = c, ii = i + PD; while (ii >= PL::nsteps(cc2)) { ii -= PL::nsteps(cc2); ++cc2; } return cc2; }();
          constexpr int i2 = [] { int cc2 = c, ii = i + PD; while (ii >= PL::nsteps(cc2)) { ii -= PL::nsteps(cc2); ++cc2; } return ii; }();
          a_rd(IntC<c2 % NSW>{}, IntC<step_off(i2)>{}, a[G % PD]);
        }
        if constexpr (i == 1 && c + 1 < HA) b_rd(IntC<(c + 1) % NTB>{}, Bf[(c + 1) % 2]);
        if constexpr (i % sp == 0 && i / sp < 5) {
          if (wl) issue(IntC<c + DW>{}, IntC<i / sp>{}); else issue(IntC<c + DB>{}, IntC<i / sp>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }

  // ---- epilogue A: dx_l-hat = alpha * (acc + residual), stored once, kept as the operand of B1; dc += acc_d -----------------------------------
  // (csrc/glu_bwd.hip: glu_bwd_pair_kernel's epilogue A; the staging tiles sit in the operand-tile area, idle between phase A and B2.)
  // Every wave is past its last reads of the operand tiles (their fragments are in registers since step 6 of the previous half-chunk);
  // the barrier below is the one in front of B1's first half-chunk, which this epilogue sits in front of.
  frag xf[NKB];
  {
    int le = lane;
    asm volatile("" : "+v"(le));
    const int ne = le & 31, he = le >> 5;
    frag rs[2 * NTX];
    const char* rp = p.g_next + ((int64_t)b * p.T + min(t0w + ne, p.T - 1)) * (int64_t)(NTX * 32 * ES) + he * 16;
    char* stg = smem + RING + wave * STGB;
    f32x4 w0[4], w1[4];
    const bool add = (p.dc_mode & 1) != 0;
    if (add && rows_valid > 0) {
      rmw_fetch(w0, p.dc_acc + row0 * 64, 0, rows_valid, le);
      rmw_fetch(w1, p.dc_acc + row0 * 64, 1, rows_valid, le);
    }
#pragma unroll
    for (int f = 0; f < NTX; ++f) rs[f] = *(const frag*)(rp + f * 32);
    __builtin_amdgcn_sched_barrier(0);
    // (the tile area is free for staging: every wave read its last operand fragments in half-chunk HA - 2 and has passed the barrier
    //  at the top of half-chunk HA - 1 since)
    if (rows_valid > 0) {
      char* o16 = (p.dc_mode & 2) ? p.dc_out + row0 * 64 * ES : nullptr;
      stage_rmw_tile<E>(stg, accd[0], w0, add, p.dc_acc + row0 * 64, o16, 0, rows_valid, le);
      stage_rmw_tile<E>(stg, accd[1], w1, add, p.dc_acc + row0 * 64, o16, 1, rows_valid, le);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = NTX; f < 2 * NTX; ++f) rs[f] = *(const frag*)(rp + f * 32);
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
  if (p.last) {      // layer 0: nothing below to gate
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // (the tile area goes back to the operand loaders without a barrier of its own: the first dS piece is requested inside a B1 half-chunk,
  //  behind that half-chunk's barrier, which every wave passes after its staging)

  // ---- phase B1: du = W_out^T dx_l-hat, one output tile per half-chunk, operand from registers -------------------------------------------------
  f32x16 accu[NTU];
#pragma unroll
  for (int m = 0; m < NTU; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) accu[m][r] = 0.f;
  const char* zrow = p.z_prev + row0 * Z2 * ES;
  f32x4 fa[8], fb[8];
  {
    frag a[PD];
    sfor<0, HB1>([&](auto jc) {
      constexpr int j = decltype(jc)::value, c = HA + j, ns = 16, sp = 3;
      top(IntC<c>{});
      if constexpr (j == 0) {
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, PD>([&](auto ic) { a_rd(IntC<c % NSW>{}, IntC<decltype(ic)::value * 1024>{}, a[decltype(ic)::value]); });
      }
      sfor<0, ns>([&](auto ic) {
        constexpr int i = decltype(ic)::value, G = j * 16 + i, GT = HB1 * 16;
        constexpr int remaining = GT - 1 - G;
        constexpr int younger_a = remaining < PD - 1 ? remaining : PD - 1;
        constexpr int extra = (j == HB1 - 1 && i >= 2 && i <= 1 + PD) ? 2 : 0;     // (the last half-chunk reads B2's first operand fragments)
        lds_wait<younger_a + extra>(a[G % PD]);
        GB8_MMA(accu[j], a[G % PD], xf[i]);
        if constexpr (remaining >= PD) {
          constexpr int G2 = G + PD;
          a_rd(IntC<(HA + G2 / 16) % NSW>{}, IntC<(G2 % 16) * 1024>{}, a[G % PD]);
        }
        if constexpr (j == HB1 - 1 && i == 1) b_rd(IntC<(c + 1) % NTB>{}, Bf[(c + 1) % 2]);
        if constexpr (i % sp == 0 && i / sp < 5) {
          if (wl) issue(IntC<c + DW>{}, IntC<i / sp>{}); else issue(IntC<c + DB>{}, IntC<i / sp>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  // the saved pre-activations of the first tile pair (xf is dead: their registers are free now); they arrive under B2
  if (rows_valid > 0) {
    stage_fetch_pass<E, 2>(fa, zrow, (int64_t)Z2 * ES, rows_valid, lane);
    stage_fetch_pass<E, 2>(fb, zrow + (int64_t)NTU * 32 * ES, (int64_t)Z2 * ES, rows_valid, lane);
  }

  // ---- phase B2: du += W_skip^T dskip ------------------------------------------------------------------------------------------------------------
  {
    frag a[PD];
    constexpr int C0 = HA + HB1, HB2 = PL::HB2, NS = 2 * NTU, GT = HB2 * NS;
    sfor<0, HB2>([&](auto jc) {
      constexpr int j = decltype(jc)::value, c = C0 + j, sp = NS / 5 > 0 ? NS / 5 : 1;
      top(IntC<c>{});
      asm volatile("" : "+v"(Bf[c % 2][0]), "+v"(Bf[c % 2][1]));
      if constexpr (j == 0) {
        __builtin_amdgcn_sched_barrier(0);
        sfor<0, PD>([&](auto ic) { a_rd(IntC<c % NSW>{}, IntC<decltype(ic)::value * 1024>{}, a[decltype(ic)::value]); });
      }
      sfor<0, NS>([&](auto ic) {
        constexpr int i = decltype(ic)::value, G = j * NS + i;
        constexpr int remaining = GT - 1 - G;
        constexpr int younger_a = remaining < PD - 1 ? remaining : PD - 1;
        constexpr int extra = (j + 1 < HB2 && i >= 2 && i <= 1 + PD) ? 2 : 0;
        lds_wait<younger_a + extra>(a[G % PD]);
        GB8_MMA(accu[i % NTU], a[G % PD], Bf[c % 2][i / NTU]);
        if constexpr (remaining >= PD) {
          constexpr int G2 = G + PD;
          a_rd(IntC<(C0 + G2 / NS) % NSW>{}, IntC<(G2 % NS) * 1024>{}, a[G % PD]);
        }
        if constexpr (j + 1 < HB2 && i == 1) b_rd(IntC<(c + 1) % NTB>{}, Bf[(c + 1) % 2]);
        if constexpr (i % sp == 0 && i / sp < 5) {
          if (wl) issue(IntC<c + DW>{}, IntC<i / sp>{}); else issue(IntC<c + DB>{}, IntC<i / sp>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                   // every wave is done with ring and tiles: staging area
  if (rows_valid <= 0) return;

  // ---- epilogue B: gate backward (modules.py:154: u = tanh(a) * sigmoid(b)):  da = du s (1 - th^2),  db = du th s (1 - s) ------------------------
  {
    int le = lane;
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
}

namespace {

template <typename E, int NTU, int SPQ>
int launch_pair8(const GbArgs& a, hipStream_t st) {
  auto kern = glu_bwd_pair8_kernel<E, NTU, SPQ>;
  const size_t lds = (size_t)4 * (16384 + 4096) + (size_t)4 * 8 * 2048;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)kern, lds_cache, lds, "glu_bwd_pair8"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(a.B * tiles), dim3(512), lds, st, a);
  return wae_check_launch("glu_bwd_pair8");
}

}  // namespace

int wae_glu_bwd8_launch(const GbArgs& a, int dtype, int ntx, int ntu, hipStream_t st, bool* handled) {
  *handled = false;
  if (!wae_is16(dtype) || ntx != 8 || !a.w_c || a.ktaps != 3 || a.Sp != 256 || a.stamps) return WAE_OK;
  if (ntu != 6 && ntu != 4) return WAE_OK;
  if ((int64_t)a.T * a.dz_stride * 2 >= (int64_t)1 << 31) return WAE_OK;            // 32-bit offsets inside a clip's descriptor
  *handled = true;
  if (dtype == WAE_BF16) return ntu == 6 ? launch_pair8<__bf16, 6, 4>(a, st) : launch_pair8<__bf16, 4, 4>(a, st);
  return ntu == 6 ? launch_pair8<f16, 6, 4>(a, st) : launch_pair8<f16, 4, 4>(a, st);
}
