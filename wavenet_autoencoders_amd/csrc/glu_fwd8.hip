// glu_fwd8: the fused ResidualConv1dGLU layer (reference: modules.py:115-163) for Rp = 256 / Ccp = 64 / three taps in 16-bit storage
// (BASELINE C2: Hp = 192; hps/vqwae.json: Hp = 128), with BOTH operand streams of GEMM 1 through LDS (round 6).
//
//   z[t]   = zb + sum_tap W1_tap x[t - (2 - tap) d] + Wc c[t]          GEMM 1   (modules.py:131-151; zb = conv bias + hoisted global conditioning)
//   u[t]   = tanh(z_a[t]) * sigmoid(z_b[t])                             gate     (modules.py:154)
//   x'[t]  = sqrt(.5) * (x[t] + W_out u[t] + b_out)                     GEMM 2   (modules.py:157-161; the skip branch is the head's contraction over u)
//
// glu_fwd_static_kernel (csrc/glu_fwd_static.hip) requests the activation operand in MFMA-operand shape (32 rows x 32 bytes per wave
// instruction) straight into registers, from the waves that also stream the weights and run the MFMAs.  What csrc/gemm_tm8.hip and
// csrc/glu_bwd8.hip measured this round holds here too: such a request costs the texture-address path four accesses where a
// 64-byte row piece costs one, and one wave carrying two request streams runs both at the depth of the shallower.  Here
//   * the activation rows travel as 16-row x 64-byte LDS-DMA pieces through per-clip buffer descriptors (a row before t = 0 -- the
//     causal pad -- wraps to an offset beyond num_records, a row past the clip's end is beyond it: the hardware writes zeros) into
//     swizzled tiles, NTB half-chunks ahead;
//   * the packed weights (the SAME packed stream: packing.py glu_w1_map / glu_pass_tiles) go through a 4-slot ring of K = 32
//     half-chunks three ahead; GEMM 2's four units follow in the same ring;
//   * waves 0-3 issue the weight pieces, waves 4-7 the operand pieces, all eight compute; one workgroup barrier per half-chunk.
// Same fragment layouts and the same MFMA order per accumulator as glu_fwd_static_kernel, the same epilogues: results are BITWISE
// its results (tests/test_gpu_parity.py; WAE_GLU_STATIC_REG in the descriptor flags selects the round-3 kernel).
#include "glu_fwd.hpp"

// timing-only ablations (tools/glu_ab.py on variant builds; results are wrong): 1 no weight pieces in the stream, 2 no operand pieces,
// 16 no MFMAs
#ifndef WAE_G8_ABL
#define WAE_G8_ABL 0
#endif

namespace {

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F>
__device__ __forceinline__ void sfor8(F&& f) {
  if constexpr (I < N) {
    f(IntC<I>{});
    sfor8<I + 1, N>(f);
  }
}
template <int CNT>
__device__ __forceinline__ void wait_vm8() {
  static_assert(CNT >= 0 && CNT < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");
}
template <int OFF, typename frag>
__device__ __forceinline__ void lds_rd8(frag& dst, unsigned addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF, typename F>
__device__ __forceinline__ void bload8(F& dst, unsigned voff, i32x4 rsrc) {
  static_assert(sizeof(F) == 16 && OFF >= 0 && OFF < 4096, "one 16-byte fragment, 12-bit offset");
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "+v"(dst) : "v"(voff), "s"(rsrc), "n"(OFF));
}
__device__ __forceinline__ i32x4 srd8(const char* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));
  r.z = __builtin_amdgcn_readfirstlane((int)bytes);
  r.w = 0x00020000;
  return r;
}

}  // namespace

// NP = Hp / 32 gate-channel tiles; NPHP = tiles per pass of the PACKED weight stream (packing.py: glu_pass_tiles); GEMM 1 runs in one pass.
template <typename E, int NP, int NPHP, bool SAVE_Z, bool NO_OUT>
__device__ __forceinline__ void glu_fwd8_body(const GluArgs& p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  static_assert(sizeof(E) == 2 && T_::CK == 64, "16-bit storage");
  constexpr int NW = 8, ES = 2, KT = 3, CPR = 4, PD = 4;
  constexpr int NM = 2 * NP, NMP = 2 * NPHP, PP = NP / NPHP, PCHB = NMP * 4 * 1024;
  constexpr int NQC = KT * CPR, PQ1 = NQC + 1, NH1 = 2 * PQ1;          // half-chunks of GEMM 1
  constexpr int SLOT = NM * 2 * 1024, NKB = NP * 2, T2U = SLOT / (NKB * 1024), NT2 = 8, NU2 = NO_OUT ? 0 : NT2 / T2U;
  constexpr int NSW = 4, DW = 3, NTB = SLOT > 16384 ? 3 : 4, DB = NTB;
  constexpr int TILEB = NW * 2048, RING = NSW * SLOT, TILES = NTB * TILEB;
  constexpr int PPW = SLOT / 4 / 1024, NOP = PPW > 4 ? PPW : 4, NSTEP = 2 * NM, SP = NSTEP / NOP;
  constexpr int ROWX = CPR * 128, ROWC = 128, RP = CPR * 64, HP = NP * 32;
  constexpr int PITCH = 128, STG = 32 * PITCH;
  static_assert(NP % NPHP == 0 && (2 * NMP) % PPW == 0 && T2U * NKB * 1024 == SLOT && NT2 % T2U == 0, "a loader's pieces stay inside one packed pass");
  static_assert(RING + TILES + (RP + 2 * HP) * 4 <= 160 * 1024 && NW * STG <= TILES && NU2 <= NSW && NOP * SP <= NSTEP, "LDS budget; staging sits in the tile area");
  static_assert(2 * SLOT <= 65536, "two slots per ds_read base register");
  // global store instructions of the gate epilogue for a full 32-row group (stage_store_tiles<E, NT, 128>: 4 per tile pair, 2 per single)
  constexpr int ST_PASS = (NP / 2) * 4 + (NP % 2) * 2;
  constexpr int GH = (NP >= 6 && NP % 2 == 0) ? NP / 2 : NP;
  constexpr int ST_U = (NP / GH) * ((GH / 2) * 4 + (GH % 2) * 2);
  constexpr int NST = (SAVE_Z ? 2 * ST_PASS : 0) + ST_U;
  static_assert(NST + 8 < 64, "vmcnt is a 6-bit field");

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
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;
  float* bias_lds = (float*)(smem + RING + TILES);
  float* zb_lds = bias_lds + RP;
  char* stg = smem + RING + wave * STG;

  const unsigned clip_x = (unsigned)p.T * ROWX, clip_c = (unsigned)p.T * ROWC;
  const i32x4 srd_xc = srd8(p.x_conv + (int64_t)b * clip_x, clip_x);   // convolution operand (modules.py:127-131)
  const i32x4 srd_xr = srd8(p.x_in + (int64_t)b * clip_x, clip_x);     // residual path (modules.py:126,161)
  const i32x4 srd_c = srd8(p.c_up + (int64_t)b * clip_c, clip_c);

  // ---- loader roles ------------------------------------------------------------------------------------------------------------------
  const bool wl = wave < 4;
  const int wq = wave & 3;
  const unsigned lane16 = lane * 16;
  // weights: pieces L0 .. L0 + PPW - 1 of every unit's LDS image ([packed pass][k-block][tile] in GEMM 1, linear in GEMM 2)
  const int L0 = wq * PPW;
  const unsigned wsrc1 = (unsigned)(((L0 / (2 * NMP)) * PQ1) * PCHB + (L0 % (2 * NMP)) * 1024) + lane16;
  const unsigned wsrc2 = (unsigned)(PP * PQ1 * PCHB + L0 * 1024) + lane16;
  // operand: piece k = rows 16 (k & 1) .. +16 of consumer wave 2 wq + (k >> 1); lane -> row (lane >> 2), 16-byte column
  // (lane & 3) ^ ((lane >> 4) & 3)   [= col ^ ((row >> 2) & 3): the tile's swizzle, applied on the global side]
  const int rowl = t0 + 64 * wq + (lane >> 2);
  const unsigned swz = ((lane & 3) ^ ((lane >> 4) & 3)) << 4;
  const int dil = p.dilation;
  auto issue = [&](auto cc, auto kc) {
    constexpr int c = decltype(cc)::value, k = decltype(kc)::value;
    if constexpr (c < NH1 + NU2) {
      constexpr int slot = c % NSW, tile = c % NTB;
      if (wl) {
        if constexpr (k < PPW && !(WAE_G8_ABL & 1)) {
          const unsigned dst = lds0 + slot * SLOT + (L0 + k) * 1024;
          const char* src;
          if constexpr (c < NH1) src = p.w + ((int64_t)(c / 2) * PCHB + (c % 2) * (2 * NMP * 1024) + k * 1024) + wsrc1;
          else src = p.w + ((int64_t)(c - NH1) * SLOT + k * 1024) + wsrc2;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(uintptr_t)dst, 16, 0, 0);
        }
      } else {
        if constexpr (k < 4 && c < NH1 && !(WAE_G8_ABL & 2)) {
          constexpr int q = c / 2, hk = c % 2;
          const unsigned m0v = lds0 + RING + tile * TILEB + (2 * wq + (k >> 1)) * 2048 + (k & 1) * 1024;
          if constexpr (q < NQC) {
            constexpr int cblk = q / KT, tap = q % KT;   // taps of one column block back to back (packing.py: glu_w1_map)
            const unsigned vo = (unsigned)(rowl + 16 * k - (KT - 1 - tap) * dil) * ROWX + swz;
            const unsigned so = cblk * 128 + hk * 64;
            const i32x4 sr = srd_xc;
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(vo), "s"(sr), "s"(so) : "m0");
          } else {
            const unsigned vo = (unsigned)(rowl + 16 * k) * ROWC + swz;
            const unsigned so = (q - NQC) * 128 + hk * 64;
            const i32x4 sr = srd_c;
            asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(m0v), "v"(vo), "s"(sr), "s"(so) : "m0");
          }
        }
      }
    }
  };
  auto issue_all = [&](auto cc) { sfor8<0, NOP>([&](auto kc) { issue(cc, kc); }); };
  // pieces a loader has in flight per unit, and its counted wait at the top of half-chunk c: everything through X(c + 1) has landed
  auto top = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    constexpr int aw = [] { int s = 0; for (int i = c + 2; i <= c + DW - 1; ++i) s += (i >= 0 && i < NH1 + NU2) ? PPW : 0; return s; }();
    constexpr int ab = [] { int s = 0; for (int i = c + 2; i <= c + DB - 1; ++i) s += (i >= 0 && i < NH1) ? 4 : 0; return s; }();
    if (wl) wait_vm8<aw>(); else wait_vm8<ab>();
    __builtin_amdgcn_s_barrier();
  };

  // ---- consumer side -----------------------------------------------------------------------------------------------------------------
  unsigned a_base[2];
  a_base[0] = lds0 + lane16;
  a_base[1] = a_base[0] + 2 * SLOT;
  auto a_rd = [&](auto cc, auto ic, frag& dst) {   // A fragment block I = (k-block jb, accumulator tile a) of half-chunk c
    constexpr int c = decltype(cc)::value, I = decltype(ic)::value, slot = c % NSW;
    constexpr int jb = I / NM, a = I % NM;
    // accumulator tiles: [tanh tiles 0 .. NP) | sigmoid tiles 0 .. NP); packed pass of a tile = tile / NPHP
    constexpr int tt = a < NP ? a : a - NP;
    constexpr int L = (tt / NPHP) * (2 * NMP) + jb * NMP + (a < NP ? tt % NPHP : NPHP + tt % NPHP);
    lds_rd8<(slot & 1) * SLOT + L * 1024>(dst, a_base[slot / 2]);
  };
  unsigned b_addr[2];
#pragma unroll
  for (int f = 0; f < 2; ++f) b_addr[f] = lds0 + RING + wave * 2048 + n * 64 + (((2 * f + h) ^ ((n >> 2) & 3)) << 4);
  auto b_rd = [&](auto tlc, frag (&dst)[2]) {
    constexpr int tl = decltype(tlc)::value;
    lds_rd8<tl * TILEB>(dst[0], b_addr[0]);
    lds_rd8<tl * TILEB>(dst[1], b_addr[1]);
  };

  // ---- prologue: tables, X(0) .. X(D - 1); half-chunk "-1": X(0) visible, the first operand fragments into registers -----------------
  static_assert(RP <= NW * 256 && 2 * HP <= NW * 256, "one 16-byte table piece per thread");
  const bool has_tb = NU2 > 0 && (int)threadIdx.x * 4 < RP, has_tz = (int)threadIdx.x * 4 < 2 * HP;
  // (asm requests: hipcc's own wait for a plain load cannot see the asm pieces behind it and drains the whole prologue burst)
  f32x4 tb = {}, tz = {};
  const unsigned tab_off = threadIdx.x * 16;
  const unsigned long long zba = (unsigned long long)(p.zb + (int64_t)b * p.zb_stride);
  const unsigned long long zbu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(zba >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)zba);
  if (has_tb) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(tb) : "v"(tab_off), "s"(p.bias_out));
  if (has_tz) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(tz) : "v"(tab_off), "s"(zbu));
  sfor8<0, DB>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (c < DW) { if (wl) issue_all(cc); }
    if (!wl) issue_all(cc);
  });
  if (wl) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(tb), "+v"(tz) : "n"(DW * PPW));
  else asm volatile("s_waitcnt vmcnt(%2)" : "+v"(tb), "+v"(tz) : "n"(DB * 4));
  if (has_tb) *(f32x4*)(bias_lds + threadIdx.x * 4) = tb;
  if (has_tz) *(f32x4*)(zb_lds + threadIdx.x * 4) = tz;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  top(IntC<-1>{});
  frag Bf[2][2];
  b_rd(IntC<0>{}, Bf[0]);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bf[0][0]), "+v"(Bf[0][1]));

  // ---- GEMM 1: accumulators start from zb = conv bias + hoisted global conditioning ----------------------------------------------------
  f32x16 acc[NM];
#pragma unroll
  for (int m = 0; m < NM; ++m) init_rows(acc[m], zb_lds + (m < NP ? 32 * m : HP + 32 * (m - NP)), h);
  {
    frag a[PD];
    constexpr int NG = NH1 * NSTEP;
    sfor8<0, NH1>([&](auto cc) {
      constexpr int c = decltype(cc)::value;
      top(cc);
      asm volatile("" : "+v"(Bf[c % 2][0]), "+v"(Bf[c % 2][1]));
      if constexpr (c == 0) {
        __builtin_amdgcn_sched_barrier(0);
        sfor8<0, PD>([&](auto ic) { a_rd(IntC<0>{}, ic, a[decltype(ic)::value]); });
      }
      sfor8<0, NSTEP>([&](auto ic) {
        constexpr int i = decltype(ic)::value, G = c * NSTEP + i;
        constexpr int remaining = NG - 1 - G;
        constexpr int younger_a = remaining < PD - 1 ? remaining : PD - 1;
        constexpr int extra = (i >= 2 && i <= 1 + PD && c + 1 < NH1) ? 2 : 0;      // the operand-fragment reads of step 1
        lds_wait<younger_a + extra>(a[G % PD]);
        if constexpr (!(WAE_G8_ABL & 16)) mma32(acc[i % NM], a[G % PD], Bf[c % 2][i / NM]);
        if constexpr (remaining >= PD) {
          constexpr int G2 = G + PD;
          a_rd(IntC<G2 / NSTEP>{}, IntC<G2 % NSTEP>{}, a[G % PD]);
        }
        if constexpr (i == 1 && c + 1 < NH1) b_rd(IntC<(c + 1) % NTB>{}, Bf[(c + 1) % 2]);
        if constexpr (i % SP == 0 && i / SP < NOP) {
          if (wl) issue(IntC<c + DW>{}, IntC<i / SP>{});      // (c + DW <= NH1 + 2: GEMM 2's first three units ride here)
          else issue(IntC<c + DB>{}, IntC<i / SP>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  const bool full_rows = __builtin_amdgcn_readfirstlane(p.T - t0w) >= 32;   // wave-uniform
  // (a laundered lane id: hipcc otherwise hoists the epilogues' per-lane staging addresses to the top of the kernel, spills them across
  //  GEMM 1 and reloads them between the stores -- and a scratch reload waits for every request in flight; csrc/glu_bwd.hip, round 5)
  int le = lane;
  asm volatile("" : "+v"(le));
  const int ne = le & 31, he = le >> 5;
  const unsigned voff_res = (unsigned)((t0w + ne) * ROWX + he * 16);
  frag res[8];   // residual x[t] as operand-shaped 16-byte fragments, four output tiles at a time
  frag uf[NKB];
  if constexpr (NU2 > 0) {
    // every wave has left GEMM 1: its last slot takes GEMM 2's last unit, the tile area becomes the staging area
    __builtin_amdgcn_s_barrier();
    if (wl) issue_all(IntC<NH1 + NU2 - 1>{});
  } else {
    __builtin_amdgcn_s_barrier();
  }
  const int rows_valid = min(max(p.T - t0w, 0), 32);
  const int64_t row0 = (int64_t)b * p.T + t0w;
  // ---- optional z save (training): rows of 2Hp elements, a-half then b-half ------------------------------------------------------------
  if constexpr (SAVE_Z) {
    if (rows_valid > 0) {
      char* zr = p.z_save + row0 * (2 * HP) * ES;
      stage_store_tiles<E, NP, PITCH>(stg, &acc[0], zr, (int64_t)2 * HP * ES, rows_valid, le);
      stage_store_tiles<E, NP, PITCH>(stg, &acc[NP], zr + (int64_t)HP * ES, (int64_t)2 * HP * ES, rows_valid, le);
    }
  }
  // ---- gate: u = tanh(a) * sigmoid(b); stored once for the head's skip GEMM, and converted in place to the operand fragments of GEMM 2
  // (csrc/glu_fwd_static.hip: the same arithmetic, in halves so that the stored tiles' registers are free before the next half's temporaries)
#pragma unroll
  for (int gh = 0; gh < NP / GH; ++gh) {
#pragma unroll
    for (int pr = gh * GH; pr < (gh + 1) * GH; ++pr) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float av = acc[pr][r], g = acc[NP + pr][r];
        // tanh(a)*sigmoid(g) = (1-ea) / ((1+ea)(1+eg)), ea = e^-2a, eg = e^-g.  a is clamped from below so that ea stays finite
        const f32x2 sc = {-2.885390081777927f, -1.4426950408889634f};
        float amax;
        asm("v_max_f32 %0, %1, %2" : "=v"(amax) : "v"(av), "v"(-15.0f));
        f32x2 ag = {amax, g};
        ag = ag * sc;
        const float ea = __builtin_amdgcn_exp2f(ag.x);
        const float eg = __builtin_amdgcn_exp2f(ag.y);
        const f32x2 one = {1.0f, 1.0f};
        const f32x2 e2 = {ea, eg};
        const f32x2 d = e2 + one;
        acc[pr][r] = (1.0f - ea) * fast_rcp(d.x * d.y);
      }
      if constexpr (NU2 > 0) {   // (the last layer's launch has no second GEMM: x' is dead, wavenet.py:205-207)
        frag tmp[2];
        acc_to_frags(acc[pr], tmp);
        uf[pr * 2] = tmp[0];
        uf[pr * 2 + 1] = tmp[1];
      }
      __builtin_amdgcn_sched_barrier(0);  // one tile at a time: keeps the gate's temporaries from piling up
    }
    if (rows_valid > 0) {
      char* ur = p.u_out + (row0 * p.u_stride + gh * GH * 32) * ES;
      stage_store_tiles<E, GH, PITCH>(stg, &acc[gh * GH], ur, p.u_stride * ES, rows_valid, le);
    }
  }

  // ---- GEMM 2 + residual epilogue: every output tile in registers, the four units back to back ------------------------------------------
  if constexpr (NU2 > 0) {
    // Residual rows of the first four output tiles (L2 hits: tap k-1 of GEMM 1 read the same bytes), requested BEHIND the gate epilogue's
    // stores and retired after GEMM 2.  (glu_fwd_static_kernel requests them ahead of those stores; with twelve accumulator tiles live
    // that leaves hipcc 32 registers for the staging passes, and in this kernel it answered by spilling a requested fragment while its
    // load was in flight: a wild write when the load landed.  The second half's rows wait for the same stores in either kernel.)
    sfor8<0, 8>([&](auto fc) {
      constexpr int f = decltype(fc)::value;
      const frag zf = {};
      res[f] = zf;
      bload8<f * 32>(res[f], voff_res, srd_xr);
    });
    // this wave's weight pieces are older than its epilogue stores and the residual requests: those stay in flight (static count for a
    // full 32-row group; a wave at the clip's tail drains)
    if (full_rows) wait_vm8<NST + 8>();
    else wait_vm8<0>();
    __builtin_amdgcn_s_barrier();
    f32x16 y[NT2];
#pragma unroll
    for (int mt = 0; mt < NT2; ++mt) init_rows(y[mt], bias_lds + 32 * mt, he);
    NoFiller nf;
    sfor8<0, NU2>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const char* buf = smem + ((NH1 + j) % NSW) * SLOT + le * 16;
      gemm_chunk_fill<T2U * NKB, T2U, NKB, true, 4>(buf, uf, *(f32x16(*)[T2U]) & y[j * T2U], nf);
    });
    const f32x2 rs = {0.70710678118654752440f, 0.70710678118654752440f};
    sfor8<0, NT2 / 4>([&](auto hc) {
      constexpr int hf = decltype(hc)::value;
      // the residual fragments of the second half are older than the first half's 8 store instructions: those stay in flight
      if constexpr (hf == 0) {
        wait_vm8<0>();
      } else {
        if (full_rows) wait_vm8<8>();
        else wait_vm8<0>();
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
        sfor8<0, 8>([&](auto fc) { bload8<(hf + 1) * 256 + decltype(fc)::value * 32>(res[decltype(fc)::value], voff_res, srd_xr); });
      if (rows_valid > 0) {
        char* orow = p.x_out + ((int64_t)b * p.T + t0w) * ROWX + (int64_t)hf * 128 * ES;
        stage_store_tiles<E, 4, PITCH>(stg, &y[4 * hf], orow, ROWX, rows_valid, le);
      }
    });
  }
}

// (two differently NAMED entry points, as in glu_fwd_static.hip: rocprofv3's kernel stats tell training launches from inference launches)
template <typename E, int NP, int NPHP, bool NO_OUT>
__global__ void __launch_bounds__(512, 1) glu_fwd8_kernel(GluArgs p) {     // inference launch: no z
  glu_fwd8_body<E, NP, NPHP, false, NO_OUT>(p);
}
template <typename E, int NP, int NPHP, bool NO_OUT>
__global__ void __launch_bounds__(512, 1) glu_fwd8_z_kernel(GluArgs p) {   // training launch: also stores the pre-activations z
  glu_fwd8_body<E, NP, NPHP, true, NO_OUT>(p);
}

namespace {

template <typename E, int NP, int NPHP, bool SAVE_Z, bool NO_OUT>
int launch8(const GluArgs& a, hipStream_t st) {
  constexpr int SLOT = 2 * NP * 2 * 1024, NTB = SLOT > 16384 ? 3 : 4;
  void (*kern)(GluArgs);
  if constexpr (SAVE_Z) kern = glu_fwd8_z_kernel<E, NP, NPHP, NO_OUT>;
  else kern = glu_fwd8_kernel<E, NP, NPHP, NO_OUT>;
  const size_t lds = (size_t)4 * SLOT + (size_t)NTB * 8 * 2048 + (size_t)(256 + 2 * NP * 32) * 4;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)kern, lds_cache, lds, "glu_fwd8"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 255) / 256;
  hipLaunchKernelGGL(kern, dim3(a.B * tiles), dim3(512), lds, st, a);
  return wae_check_launch("glu_fwd8");
}

template <typename E, int NP, int NPHP>
int launch8_flags(const GluArgs& a, hipStream_t st) {
  const bool sz = a.flags & WAE_GLU_SAVE_Z, no = a.flags & WAE_GLU_NO_OUT;
  if (sz) return no ? launch8<E, NP, NPHP, true, true>(a, st) : launch8<E, NP, NPHP, true, false>(a, st);
  return no ? launch8<E, NP, NPHP, false, true>(a, st) : launch8<E, NP, NPHP, false, false>(a, st);
}

template <typename E>
int dispatch8(const GluArgs& a, hipStream_t st, bool* handled) {
  *handled = true;
  if (a.Hp == 192) return launch8_flags<E, 6, 3>(a, st);   // BASELINE C2 (inae dims, G = 368)
  if (a.Hp == 128) return launch8_flags<E, 4, 4>(a, st);   // hps/vqwae.json (C1 / C3 / C4)
  *handled = false;
  return WAE_OK;
}

}  // namespace

int wae_glu_fwd8_launch(const GluArgs& a, int dtype, hipStream_t st, bool* handled) {
  *handled = false;
  if (!wae_is16(dtype) || !a.c_up || a.stamps) return WAE_OK;
  if (a.ktaps != 3 || a.Ccp != 64 || a.Rp != 256) return WAE_OK;
  // the buffer descriptors address a clip with 32-bit offsets; a negative row must wrap beyond num_records
  if ((int64_t)a.T * a.Rp * 2 >= (int64_t)1 << 31) return WAE_OK;
  if (dtype == WAE_BF16) return dispatch8<__bf16>(a, st, handled);
  return dispatch8<f16>(a, st, handled);
}
