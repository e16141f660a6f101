// wae_glu_layer_fwd: one ResidualConv1dGLU layer + skip accumulate, fused (reference: modules.py:115-163,
// wavenet.py:204-207).
//
//   z[2Hp, t]  = zb + W1[2Hp, k*Rp + Ccp] . [x[t-(k-1)d] ; ... ; x[t] ; c[t]]      (GEMM 1, MFMA)
//   u[Hp, t]   = tanh(z_a) * sigmoid(z_b)                                           (registers)
//   y[Rp+Sp,t] = bias2 + W2[Rp+Sp, Hp] . u                                          (GEMM 2, MFMA, u never leaves
//   x'[t] = (y_out + x[t]) * sqrt(.5) ;  skip[t] (+)= y_skip                         the register file)
//
// Work decomposition: one workgroup = 128 consecutive time steps of one clip, 4 waves (one per SIMD), each
// wave owns 32 time columns and ALL channels, so the gate and the second GEMM need no cross-wave exchange:
// the 32x32 accumulator tiles of GEMM 1 (column = time on the lane, rows = channels in the registers)
// are converted in place into the B operand of GEMM 2 (cdna_hip_programming.md section 3, "An accumulator
// tile as the next MFMA's operand").  Weights arrive pre-packed in A-fragment order and are streamed through
// a double-buffered LDS ring by LDS-DMA, shared by the 4 waves; the activation (B) operand is read straight
// from HBM/L2 as 16-byte fragments (time-major rows, channels innermost), zero-filled before t=0 (causal pad).
#include "wae_common.hpp"

// timing-only ablation bits (tools/ablate_glu.py); outputs are wrong when any is set
#define DBG_NO_DMA 0x100
#define DBG_NO_BLOAD 0x200
#define DBG_NO_EPI 0x400
#define DBG_NO_GATE 0x800

struct GluArgs {
  const char* x_in;
  char* x_out;
  const char* c_up;
  float* skip;
  const float* zb;
  char* z_save;
  const char* w;
  const float* bias2;
  int64_t zb_stride;
  int B, T, Rp, Sp, Ccp, Hp, ktaps, dilation, flags;
  unsigned long long* stamps;  // diagnostic only (wae_debug_set_stamps): 16 x u64 per workgroup, else null
};

static unsigned long long* g_stamps = nullptr;
extern "C" void wae_debug_set_stamps(unsigned long long* dev_buf) { g_stamps = dev_buf; }
#define STAMP(i)                                                              \
  do {                                                                        \
    if (p.stamps) {                                                           \
      __builtin_amdgcn_sched_barrier(0);                                      \
      st_[i] = __builtin_amdgcn_s_memtime();                                  \
      __builtin_amdgcn_sched_barrier(0);                                      \
    }                                                                         \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// staged_rows: one wave moves a [32 time rows x 256 B] tile between the MFMA accumulator layout (lane = time
// column n, 4 consecutive channels per register group) and global-memory rows, through a wave-private 8 KiB
// LDS tile whose 16-byte chunks are XOR-swizzled by the row (conflict-free row reads, 2-way column accesses).
//   A. (LOAD_OLD) coalesced 16-B loads of the old rows  -> LDS
//   B. every lane reads its accumulator-layout pieces, applies op(y, old), writes the result back in place
//   C. coalesced 16-B row reads from LDS -> global stores
// EO = element type in memory (bf16: NTP = 4 tiles per 256-B row segment, f32: NTP = 2).
// ---------------------------------------------------------------------------------------------------
#define STG_BYTES 8192
// A: issue the coalesced loads of the old rows (call early; the data is consumed in stage_finish)
__device__ __forceinline__ void stage_load(f32x4 (&old)[8], const char* gin, int64_t row_stride, int rows_valid, int lane) {
  const int rr = lane >> 4, ck = lane & 15;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = min(4 * i + rr, rows_valid - 1);   // rows past T re-read the last valid row; never stored
    old[i] = *(const f32x4*)(gin + row * row_stride + ck * 16);
  }
}
// B + C
template <typename EO, int NTP, bool LOAD_OLD, typename F>
__device__ __forceinline__ void stage_finish(char* stg, f32x16* y, const f32x4 (&old)[8], char* gout, int64_t row_stride,
                                             int rows_valid, int lane, F op) {
  static_assert(NTP * 32 * sizeof(EO) == 256, "a staging pass covers 256 bytes per row");
  using vec4 = typename ET<EO>::vec4;
  const int n = lane & 31, h = lane >> 5;
  const int rr = lane >> 4, ck = lane & 15;
  if constexpr (LOAD_OLD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = 4 * i + rr;
      *(f32x4*)(stg + row * 256 + ((ck ^ (row & 15)) << 4)) = old[i];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
#pragma unroll
  for (int mt = 0; mt < NTP; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int c16, sub;
      if constexpr (sizeof(EO) == 2) { c16 = 4 * mt + g; sub = 8 * h; } else { c16 = 8 * mt + 2 * g + h; sub = 0; }
      char* a = stg + n * 256 + ((c16 ^ (n & 15)) << 4) + sub;
      f32x4 o = {0.f, 0.f, 0.f, 0.f};
      if constexpr (LOAD_OLD) o = to_f32x4(*(const vec4*)a);
      const f32x4 v = {y[mt][4 * g], y[mt][4 * g + 1], y[mt][4 * g + 2], y[mt][4 * g + 3]};
      *(vec4*)a = from_f32x4<EO>(op(v, o));
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = 4 * i + rr;
    const f32x4 v = *(const f32x4*)(stg + row * 256 + ((ck ^ (row & 15)) << 4));
    if (row < rows_valid) *(f32x4*)(gout + row * row_stride + ck * 16) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// 16 accumulator registers of a tile start from a per-row constant table: rows 8g + 4h + j, j < 4
__device__ __forceinline__ void init_rows(f32x16& acc, const float* tab /* 32 floats, 16-B aligned */, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = *(const f32x4*)(tab + 8 * g + 4 * h);
    acc[4 * g + 0] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
  }
}

template <typename E, int NP, bool EXACT>
__global__ void __launch_bounds__(256, 1) glu_fwd_kernel(GluArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  using vec4 = typename T_::vec4;
  constexpr int NM = 2 * NP;
  constexpr int CHB = NM * 4 * 1024;  // bytes per weight chunk
  constexpr int ES = sizeof(E);
  constexpr int KBU = T_::KBU;
  constexpr int MT2 = T_::MT2;
  constexpr int NKB = NP * KBU;  // 16-B k-blocks of GEMM 2
  static_assert(MT2 * NKB * 1024 == CHB, "GEMM-2 chunk must equal GEMM-1 chunk");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long st_[16] = {};
  STAMP(0);

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int tiles_per_b = (p.T + 127) >> 7;
  const int b = blockIdx.x / tiles_per_b;
  const int t = (blockIdx.x % tiles_per_b) * 128 + wave * 32 + n;
  const bool tvalid = t < p.T;

  const int cpr = p.Rp / T_::CK;            // chunks per tap
  const int nq_conv = p.ktaps * cpr;
  const int nq1 = nq_conv + p.Ccp / T_::CK;
  const int n_mt2 = (p.Rp + p.Sp) >> 5;
  const int mt2_first = (p.flags & WAE_GLU_NO_OUT) ? (p.Rp >> 5) : 0;
  const int nq2_first = mt2_first / MT2;
  const int nq2 = n_mt2 / MT2;
  const int nq_total = nq1 + (nq2 - nq2_first);

  const int64_t row_x = (int64_t)p.Rp * ES;
  const int64_t row_c = (int64_t)p.Ccp * ES;
  const char* xb = p.x_in + (int64_t)b * p.T * row_x;
  const char* cb = p.c_up ? p.c_up + (int64_t)b * p.T * row_c : nullptr;

  // chunk index in the packed stream -> byte offset (GEMM-2 chunks may start past the skipped out tiles)
  auto chunk_src = [&](int qi) -> const char* {
    int q = qi < nq1 ? qi : nq1 + nq2_first + (qi - nq1);
    return p.w + (int64_t)q * CHB;
  };

  const bool dbg_dma = !(p.flags & DBG_NO_DMA);
  frag Bn[4], Bc[4];
  auto load_B = [&](int q, frag (&Bf)[4]) {
    const char* src;
    bool ok = tvalid && !(p.flags & DBG_NO_BLOAD);
    if (q < nq_conv) {
      const int tap = q / cpr, cblk = q - tap * cpr;
      const int ts = t - (p.ktaps - 1 - tap) * p.dilation;
      ok = ok && ts >= 0;
      src = xb + (int64_t)ts * row_x + cblk * 128 + h * 16;
    } else {
      src = cb + (int64_t)t * row_c + (q - nq_conv) * 128 + h * 16;
    }
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

  // ---- accumulators start from zb = conv bias + hoisted global conditioning -------------------------
  f32x16 acc[NM];
  {
    const float* zbb = p.zb + (int64_t)b * p.zb_stride;
#pragma unroll
    for (int m = 0; m < NM; ++m) init_rows(acc[m], zbb + (m < NP ? 32 * m : p.Hp + 32 * (m - NP)), h);
  }

  // bias2 -> LDS once (read back per chunk with ds_read: keeps the second GEMM's accumulator init off vmcnt,
  // where it would drain the LDS-DMA and row-prefetch queues)
  float* bias_lds = (float*)(smem + 2 * CHB + 4 * STG_BYTES);
  for (int i = threadIdx.x * 4; i < p.Rp + p.Sp; i += 1024) *(f32x4*)(bias_lds + i) = *(const f32x4*)(p.bias2 + i);
  if (dbg_dma) dma_chunk(chunk_src(0), smem, CHB, wave, lane);
  load_B(0, Bn);
  STAMP(1);

  // ---- GEMM 1 ----------------------------------------------------------------------------------------
  for (int q = 0; q < nq1; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
    if (q + 1 < nq_total && dbg_dma) dma_chunk(chunk_src(q + 1), smem + ((q + 1) & 1) * CHB, CHB, wave, lane);
    if (q + 1 < nq1) load_B(q + 1, Bn);
    const char* buf = smem + (q & 1) * CHB + lane * 16;
    gemm_chunk<4 * NM, NM, 4>(buf, Bc, acc);
  }

  STAMP(2);
  // ---- optional z save (training) ----------------------------------------------------------------------
  if ((p.flags & WAE_GLU_SAVE_Z) && tvalid) {
    char* zr = p.z_save + ((int64_t)b * p.T + t) * (2 * p.Hp) * ES;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      const int row0 = (m < NP ? 32 * m : p.Hp + 32 * (m - NP)) + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
        *(vec4*)(zr + (row0 + 8 * g) * ES) = from_f32x4<E>(v);
      }
    }
  }

  // ---- gate: u = tanh(a) * sigmoid(b), converted in place to GEMM-2 operand fragments ------------------
  frag uf[NKB];
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
    f32x16 u;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float a = acc[pr][r], g = acc[NP + pr][r];
      if (p.flags & DBG_NO_GATE) {
        u[r] = a * g;
      } else if constexpr (EXACT) {
        u[r] = tanhf(a) * (1.0f / (1.0f + expf(-g)));
      } else {
        // tanh(a)*sigmoid(g) = (1-ea) / ((1+ea)(1+eg)), ea = e^-2a, eg = e^-g; exponents clamped so that the
        // product of the two denominators stays finite (|a| <= 15 is exact in fp32: tanh(15) == 1 - 2e-13)
        const float ea = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(a, -15.0f, 15.0f) * -2.885390081777927f);
        const float eg = __builtin_amdgcn_exp2f(fminf(g * -1.4426950408889634f, 60.0f));
        u[r] = (1.0f - ea) * fast_rcp((1.0f + ea) * (1.0f + eg));
      }
    }
    frag tmp[KBU];
    acc_to_frags(u, tmp);
#pragma unroll
    for (int s = 0; s < KBU; ++s) uf[pr * KBU + s] = tmp[s];
  }

  STAMP(3);
  // ---- GEMM 2 + epilogues ------------------------------------------------------------------------------
  // Each chunk = MT2 M-tiles (out or skip rows) against all of u.  The accumulator layout (lane = time column,
  // registers = channels) is turned into full 256-byte row segments through a wave-private swizzled LDS tile so
  // that every global access of the residual read, the x' store and the skip read-modify-write is coalesced.
  char* stg = smem + 2 * CHB + wave * STG_BYTES;
  const int t0w = (blockIdx.x % tiles_per_b) * 128 + wave * 32;
  const int rows_valid = min(max(p.T - t0w, 0), 32);
  const bool skip_init = p.flags & WAE_GLU_SKIP_INIT;
  const float* bias2 = bias_lds;
  constexpr int NPASS_SKIP = MT2 / 2;   // fp32 skip rows: 2 tiles (64 channels) per 256-byte pass
  for (int q2 = nq2_first; q2 < nq2; ++q2) {
    const int qi = nq1 + (q2 - nq2_first);
    unsigned long long sa = 0, sb = 0, sc = 0, sd = 0;
    if (p.stamps) { __builtin_amdgcn_sched_barrier(0); sa = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    // the only VMEM ops younger than DMA(qi) are the previous chunk's row stores (<= 8 per pass): a counted
    // wait retires the DMA without waiting for those stores to be acknowledged
    if (q2 == nq2_first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (NPASS_SKIP == 2 && (q2 - 1) * MT2 >= (p.Rp >> 5)) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.stamps) { __builtin_amdgcn_sched_barrier(0); sb = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    if (qi + 1 < nq_total && dbg_dma) dma_chunk(chunk_src(qi + 1), smem + ((qi + 1) & 1) * CHB, CHB, wave, lane);
    const char* buf = smem + (qi & 1) * CHB + lane * 16;
    const int gm0 = q2 * MT2;
    const bool is_out = gm0 < (p.Rp >> 5);
    const bool no_epi = p.flags & DBG_NO_EPI;
    // old rows of this chunk: issued now, consumed after the MFMAs
    f32x4 old[NPASS_SKIP][8];
    const int64_t roff = ((int64_t)b * p.T + t0w) * row_x + (int64_t)gm0 * 32 * ES;
    char* srow0 = (char*)(p.skip + ((int64_t)b * p.T + t0w) * p.Sp + 32 * (gm0 - (p.Rp >> 5)));
    const int64_t srs = (int64_t)p.Sp * 4;
    if (!no_epi && rows_valid > 0) {   // wave-uniform: a wave wholly past T touches no row
      if (is_out) {
        stage_load(old[0], p.x_in + roff, row_x, rows_valid, lane);
      } else if (!skip_init) {
#pragma unroll
        for (int hf = 0; hf < NPASS_SKIP; ++hf) stage_load(old[hf], srow0 + 256 * hf, srs, rows_valid, lane);
      }
    }
    f32x16 y[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) init_rows(y[mt], bias2 + 32 * (gm0 + mt), h);
    if (p.stamps) { __builtin_amdgcn_sched_barrier(0); sc = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
    gemm_chunk<MT2 * NKB, MT2, NKB, true>(buf, uf, y);
    if (p.stamps) { __builtin_amdgcn_sched_barrier(0); sd = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
      st_[8] += sb - sa; st_[9] += sc - sb; st_[10] += sd - sc; }
    if (no_epi) {
      if (y[0][0] == 12345.678f && tvalid) p.skip[t] = y[MT2 - 1][3];  // keep the MFMAs alive
      continue;
    }
    if (is_out) {
      // x' = (y + x) * sqrt(.5): rows of MT2*32 channels = 256 bytes in E
      stage_finish<E, MT2, true>(stg, y, old[0], p.x_out + roff, row_x, rows_valid, lane,
                                 [](const f32x4& v, const f32x4& o) {
                                   const float rs = 0.70710678118654752440f;
                                   f32x4 r = {(v.x + o.x) * rs, (v.y + o.y) * rs, (v.z + o.z) * rs, (v.w + o.w) * rs};
                                   return r;
                                 });
    } else {
#pragma unroll
      for (int hf = 0; hf < NPASS_SKIP; ++hf) {
        if (skip_init)
          stage_finish<float, 2, false>(stg, &y[2 * hf], old[hf], srow0 + 256 * hf, srs, rows_valid, lane,
                                        [](const f32x4& v, const f32x4&) { return v; });
        else
          stage_finish<float, 2, true>(stg, &y[2 * hf], old[hf], srow0 + 256 * hf, srs, rows_valid, lane,
                                       [](const f32x4& v, const f32x4& o) { f32x4 r = v + o; return r; });
      }
    }
    if (p.stamps) { __builtin_amdgcn_sched_barrier(0); st_[11] += __builtin_amdgcn_s_memtime() - sd; __builtin_amdgcn_sched_barrier(0); }
  }
  STAMP(4);
  if (p.stamps && threadIdx.x == 0) {
    st_[5] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) p.stamps[(size_t)blockIdx.x * 16 + i] = st_[i];
  }
}

template <typename E, int NP, bool EXACT>
static int launch_glu(const GluArgs& a, hipStream_t st) {
  constexpr int CHB = 2 * NP * 4 * 1024;
  const size_t lds = 2 * CHB + 4 * STG_BYTES + (size_t)(a.Rp + a.Sp) * 4;
  if (lds > 160 * 1024) {
    wae_set_error("glu_fwd: needs %zu bytes of LDS (> 160 KiB): Hp=%d with Rp+Sp=%d is not supported yet", lds, NP * 32,
                  a.Rp + a.Sp);
    return WAE_EUNSUPPORTED;
  }
  static size_t attr_done = 0;
  if (attr_done < lds) {
    if (hipFuncSetAttribute((const void*)glu_fwd_kernel<E, NP, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      wae_set_error("glu_fwd: cannot raise dynamic LDS to %zu", lds);
      return WAE_EHIP;
    }
    attr_done = lds;
  }
  const int tiles = (a.T + 127) / 128;
  hipLaunchKernelGGL((glu_fwd_kernel<E, NP, EXACT>), dim3(a.B * tiles), dim3(256), lds, st, a);
  return wae_check_launch("glu_fwd");
}

template <typename E, bool EXACT>
static int dispatch_np(int np, const GluArgs& a, hipStream_t st) {
  switch (np) {
    case 1: return launch_glu<E, 1, EXACT>(a, st);
    case 2: return launch_glu<E, 2, EXACT>(a, st);
    case 3: return launch_glu<E, 3, EXACT>(a, st);
    case 4: return launch_glu<E, 4, EXACT>(a, st);
    case 6: return launch_glu<E, 6, EXACT>(a, st);
    case 8: return launch_glu<E, 8, EXACT>(a, st);
    default:
      wae_set_error("glu_fwd: unsupported Hp=%d (Hp/32 must be 1,2,3,4,6 or 8)", np * 32);
      return WAE_EUNSUPPORTED;
  }
}

static int glu_validate(const wae_glu_desc* d) {
  WAE_REQUIRE(d != nullptr, "glu: null desc");
  WAE_REQUIRE(d->dtype == WAE_F32 || d->dtype == WAE_BF16, "glu: bad dtype %d", d->dtype);
  WAE_REQUIRE(d->B > 0 && d->T > 0, "glu: B,T must be positive");
  WAE_REQUIRE(d->Rp > 0 && d->Rp % 128 == 0 && d->Sp > 0 && d->Sp % 128 == 0 && d->Ccp >= 0 && d->Ccp % 64 == 0,
              "glu: Rp,Sp must be multiples of 128 and Ccp of 64 (got %d,%d,%d)", d->Rp, d->Sp, d->Ccp);
  WAE_REQUIRE(d->Hp > 0 && d->Hp % 32 == 0 && d->Hp <= 256, "glu: Hp must be a multiple of 32, <= 256");
  WAE_REQUIRE(d->ktaps >= 1 && d->dilation >= 1, "glu: ktaps, dilation must be >= 1");
  return WAE_OK;
}

extern "C" int64_t wae_glu_packed_bytes(const wae_glu_desc* d) {
  if (glu_validate(d) != WAE_OK) return WAE_EINVAL;
  const int ck = d->dtype == WAE_BF16 ? 64 : 32;
  const int mt2 = d->dtype == WAE_BF16 ? 4 : 2;
  const int64_t chb = (int64_t)2 * (d->Hp / 32) * 4 * 1024;
  const int64_t nq1 = (int64_t)d->ktaps * (d->Rp / ck) + d->Ccp / ck;
  const int64_t nq2 = ((d->Rp + d->Sp) / 32) / mt2;
  return (nq1 + nq2) * chb;
}

extern "C" int wae_glu_layer_fwd(const wae_glu_desc* d, const void* x_in, void* x_out, const void* c_up, float* skip,
                                 const float* zb, int64_t zb_stride, void* z_save, const void* w_packed,
                                 const float* bias2, void* stream) {
  int rc = glu_validate(d);
  if (rc != WAE_OK) return rc;
  WAE_REQUIRE(x_in && skip && zb && w_packed && bias2, "glu: null pointer argument");
  WAE_REQUIRE((d->flags & WAE_GLU_NO_OUT) || x_out, "glu: x_out is null but WAE_GLU_NO_OUT is not set");
  WAE_REQUIRE(d->Ccp == 0 || c_up, "glu: Ccp > 0 but c_up is null");
  WAE_REQUIRE(!(d->flags & WAE_GLU_SAVE_Z) || z_save, "glu: WAE_GLU_SAVE_Z without z_save");
  GluArgs a;
  a.x_in = (const char*)x_in; a.x_out = (char*)x_out; a.c_up = (const char*)c_up; a.skip = skip; a.zb = zb;
  a.z_save = (char*)z_save; a.w = (const char*)w_packed; a.bias2 = bias2; a.zb_stride = zb_stride;
  a.B = d->B; a.T = d->T; a.Rp = d->Rp; a.Sp = d->Sp; a.Ccp = d->Ccp; a.Hp = d->Hp; a.ktaps = d->ktaps;
  a.dilation = d->dilation; a.flags = d->flags; a.stamps = g_stamps;
  hipStream_t st = as_stream(stream);
  if (d->dtype == WAE_BF16) return dispatch_np<__bf16, false>(d->Hp / 32, a, st);
  return dispatch_np<float, true>(d->Hp / 32, a, st);
}
