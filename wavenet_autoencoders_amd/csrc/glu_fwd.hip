// wae_glu_layer_fwd: one ResidualConv1dGLU layer, fused (reference: modules.py:115-163).
//
//   z[2Hp, t] = zb + W1[2Hp, k*Rp + Ccp] . [x[t-(k-1)d] ; ... ; x[t] ; c[t]]      (GEMM 1, MFMA)
//   u[Hp, t]  = tanh(z_a) * sigmoid(z_b)                                           (registers) -> stored (B,T,*)
//   y[Rp, t]  = b_out + W_out[Rp, Hp] . u                                          (GEMM 2, MFMA; u never leaves
//   x'[t]     = (y + x[t]) * sqrt(.5)                                               the register file)
//
// The skip path (conv1x1_skip, modules.py:157, and `skips += h`, wavenet.py:204-207) is NOT accumulated here:
// every layer stores its gated activation u_l once, and the head kernel contracts all of them in ONE GEMM
//   skips = sum_l (W_skip_l . u_l + b_skip_l) = [W_skip_0 ... W_skip_{L-1}] . [u_0 ; ... ; u_{L-1}] + sum_l b_skip_l
// -- same arithmetic (fp32 accumulation over K = L*Hp inside the MFMA) without L read-modify-write passes over
// a (B,T,S) fp32 buffer: per sample and layer the kernel reads R+Cc and writes R+H elements instead of
// reading R+Cc+2S and writing R+2S.
//
// Work decomposition: one workgroup = 128 consecutive time steps of one clip, 4 waves (one per SIMD), each
// wave owns 32 time columns and ALL channels, so the gate and the second GEMM need no cross-wave exchange:
// the 32x32 accumulator tiles of GEMM 1 (column = time on the lane, rows = channels in the registers)
// are converted in place into the B operand of GEMM 2 (cdna_hip_programming.md section 3, "An accumulator
// tile as the next MFMA's operand").  Weights arrive pre-packed in A-fragment order and are streamed through
// a double-buffered LDS ring by LDS-DMA, shared by the 4 waves; the activation (B) operand is read straight
// from HBM/L2 as 16-byte fragments (time-major rows, channels innermost), zero-filled before t=0 (causal pad).
// Outputs leave through a wave-private swizzled LDS tile so that every global store is a full-row 16-B access.
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
  char* u_out;
  const float* zb;
  char* z_save;
  const char* w;
  const float* bias_out;
  int64_t zb_stride;
  int64_t u_stride;  // elements per time row of u_out
  int B, T, Rp, Ccp, Hp, ktaps, dilation, flags;
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

template <typename E, int NP, bool EXACT>
__global__ void __launch_bounds__(256, 1) glu_fwd_kernel(GluArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
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
  const int t0w = (blockIdx.x % tiles_per_b) * 128 + wave * 32;
  const int t = t0w + n;
  const bool tvalid = t < p.T;
  const int rows_valid = min(max(p.T - t0w, 0), 32);

  const int cpr = p.Rp / T_::CK;  // chunks per tap
  const int nq_conv = p.ktaps * cpr;
  const int nq1 = nq_conv + p.Ccp / T_::CK;
  const int nq2 = (p.flags & WAE_GLU_NO_OUT) ? 0 : (p.Rp >> 5) / MT2;
  const int nq_total = nq1 + nq2;

  const int64_t row_x = (int64_t)p.Rp * ES;
  const int64_t row_c = (int64_t)p.Ccp * ES;
  const char* xb = p.x_in + (int64_t)b * p.T * row_x;
  const char* cb = p.c_up ? p.c_up + (int64_t)b * p.T * row_c : nullptr;

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

  // out bias -> LDS once (read back per chunk with ds_read: keeps the second GEMM's accumulator init off vmcnt,
  // where it would drain the LDS-DMA and residual-prefetch queues)
  float* bias_lds = (float*)(smem + 2 * CHB + 4 * STG_BYTES);
  if (nq2 > 0)
    for (int i = threadIdx.x * 4; i < p.Rp; i += 1024) *(f32x4*)(bias_lds + i) = *(const f32x4*)(p.bias_out + i);
  if (dbg_dma) dma_chunk(p.w, smem, CHB, wave, lane);
  load_B(0, Bn);
  STAMP(1);

  // ---- GEMM 1 ----------------------------------------------------------------------------------------
  for (int q = 0; q < nq1; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
    if (q + 1 < nq_total && dbg_dma) dma_chunk(p.w + (int64_t)(q + 1) * CHB, smem + ((q + 1) & 1) * CHB, CHB, wave, lane);
    if (q + 1 < nq1) load_B(q + 1, Bn);
    const char* buf = smem + (q & 1) * CHB + lane * 16;
    gemm_chunk<4 * NM, NM, 4>(buf, Bc, acc);
  }
  STAMP(2);

  char* stg = smem + 2 * CHB + wave * STG_BYTES;
  const bool no_epi = p.flags & DBG_NO_EPI;

  // ---- optional z save (training): rows of 2Hp elements, a-half then b-half ------------------------------
  if ((p.flags & WAE_GLU_SAVE_Z) && rows_valid > 0) {
    char* zr = p.z_save + ((int64_t)b * p.T + t0w) * (2 * p.Hp) * ES;
    stage_store_tiles<E, NP>(stg, &acc[0], zr, (int64_t)2 * p.Hp * ES, rows_valid, lane);
    stage_store_tiles<E, NP>(stg, &acc[NP], zr + (int64_t)p.Hp * ES, (int64_t)2 * p.Hp * ES, rows_valid, lane);
  }

  // ---- gate: u = tanh(a) * sigmoid(b); stored once for the head's skip GEMM, and converted in place to the
  //      operand fragments of GEMM 2 -----------------------------------------------------------------------
  frag uf[NKB];
#pragma unroll
  for (int pr = 0; pr < NP; ++pr) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float a = acc[pr][r], g = acc[NP + pr][r];
      float u;
      if (p.flags & DBG_NO_GATE) {
        u = a * g;
      } else if constexpr (EXACT) {
        u = tanhf(a) * (1.0f / (1.0f + expf(-g)));
      } else {
        // tanh(a)*sigmoid(g) = (1-ea) / ((1+ea)(1+eg)), ea = e^-2a, eg = e^-g.  a is clamped from below so that
        // ea stays finite (tanh(-15) == -1 in fp32); eg = inf gives rcp(inf) = 0, the correct limit.
        const float ea = __builtin_amdgcn_exp2f(fmaxf(a, -15.0f) * -2.885390081777927f);
        const float eg = __builtin_amdgcn_exp2f(g * -1.4426950408889634f);
        u = (1.0f - ea) * fast_rcp((1.0f + ea) * (1.0f + eg));
      }
      acc[pr][r] = u;
    }
    frag tmp[KBU];
    acc_to_frags(acc[pr], tmp);
#pragma unroll
    for (int s = 0; s < KBU; ++s) uf[pr * KBU + s] = tmp[s];
  }
  if (!no_epi && rows_valid > 0) {
    char* ur = p.u_out + ((int64_t)b * p.T + t0w) * p.u_stride * ES;
    stage_store_tiles<E, NP>(stg, &acc[0], ur, p.u_stride * ES, rows_valid, lane);
  }
  STAMP(3);

  // ---- GEMM 2 + residual epilogue ------------------------------------------------------------------------
  // Each chunk = MT2 M-tiles (MT2*32 output channels = 256 bytes per time row) against all of u.
  for (int q2 = 0; q2 < nq2; ++q2) {
    const int qi = nq1 + q2;
    // the only VMEM ops younger than DMA(qi) are the previous epilogue's row stores: a counted wait retires
    // the DMA without waiting for those stores to be acknowledged (8 stores per 256-byte pass)
    if (q2 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (qi + 1 < nq_total && dbg_dma) dma_chunk(p.w + (int64_t)(qi + 1) * CHB, smem + ((qi + 1) & 1) * CHB, CHB, wave, lane);
    const char* buf = smem + (qi & 1) * CHB + lane * 16;
    const int gm0 = q2 * MT2;
    // residual x[t] for this chunk's channels, as operand-shaped 16-byte fragments (L2 hits: tap k-1 of GEMM 1
    // read the same bytes); issued now, consumed after the MFMAs
    constexpr int NRES = ES == 2 ? 2 * MT2 : 4 * MT2;  // 16-byte fragments per lane (32 bytes of the row per pair)
    frag res[NRES];
    {
      const char* rsrc = xb + (int64_t)(tvalid ? t : 0) * row_x + (int64_t)gm0 * 32 * ES + h * 16;
#pragma unroll
      for (int f = 0; f < NRES; ++f) res[f] = *(const frag*)(rsrc + f * 32);
    }
    f32x16 y[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) init_rows(y[mt], bias_lds + 32 * (gm0 + mt), h);
    gemm_chunk<MT2 * NKB, MT2, NKB, true>(buf, uf, y);
    if (no_epi) {
      if (y[0][0] == 12345.678f && tvalid) p.x_out[t] = (char)y[MT2 - 1][3];  // keep the MFMAs alive
      continue;
    }
    // x' = (y + x) * sqrt(.5) in the accumulator layout
    const float rs = 0.70710678118654752440f;
    residual_to_acc_layout(res);
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 r4 = residual_piece<E>(res, mt, g);
        y[mt][4 * g + 0] = (y[mt][4 * g + 0] + r4.x) * rs;
        y[mt][4 * g + 1] = (y[mt][4 * g + 1] + r4.y) * rs;
        y[mt][4 * g + 2] = (y[mt][4 * g + 2] + r4.z) * rs;
        y[mt][4 * g + 3] = (y[mt][4 * g + 3] + r4.w) * rs;
      }
    }
    if (rows_valid > 0) {
      char* orow = p.x_out + ((int64_t)b * p.T + t0w) * row_x + (int64_t)gm0 * 32 * ES;
      stage_store_tiles<E, MT2>(stg, y, orow, row_x, rows_valid, lane);
    }
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
  const size_t lds = 2 * CHB + 4 * STG_BYTES + (size_t)a.Rp * 4;
  if (lds > 160 * 1024) {
    wae_set_error("glu_fwd: needs %zu bytes of LDS (> 160 KiB): Hp=%d with Rp=%d is not supported yet", lds, NP * 32, a.Rp);
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
    default:
      wae_set_error("glu_fwd: unsupported Hp=%d (Hp/32 must be 1,2,3,4 or 6)", np * 32);
      return WAE_EUNSUPPORTED;
  }
}

static int glu_validate(const wae_glu_desc* d) {
  WAE_REQUIRE(d != nullptr, "glu: null desc");
  WAE_REQUIRE(d->dtype == WAE_F32 || d->dtype == WAE_BF16, "glu: bad dtype %d", d->dtype);
  WAE_REQUIRE(d->B > 0 && d->T > 0, "glu: B,T must be positive");
  WAE_REQUIRE(d->Rp > 0 && d->Rp % 128 == 0 && d->Ccp >= 0 && d->Ccp % 64 == 0,
              "glu: Rp must be a multiple of 128 and Ccp of 64 (got %d,%d)", d->Rp, d->Ccp);
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
  const int64_t nq2 = (d->Rp / 32) / mt2;
  return (nq1 + nq2) * chb;
}

extern "C" int wae_glu_layer_fwd(const wae_glu_desc* d, const void* x_in, void* x_out, const void* c_up, void* u_out,
                                 int64_t u_stride, const float* zb, int64_t zb_stride, void* z_save, const void* w_packed,
                                 const float* bias_out, void* stream) {
  int rc = glu_validate(d);
  if (rc != WAE_OK) return rc;
  WAE_REQUIRE(x_in && u_out && zb && w_packed, "glu: null pointer argument");
  WAE_REQUIRE(u_stride >= d->Hp, "glu: u_stride (%lld) < Hp", (long long)u_stride);
  WAE_REQUIRE((d->flags & WAE_GLU_NO_OUT) || (x_out && bias_out), "glu: x_out/bias_out null but WAE_GLU_NO_OUT is not set");
  WAE_REQUIRE(d->Ccp == 0 || c_up, "glu: Ccp > 0 but c_up is null");
  WAE_REQUIRE(!(d->flags & WAE_GLU_SAVE_Z) || z_save, "glu: WAE_GLU_SAVE_Z without z_save");
  GluArgs a;
  a.x_in = (const char*)x_in; a.x_out = (char*)x_out; a.c_up = (const char*)c_up; a.u_out = (char*)u_out; a.zb = zb;
  a.z_save = (char*)z_save; a.w = (const char*)w_packed; a.bias_out = bias_out; a.zb_stride = zb_stride;
  a.u_stride = u_stride; a.B = d->B; a.T = d->T; a.Rp = d->Rp; a.Ccp = d->Ccp; a.Hp = d->Hp; a.ktaps = d->ktaps;
  a.dilation = d->dilation; a.flags = d->flags; a.stamps = g_stamps;
  hipStream_t st = as_stream(stream);
  if (d->dtype == WAE_BF16) return dispatch_np<__bf16, false>(d->Hp / 32, a, st);
  return dispatch_np<float, true>(d->Hp / 32, a, st);
}
