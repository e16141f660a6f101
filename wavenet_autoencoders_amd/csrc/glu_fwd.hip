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
};

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

  frag Bn[4], Bc[4];
  auto load_B = [&](int q, frag (&Bf)[4]) {
    const char* src;
    bool ok = tvalid;
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
    for (int m = 0; m < NM; ++m) {
      const int row0 = (m < NP ? 32 * m : p.Hp + 32 * (m - NP)) + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(zbb + row0 + 8 * g);
        acc[m][4 * g + 0] = v.x; acc[m][4 * g + 1] = v.y; acc[m][4 * g + 2] = v.z; acc[m][4 * g + 3] = v.w;
      }
    }
  }

  dma_chunk(chunk_src(0), smem, CHB, wave, lane);
  load_B(0, Bn);

  // ---- GEMM 1 ----------------------------------------------------------------------------------------
  for (int q = 0; q < nq1; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
    if (q + 1 < nq_total) dma_chunk(chunk_src(q + 1), smem + ((q + 1) & 1) * CHB, CHB, wave, lane);
    if (q + 1 < nq1) load_B(q + 1, Bn);
    const char* buf = smem + (q & 1) * CHB + lane * 16;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const frag a = *(const frag*)(buf + (blk * NM + m) * 1024);
        mma32(acc[m], a, Bc[blk]);
      }
    }
  }

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
      if constexpr (EXACT) {
        u[r] = tanhf(a) * (1.0f / (1.0f + expf(-g)));
      } else {
        const float ac = fminf(fmaxf(a, -15.0f), 15.0f);
        const float ea = __expf(-2.0f * ac);
        const float eg = __expf(-g);
        u[r] = (1.0f - ea) * fast_rcp((1.0f + ea) * (1.0f + eg));
      }
    }
    frag tmp[KBU];
    acc_to_frags(u, tmp);
#pragma unroll
    for (int s = 0; s < KBU; ++s) uf[pr * KBU + s] = tmp[s];
  }

  // ---- GEMM 2 + epilogues ------------------------------------------------------------------------------
  const float rs = 0.70710678118654752440f;
  const bool skip_init = p.flags & WAE_GLU_SKIP_INIT;
  for (int q2 = nq2_first; q2 < nq2; ++q2) {
    const int qi = nq1 + (q2 - nq2_first);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (qi + 1 < nq_total) dma_chunk(chunk_src(qi + 1), smem + ((qi + 1) & 1) * CHB, CHB, wave, lane);
    const char* buf = smem + (qi & 1) * CHB + lane * 16;
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
      const int gm = q2 * MT2 + mt;
      f32x16 y;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(p.bias2 + 32 * gm + 8 * g + 4 * h);
        y[4 * g + 0] = v.x; y[4 * g + 1] = v.y; y[4 * g + 2] = v.z; y[4 * g + 3] = v.w;
      }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const frag a = *(const frag*)(buf + (mt * NKB + kb) * 1024);
        mma32(y, a, uf[kb]);
      }
      if (tvalid) {
        if (gm < (p.Rp >> 5)) {
          const int64_t off = ((int64_t)b * p.T + t) * row_x + (32 * gm + 4 * h) * ES;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 res = to_f32x4(*(const vec4*)(p.x_in + off + 8 * g * ES));
            f32x4 o;
            o.x = (y[4 * g + 0] + res.x) * rs; o.y = (y[4 * g + 1] + res.y) * rs;
            o.z = (y[4 * g + 2] + res.z) * rs; o.w = (y[4 * g + 3] + res.w) * rs;
            *(vec4*)(p.x_out + off + 8 * g * ES) = from_f32x4<E>(o);
          }
        } else {
          float* sp = p.skip + ((int64_t)b * p.T + t) * p.Sp + 32 * (gm - (p.Rp >> 5)) + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 o = {y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
            if (!skip_init) {
              const f32x4 old = *(const f32x4*)(sp + 8 * g);
              o += old;
            }
            *(f32x4*)(sp + 8 * g) = o;
          }
        }
      }
    }
  }
}

template <typename E, int NP, bool EXACT>
static int launch_glu(const GluArgs& a, hipStream_t st) {
  constexpr int CHB = 2 * NP * 4 * 1024;
  const size_t lds = 2 * CHB;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)glu_fwd_kernel<E, NP, EXACT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds) != hipSuccess) {
      wae_set_error("glu_fwd: cannot raise dynamic LDS to %zu", lds);
      return WAE_EHIP;
    }
    attr_done = true;
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
  a.dilation = d->dilation; a.flags = d->flags;
  hipStream_t st = as_stream(stream);
  if (d->dtype == WAE_BF16) return dispatch_np<__bf16, false>(d->Hp / 32, a, st);
  return dispatch_np<float, true>(d->Hp / 32, a, st);
}
