// wae_head_fwd: decoder head + fused softmax cross-entropy (reference: wavenet.py:136-141,208-214;
// vqwae_train.py:363-379 with the one-step shift of :764).
//
//   h0 = relu(skip * sqrt(1/L))          (fp32 skip accumulators, converted to MFMA fragments on load)
//   h1 = relu(b1 + W1[Sp,Sp] . h0)       (GEMM 1; stays in registers as the operand of GEMM 2)
//   y  = b3 + W3[Op,Sp] . h1             (GEMM 2) -> logits (B,O,T) fp32 and/or nll[b,t] = lse(y) - y[target[t+1]]
//
// Same decomposition as glu_fwd.hip: 128 time steps per workgroup, one wave per 32 time columns, weights in
// A-fragment order through a double-buffered LDS ring, accumulator tiles reused as the next MFMA's B operand.
#include "wae_common.hpp"

struct HeadArgs {
  const float* skip;
  const char* w;
  const float* bias;  // [Sp | Op]
  float* logits;
  const int32_t* target;
  float* nll;
  char* h1_save;
  int B, T, Sp, Op, O;
  float scale;
};

template <typename E>
__device__ __forceinline__ typename ET<E>::frag load_skip_frag(const float* p, float scale);
template <>
__device__ __forceinline__ f32x4 load_skip_frag<float>(const float* p, float scale) {
  f32x4 v = *(const f32x4*)p;
  v.x = fmaxf(v.x * scale, 0.f); v.y = fmaxf(v.y * scale, 0.f); v.z = fmaxf(v.z * scale, 0.f); v.w = fmaxf(v.w * scale, 0.f);
  return v;
}
template <>
__device__ __forceinline__ bf16x8 load_skip_frag<__bf16>(const float* p, float scale) {
  const f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
  bf16x8 r;
  r[0] = (__bf16)fmaxf(a.x * scale, 0.f); r[1] = (__bf16)fmaxf(a.y * scale, 0.f);
  r[2] = (__bf16)fmaxf(a.z * scale, 0.f); r[3] = (__bf16)fmaxf(a.w * scale, 0.f);
  r[4] = (__bf16)fmaxf(b.x * scale, 0.f); r[5] = (__bf16)fmaxf(b.y * scale, 0.f);
  r[6] = (__bf16)fmaxf(b.z * scale, 0.f); r[7] = (__bf16)fmaxf(b.w * scale, 0.f);
  return r;
}

template <typename E, int NT>
__global__ void __launch_bounds__(256, 1) head_fwd_kernel(HeadArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  using vec4 = typename T_::vec4;
  constexpr int CHB = NT * 4 * 1024;
  constexpr int ES = sizeof(E);
  constexpr int KBU = T_::KBU;
  constexpr int NKB = NT * KBU;
  constexpr int MT2 = 4 / KBU;  // bf16: 2, f32: 1  (MT2 * NKB KiB == CHB)
  constexpr int EPL = T_::EPL;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int tiles_per_b = (p.T + 127) >> 7;
  const int b = blockIdx.x / tiles_per_b;
  const int t = (blockIdx.x % tiles_per_b) * 128 + wave * 32 + n;
  const bool tvalid = t < p.T;

  const int nq1 = p.Sp / T_::CK;
  const int nq2 = (p.Op >> 5) / MT2;
  const int nq_total = nq1 + nq2;
  const float* srow = p.skip + ((int64_t)b * p.T + (tvalid ? t : 0)) * p.Sp;

  frag Bn[4], Bc[4];
  auto load_B = [&](int q, frag (&Bf)[4]) {
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      if (tvalid) {
        Bf[blk] = load_skip_frag<E>(srow + q * T_::CK + blk * 2 * EPL + h * EPL, p.scale);
      } else {
        frag zf = {};
        Bf[blk] = zf;
      }
    }
  };

  f32x16 acc[NT];
#pragma unroll
  for (int m = 0; m < NT; ++m)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *(const f32x4*)(p.bias + 32 * m + 8 * g + 4 * h);
      acc[m][4 * g + 0] = v.x; acc[m][4 * g + 1] = v.y; acc[m][4 * g + 2] = v.z; acc[m][4 * g + 3] = v.w;
    }

  dma_chunk(p.w, smem, CHB, wave, lane);
  load_B(0, Bn);
  for (int q = 0; q < nq1; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
    if (q + 1 < nq_total) dma_chunk(p.w + (int64_t)(q + 1) * CHB, smem + ((q + 1) & 1) * CHB, CHB, wave, lane);
    if (q + 1 < nq1) load_B(q + 1, Bn);
    const char* buf = smem + (q & 1) * CHB + lane * 16;
    gemm_chunk<4 * NT, NT, 4>(buf, Bc, acc);
  }

  // relu -> operand fragments (and optional save for backward)
  frag uf[NKB];
#pragma unroll
  for (int m = 0; m < NT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = fmaxf(acc[m][r], 0.f);
    if (p.h1_save && tvalid) {
      char* hr = p.h1_save + (((int64_t)b * p.T + t) * p.Sp + 32 * m + 4 * h) * ES;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = {acc[m][4 * g], acc[m][4 * g + 1], acc[m][4 * g + 2], acc[m][4 * g + 3]};
        *(vec4*)(hr + 8 * g * ES) = from_f32x4<E>(v);
      }
    }
    frag tmp[KBU];
    acc_to_frags(acc[m], tmp);
#pragma unroll
    for (int s = 0; s < KBU; ++s) uf[m * KBU + s] = tmp[s];
  }

  // GEMM 2 + logits store + online log-sum-exp
  const bool want_ce = p.target != nullptr && p.nll != nullptr;
  int tgt = -1;
  if (want_ce && tvalid && t + 1 < p.T) tgt = p.target[(int64_t)b * p.T + t + 1];
  float run_m = -INFINITY, run_s = 0.f, picked = 0.f;
  const float* b3 = p.bias + p.Sp;
  for (int q2 = 0; q2 < nq2; ++q2) {
    const int qi = nq1 + q2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (qi + 1 < nq_total) dma_chunk(p.w + (int64_t)(qi + 1) * CHB, smem + ((qi + 1) & 1) * CHB, CHB, wave, lane);
    const char* buf = smem + (qi & 1) * CHB + lane * 16;
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
      const int gm = q2 * MT2 + mt;
      f32x16 y;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 v = *(const f32x4*)(b3 + 32 * gm + 8 * g + 4 * h);
        y[4 * g + 0] = v.x; y[4 * g + 1] = v.y; y[4 * g + 2] = v.z; y[4 * g + 3] = v.w;
      }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const frag a = *(const frag*)(buf + (mt * NKB + kb) * 1024);
        mma32(y, a, uf[kb]);
      }
      if (tvalid) {
        float tile_m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cls = 32 * gm + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (cls < p.O) {
            if (p.logits) p.logits[((int64_t)b * p.O + cls) * p.T + t] = y[r];
            tile_m = fmaxf(tile_m, y[r]);
            if (cls == tgt) picked = y[r];
          }
        }
        if (want_ce && tile_m > -INFINITY) {
          const float nm = fmaxf(run_m, tile_m);
          float s = run_s * __expf(run_m - nm);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cls = 32 * gm + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cls < p.O) s += __expf(y[r] - nm);
          }
          run_m = nm;
          run_s = s;
        }
      }
    }
  }
  if (want_ce) {
    // merge the two lane halves (rows 4h..4h+3 of every 8) of the same time column
    const float om = __shfl_xor(run_m, 32), os = __shfl_xor(run_s, 32), op = __shfl_xor(picked, 32);
    const float nm = fmaxf(run_m, om);
    float s = 0.f;
    if (run_m > -INFINITY) s += run_s * __expf(run_m - nm);
    if (om > -INFINITY) s += os * __expf(om - nm);
    if (tvalid && h == 0) {
      float v = 0.f;
      if (tgt >= 0) v = (nm + __logf(s)) - (picked + op);
      p.nll[(int64_t)b * p.T + t] = v;
    }
  }
}

template <typename E, int NT>
static int launch_head(const HeadArgs& a, hipStream_t st) {
  constexpr int CHB = NT * 4 * 1024;
  const size_t lds = 2 * CHB;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)head_fwd_kernel<E, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      wae_set_error("head_fwd: cannot raise dynamic LDS to %zu", lds);
      return WAE_EHIP;
    }
    attr_done = true;
  }
  const int tiles = (a.T + 127) / 128;
  hipLaunchKernelGGL((head_fwd_kernel<E, NT>), dim3(a.B * tiles), dim3(256), lds, st, a);
  return wae_check_launch("head_fwd");
}

static int head_validate(const wae_head_desc* d) {
  WAE_REQUIRE(d != nullptr, "head: null desc");
  WAE_REQUIRE(d->dtype == WAE_F32 || d->dtype == WAE_BF16, "head: bad dtype %d", d->dtype);
  WAE_REQUIRE(d->B > 0 && d->T > 0 && d->O > 0, "head: B,T,O must be positive");
  WAE_REQUIRE(d->Sp % 128 == 0 && (d->Sp == 128 || d->Sp == 256), "head: Sp must be 128 or 256 (got %d)", d->Sp);
  WAE_REQUIRE(d->Op % 128 == 0 && d->Op >= d->O, "head: Op must be a multiple of 128 and >= O");
  return WAE_OK;
}

extern "C" int64_t wae_head_packed_bytes(const wae_head_desc* d) {
  if (head_validate(d) != WAE_OK) return WAE_EINVAL;
  const int ck = d->dtype == WAE_BF16 ? 64 : 32;
  const int mt2 = d->dtype == WAE_BF16 ? 2 : 1;
  const int64_t chb = (int64_t)(d->Sp / 32) * 4 * 1024;
  return (int64_t)(d->Sp / ck + (d->Op / 32) / mt2) * chb;
}

extern "C" int wae_head_fwd(const wae_head_desc* d, const float* skip, const void* w_packed, const float* bias,
                            float* logits, const int32_t* target, float* nll, void* h1_save, void* stream) {
  int rc = head_validate(d);
  if (rc != WAE_OK) return rc;
  WAE_REQUIRE(skip && w_packed && bias, "head: null pointer argument");
  WAE_REQUIRE(logits || (target && nll), "head: nothing to produce (logits and nll both null)");
  HeadArgs a;
  a.skip = skip; a.w = (const char*)w_packed; a.bias = bias; a.logits = logits; a.target = target; a.nll = nll;
  a.h1_save = (char*)h1_save; a.B = d->B; a.T = d->T; a.Sp = d->Sp; a.Op = d->Op; a.O = d->O; a.scale = d->scale;
  hipStream_t st = as_stream(stream);
  const int nt = d->Sp / 32;
  if (d->dtype == WAE_BF16) return nt == 4 ? launch_head<__bf16, 4>(a, st) : launch_head<__bf16, 8>(a, st);
  return nt == 4 ? launch_head<float, 4>(a, st) : launch_head<float, 8>(a, st);
}
