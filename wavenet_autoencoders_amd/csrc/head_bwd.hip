// wae_head_bwd: backward of the head + fused softmax cross-entropy down to dskip (autograd of wavenet.py:208-214 and
// vqwae_train.py:363-379,:764).
//
//   y   = b3 + W3 h1                      (recomputed; logits were never stored)        GEMM a
//   dy  = (softmax(y) - onehot(target[t+1])) * w[t] ,  w[t] = [t < len-1] / sum(mask)   (or an external dy: DMoL)
//   dh1 = (W3^T dy) * [h1 > 0]                                                          GEMM b
//   dh0 = (W1^T dh1) * [h0 > 0] ;  dskip = dh0 * sqrt(1/L)                              GEMM c
// dy, dh1 and dskip are stored time-major for the weight-gradient contractions (wae_gemm_tn).  Same skeleton as
// head_fwd.hip: accumulator tiles feed the next GEMM as MFMA B operands, weights stream through an LDS ring.
#include "wae_common.hpp"

struct HeadBwdArgs {
  const char* h0;
  const char* h1;
  const char* w;        // [W3 first-order | W3^T second-order | W1^T second-order]
  const float* b3;      // Op
  const float* lse;     // (B,T) log-sum-exp from the forward (nll + picked)
  const int32_t* target;
  const int32_t* lengths;
  const char* ext_dy;   // (B,T,Op) dtype or null
  char* dy_out;         // (B,T,Op)
  char* dh1_out;        // (B,T,Sp)
  char* dskip_out;      // (B,T,Sp)
  int B, T, Sp, Op, O;
  float scale, inv_count;
};

template <typename E, int NTO, int NTS>
__global__ void __launch_bounds__(256, 1) head_bwd_kernel(HeadBwdArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  constexpr int ES = sizeof(E);
  constexpr int KBU = T_::KBU;
  constexpr int MT2 = 4 / KBU;
  constexpr int CHA = NTO * 4 * 1024;            // chunk bytes of GEMM a and b
  constexpr int CHC = NTS * 4 * 1024;            // chunk bytes of GEMM c
  constexpr int CHMAX = CHA > CHC ? CHA : CHC;
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
  char* stg = smem + 2 * CHMAX + wave * STG_BYTES;
  float* bias_lds = (float*)(smem + 2 * CHMAX + 4 * STG_BYTES);
  const int64_t rowS = (int64_t)p.Sp * ES, rowO = (int64_t)p.Op * ES;
  const int64_t base_t = (int64_t)b * p.T + t0w;

  const int nqa = p.ext_dy ? 0 : p.Sp / T_::CK;
  const int nqb = NTS / MT2;
  const int nqc = NTS / MT2;
  // byte offsets of the three streams
  const int64_t offB = (int64_t)(p.Sp / T_::CK) * CHA;
  const int64_t offC = offB + (int64_t)nqb * CHA;
  int buf_i = 0;
  auto ring = [&](int i) { return smem + (i & 1) * CHMAX; };

  for (int i = threadIdx.x * 4; i < p.Op; i += 1024) *(f32x4*)(bias_lds + i) = *(const f32x4*)(p.b3 + i);

  // ---- dy tiles ----------------------------------------------------------------------------------------------
  f32x16 y[NTO];
  if (p.ext_dy) {
    dma_chunk(p.w + offB, ring(0), CHA, wave, lane);
    if (rows_valid > 0) stage_load_tiles<E, NTO>(stg, y, p.ext_dy + base_t * rowO, rowO, rows_valid, lane);
  } else {
    frag Bn[4], Bc[4];
    const char* hrow = p.h1 + ((int64_t)b * p.T + (tvalid ? t : 0)) * rowS + h * 16;
    auto load_B = [&](int q, frag (&Bf)[4]) {
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        if (tvalid) {
          Bf[blk] = *(const frag*)(hrow + q * 128 + blk * 32);
        } else {
          frag zf = {};
          Bf[blk] = zf;
        }
      }
    };
    dma_chunk(p.w, ring(0), CHA, wave, lane);
    load_B(0, Bn);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int m = 0; m < NTO; ++m) init_rows(y[m], bias_lds + 32 * m, h);
    for (int q = 0; q < nqa; ++q) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
      // the next chunk is either GEMM a's or the first of GEMM b (contiguous in the stream)
      dma_chunk(p.w + (int64_t)(q + 1) * CHA, ring(q + 1), CHA, wave, lane);
      if (q + 1 < nqa) load_B(q + 1, Bn);
      gemm_chunk<4 * NTO, NTO, 4>(ring(q) + lane * 16, Bc, y);
    }
    buf_i = nqa;
    // softmax gradient in registers
    const float lse = tvalid ? p.lse[(int64_t)b * p.T + t] : 0.f;
    int tgt = -1;
    float wt = 0.f;
    if (tvalid && t + 1 < p.T) {
      const int len = p.lengths ? min(p.lengths[b], p.T) : p.T;
      if (t < len - 1) { wt = p.inv_count; tgt = p.target[(int64_t)b * p.T + t + 1]; }
    }
#pragma unroll
    for (int m = 0; m < NTO; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cls = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
        float g = 0.f;
        if (cls < p.O) g = (expf(y[m][r] - lse) - (cls == tgt ? 1.f : 0.f)) * wt;
        y[m][r] = g;
      }
  }
  if (p.dy_out && !p.ext_dy && rows_valid > 0) stage_store_tiles<E, NTO>(stg, y, p.dy_out + base_t * rowO, rowO, rows_valid, lane);
  constexpr int NKBO = NTO * KBU;
  frag df[NKBO];
#pragma unroll
  for (int m = 0; m < NTO; ++m) {
    frag tmp[KBU];
    acc_to_frags(y[m], tmp);
#pragma unroll
    for (int s = 0; s < KBU; ++s) df[m * KBU + s] = tmp[s];
  }

  // ---- GEMM b: dh1 = W3^T dy, masked by h1 > 0 -------------------------------------------------------------------
  f32x16 acc[NTS];
#pragma unroll
  for (int qb = 0; qb < nqb; ++qb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int64_t nxt = qb + 1 < nqb ? offB + (int64_t)(qb + 1) * CHA : offC;
    dma_chunk(p.w + nxt, ring(buf_i + qb + 1), qb + 1 < nqb ? CHA : CHC, wave, lane);
    f32x16(&yy)[MT2] = *reinterpret_cast<f32x16(*)[MT2]>(&acc[qb * MT2]);
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) yy[mt][r] = 0.f;
    gemm_chunk<MT2 * NKBO, MT2, NKBO, true>(ring(buf_i + qb) + lane * 16, df, yy);
  }
  buf_i += nqb;
  {
    f32x16 hv[NTS];
    if (rows_valid > 0) stage_load_tiles<E, NTS>(stg, hv, p.h1 + base_t * rowS, rowS, rows_valid, lane);
#pragma unroll
    for (int m = 0; m < NTS; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = hv[m][r] > 0.f ? acc[m][r] : 0.f;
  }
  if (rows_valid > 0) stage_store_tiles<E, NTS>(stg, acc, p.dh1_out + base_t * rowS, rowS, rows_valid, lane);
  constexpr int NKBS = NTS * KBU;
  frag hf[NKBS];
#pragma unroll
  for (int m = 0; m < NTS; ++m) {
    frag tmp[KBU];
    acc_to_frags(acc[m], tmp);
#pragma unroll
    for (int s = 0; s < KBU; ++s) hf[m * KBU + s] = tmp[s];
  }

  // ---- GEMM c: dh0 = W1^T dh1, masked by h0 > 0, times sqrt(1/L) ---------------------------------------------------
#pragma unroll
  for (int qc = 0; qc < nqc; ++qc) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (qc + 1 < nqc) dma_chunk(p.w + offC + (int64_t)(qc + 1) * CHC, ring(buf_i + qc + 1), CHC, wave, lane);
    f32x16(&yy)[MT2] = *reinterpret_cast<f32x16(*)[MT2]>(&acc[qc * MT2]);
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) yy[mt][r] = 0.f;
    gemm_chunk<MT2 * NKBS, MT2, NKBS, true>(ring(buf_i + qc) + lane * 16, hf, yy);
  }
  if (rows_valid > 0) {
    f32x16 hv[NTS];
    stage_load_tiles<E, NTS>(stg, hv, p.h0 + base_t * rowS, rowS, rows_valid, lane);
#pragma unroll
    for (int m = 0; m < NTS; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = hv[m][r] > 0.f ? acc[m][r] * p.scale : 0.f;
    stage_store_tiles<E, NTS>(stg, acc, p.dskip_out + base_t * rowS, rowS, rows_valid, lane);
  }
}

template <typename E, int NTO, int NTS>
static int launch_head_bwd(const HeadBwdArgs& a, hipStream_t st) {
  constexpr int CHMAX = (NTO > NTS ? NTO : NTS) * 4 * 1024;
  const size_t lds = 2 * CHMAX + 4 * STG_BYTES + (size_t)a.Op * 4;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)head_bwd_kernel<E, NTO, NTS>, lds_cache, lds, "head_bwd"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 127) / 128;
  hipLaunchKernelGGL((head_bwd_kernel<E, NTO, NTS>), dim3(a.B * tiles), dim3(256), lds, st, a);
  return wae_check_launch("head_bwd");
}

template <typename E>
static int dispatch_head_bwd(const HeadBwdArgs& a, hipStream_t st) {
  const int nto = a.Op / 32, nts = a.Sp / 32;
  if (nto == 4 && nts == 4) return launch_head_bwd<E, 4, 4>(a, st);
  if (nto == 8 && nts == 4) return launch_head_bwd<E, 8, 4>(a, st);
  if (nto == 4 && nts == 8) return launch_head_bwd<E, 4, 8>(a, st);
  if (nto == 8 && nts == 8) return launch_head_bwd<E, 8, 8>(a, st);
  wae_set_error("head_bwd: Op (%d) and Sp (%d) must each be 128 or 256", a.Op, a.Sp);
  return WAE_EUNSUPPORTED;
}

extern "C" int64_t wae_head_bwd_packed_bytes(const wae_head_desc* d) {
  if (!d) return WAE_EINVAL;
  const int ck = wae_is16(d->dtype) ? 64 : 32;
  const int mt2 = wae_is16(d->dtype) ? 2 : 1;
  const int64_t cha = (int64_t)(d->Op / 32) * 4 * 1024, chc = (int64_t)(d->Sp / 32) * 4 * 1024;
  return (int64_t)(d->Sp / ck) * cha + (int64_t)((d->Sp / 32) / mt2) * (cha + chc);
}

extern "C" int wae_head_bwd(const wae_head_desc* d, const void* h0, const void* h1, const void* w_packed, const float* b3,
                            const float* lse, const int32_t* target, const int32_t* lengths, float inv_count, const void* ext_dy,
                            void* dy_out, void* dh1_out, void* dskip_out, void* stream) {
  WAE_REQUIRE(d && h0 && h1 && w_packed && dh1_out && dskip_out, "head_bwd: null pointer argument");
  WAE_REQUIRE(ext_dy || (b3 && lse && target && dy_out), "head_bwd: CE mode needs b3, lse, target, dy_out");
  WAE_REQUIRE(wae_dtype_ok(d->dtype), "head_bwd: bad dtype");
  HeadBwdArgs a;
  a.h0 = (const char*)h0; a.h1 = (const char*)h1; a.w = (const char*)w_packed; a.b3 = b3; a.lse = lse; a.target = target;
  a.lengths = lengths; a.ext_dy = (const char*)ext_dy; a.dy_out = (char*)dy_out; a.dh1_out = (char*)dh1_out;
  a.dskip_out = (char*)dskip_out; a.B = d->B; a.T = d->T; a.Sp = d->Sp; a.Op = d->Op; a.O = d->O; a.scale = d->scale;
  a.inv_count = inv_count;
  hipStream_t st = as_stream(stream);
  if (d->dtype == WAE_F16) return dispatch_head_bwd<f16>(a, st);
  return d->dtype == WAE_BF16 ? dispatch_head_bwd<__bf16>(a, st) : dispatch_head_bwd<float>(a, st);
}
