// wae_gemm_tm: C[t][M] = sum_sources W_s[M, K_s] . X_s[t + shift_s][K_s]   on time-major (rows = time) operands.
//
// The backward data path of the gated stack is three instances of this one kernel (autograd of
// modules.py:115-163 / wavenet.py:204-214):
//   du/dz  : du_l = (sqrt(.5) W_out_l)^T dx_{l+1}-hat + W_skip_l^T dskip ;  dz_l = gate'(z_l) * du_l      (GATE_BWD)
//   dx     : dx_l-hat = sqrt(.5) * ( dx_{l+1}-hat + sum_tap W1_tap^T dz_l[t + (k-1-tap) d] )                (RESIDUAL)
//   dc     : dc = [Wc_0^T .. Wc_{L-1}^T] . [dz_0 ; .. ; dz_{L-1}]                                           (PLAIN)
// ("-hat" = the stored gradient carries the layer's sqrt(.5) factor already.)
// Same decomposition as the forward kernels: 128 time steps per workgroup, one wave per 32 time columns owning all
// M rows (NT 32x32 accumulator tiles), weights in A-fragment order through a double-buffered LDS ring, operands as
// 16-byte fragments straight from L2/HBM with the clip boundary as a zero-fill predicate.
#include "wae_common.hpp"

// timing-only ablation (tools/ablate_tm.sh): -DWAE_TM_ABLATE=bits; 1 no operand loads, 2 no weight DMA, 4 no MFMA, 8 no epilogue
#ifndef WAE_TM_ABLATE
#define WAE_TM_ABLATE 0
#endif
#define TM_ABL(bit) ((WAE_TM_ABLATE & (bit)) != 0)

#define TM_MAX_SRC 4
#define TM_PLAIN 0
#define TM_RESIDUAL 1  // out = alpha * (acc + res[t])
#define TM_GATE_BWD 2  // acc = du (NT = Hp/32 tiles); out (t, 2Hp) = [da | db] from z (t, 2Hp)

struct TmArgs {
  const char* src[TM_MAX_SRC];
  int64_t src_stride[TM_MAX_SRC];  // elements per row
  int src_cols[TM_MAX_SRC];        // multiple of CK
  int src_shift[TM_MAX_SRC];       // operand row = t + shift (zero outside [0,T))
  int nsrc;
  const char* w;
  char* out;
  int64_t out_stride;
  const char* aux;  // RESIDUAL: res (t, M) ; GATE_BWD: z (t, 2Hp)
  int64_t aux_stride;
  float alpha;
  int B, T, mode;
};

template <typename E, int NT, int MODE, int OCC>
__global__ void __launch_bounds__(256, OCC) gemm_tm_kernel(TmArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  constexpr int CHB = NT * 4 * 1024;
  constexpr int ES = sizeof(E);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  const int tiles_per_b = (p.T + 127) >> 7;
  const int b = blockIdx.x / tiles_per_b;
  const int t0w = (blockIdx.x % tiles_per_b) * 128 + wave * 32;
  const int t = t0w + n;
  const bool tvalid = t < p.T;
  const int rows_valid = min(max(p.T - t0w, 0), 32);
  // PAIRED: two workgroups per CU; the gate-backward epilogue then walks the tiles two at a time (fetch of the next
  // pair under the math of this one) so that it fits 256 registers, and stages through 4 KiB per wave
  constexpr bool PAIRED = OCC == 2 && MODE != TM_PLAIN && NT % 2 == 0 && sizeof(E) == 2;
  constexpr int STGB = PAIRED ? 4096 : STG_BYTES;
  // PAIRED stages through the (then idle) weight ring after the chunk loop: 2 x (2 CHB) <= 128 KiB of LDS per CU
  char* stg = PAIRED ? smem + wave * STGB : smem + 2 * CHB + wave * STGB;

  // chunk -> (source, column block)
  int qend[TM_MAX_SRC];
  int nq = 0;
#pragma unroll
  for (int s = 0; s < TM_MAX_SRC; ++s) {
    if (s < p.nsrc) nq += p.src_cols[s] / T_::CK;
    qend[s] = nq;
  }
  frag Bn[4], Bc[4];
  auto load_B = [&](int q, frag (&Bf)[4]) {
    int s = 0, q0 = 0;
#pragma unroll
    for (int i = 0; i < TM_MAX_SRC - 1; ++i)
      if (q >= qend[i] && i + 1 < p.nsrc) { s = i + 1; q0 = qend[i]; }
    const int ts = t + p.src_shift[s];
    const bool ok = tvalid && ts >= 0 && ts < p.T;
    const char* src = p.src[s] + (((int64_t)b * p.T + (ok ? ts : 0)) * p.src_stride[s]) * ES + (q - q0) * 128 + h * 16;
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      if (ok && !TM_ABL(1)) {
        Bf[blk] = *(const frag*)(src + blk * 32);
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
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // rows the epilogue needs (residual, or the saved pre-activations z): fetched now, they arrive under the MFMAs
  constexpr int NAUX = PAIRED ? 1 : (MODE == TM_GATE_BWD ? 2 * NT : (MODE == TM_RESIDUAL ? NT : 1));
  constexpr int NPASS_AUX = StagePasses<NAUX, E>::N;
  [[maybe_unused]] f32x4 fa[8], fb[8];
  f32x4 pre_a[NPASS_AUX][8];
  [[maybe_unused]] f32x4 pre_b[NPASS_AUX][8];
  if constexpr (PAIRED && MODE == TM_RESIDUAL) {
    if (rows_valid > 0 && !TM_ABL(8))
      stage_fetch_pass<E, 2>(fa, p.aux + ((int64_t)b * p.T + t0w) * p.aux_stride * ES, p.aux_stride * ES, rows_valid, lane);
  } else if constexpr (MODE == TM_RESIDUAL) {
    if (rows_valid > 0 && !TM_ABL(8))
      stage_fetch_tiles<E, NT>(pre_a, p.aux + ((int64_t)b * p.T + t0w) * p.aux_stride * ES, p.aux_stride * ES, rows_valid, lane);
  } else if constexpr (PAIRED) {
    if (rows_valid > 0 && !TM_ABL(8)) {
      const char* zrow = p.aux + ((int64_t)b * p.T + t0w) * p.aux_stride * ES;
      stage_fetch_pass<E, 2>(fa, zrow, p.aux_stride * ES, rows_valid, lane);
      stage_fetch_pass<E, 2>(fb, zrow + (int64_t)NT * 32 * ES, p.aux_stride * ES, rows_valid, lane);
    }
  } else if constexpr (MODE == TM_GATE_BWD) {
    if (rows_valid > 0 && !TM_ABL(8)) {
      const char* zrow = p.aux + ((int64_t)b * p.T + t0w) * p.aux_stride * ES;
      stage_fetch_tiles<E, NT>(pre_a, zrow, p.aux_stride * ES, rows_valid, lane);
      stage_fetch_tiles<E, NT>(pre_b, zrow + (int64_t)NT * 32 * ES, p.aux_stride * ES, rows_valid, lane);
    }
  }

  if (!TM_ABL(2)) dma_chunk(p.w, smem, CHB, wave, lane);
  load_B(0, Bn);
  for (int q = 0; q < nq; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
    if (q + 1 < nq) {
      if (!TM_ABL(2)) dma_chunk(p.w + (int64_t)(q + 1) * CHB, smem + ((q + 1) & 1) * CHB, CHB, wave, lane);
      load_B(q + 1, Bn);
    }
    const char* buf = smem + (q & 1) * CHB + lane * 16;
    if (!TM_ABL(4)) gemm_chunk<4 * NT, NT, 4>(buf, Bc, acc);
    else asm volatile("" : "+v"(Bc[0]), "+v"(Bc[1]), "+v"(Bc[2]), "+v"(Bc[3]));
  }
  if constexpr (PAIRED) __syncthreads();   // every wave is done with the weight ring: it becomes the staging area
  if (rows_valid <= 0) return;
  if (TM_ABL(8)) {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < NT; ++m) s += acc[m][0] + acc[m][7];
    if (s == 123.456f) p.out[0] = 1;
    return;
  }

  if constexpr (MODE == TM_PLAIN) {
    char* orow = p.out + ((int64_t)b * p.T + t0w) * p.out_stride * ES;
    stage_store_tiles<E, NT>(stg, acc, orow, p.out_stride * ES, rows_valid, lane);
  } else if constexpr (MODE == TM_RESIDUAL && PAIRED) {
    const char* arow = p.aux + ((int64_t)b * p.T + t0w) * p.aux_stride * ES;
    char* orow = p.out + ((int64_t)b * p.T + t0w) * p.out_stride * ES;
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
      f32x16 res[2];
      stage_unpack_pass<E, 2, 128>(stg, res, fa, lane);
      if (pr + 1 < NT / 2) stage_fetch_pass<E, 2>(fa, arow + (pr + 1) * 64 * ES, p.aux_stride * ES, rows_valid, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[2 * pr + i][r] = p.alpha * (acc[2 * pr + i][r] + res[i][r]);
      stage_store_pass<E, 2, 128>(stg, &acc[2 * pr], orow + pr * 64 * ES, p.out_stride * ES, rows_valid, lane);
    }
  } else if constexpr (MODE == TM_RESIDUAL) {
    f32x16 res[NT];
    stage_unpack_tiles<E, NT>(stg, res, pre_a, lane);
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = p.alpha * (acc[m][r] + res[m][r]);
    char* orow = p.out + ((int64_t)b * p.T + t0w) * p.out_stride * ES;
    stage_store_tiles<E, NT>(stg, acc, orow, p.out_stride * ES, rows_valid, lane);
  } else {
    // gate backward (modules.py:154: u = tanh(a) * sigmoid(b)):  da = du * s * (1 - th^2),  db = du * th * s * (1 - s)
    char* orow = p.out + ((int64_t)b * p.T + t0w) * p.out_stride * ES;
    auto gate_bwd = [&](f32x16& za, f32x16& zg, const f32x16& du_t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float th, sg;
        if constexpr (sizeof(E) == 4) {
          th = tanhf(za[r]);
          sg = 1.0f / (1.0f + expf(-zg[r]));
        } else {
          const float ea = __builtin_amdgcn_exp2f(fmaxf(za[r], -15.0f) * -2.885390081777927f);
          th = (1.0f - ea) * fast_rcp(1.0f + ea);
          sg = fast_rcp(1.0f + __builtin_amdgcn_exp2f(zg[r] * -1.4426950408889634f));
        }
        const float du = du_t[r];
        za[r] = du * sg * (1.0f - th * th);
        zg[r] = du * th * sg * (1.0f - sg);
      }
    };
    if constexpr (PAIRED) {
      const char* zrow = p.aux + ((int64_t)b * p.T + t0w) * p.aux_stride * ES;
#pragma unroll
      for (int pr = 0; pr < NT / 2; ++pr) {
        f32x16 za[2], zg[2];
        stage_unpack_pass<E, 2, 128>(stg, za, fa, lane);
        stage_unpack_pass<E, 2, 128>(stg, zg, fb, lane);
        if (pr + 1 < NT / 2) {
          stage_fetch_pass<E, 2>(fa, zrow + (pr + 1) * 64 * ES, p.aux_stride * ES, rows_valid, lane);
          stage_fetch_pass<E, 2>(fb, zrow + ((int64_t)NT * 32 + (pr + 1) * 64) * ES, p.aux_stride * ES, rows_valid, lane);
        }
        gate_bwd(za[0], zg[0], acc[2 * pr]);
        gate_bwd(za[1], zg[1], acc[2 * pr + 1]);
        stage_store_pass<E, 2, 128>(stg, za, orow + pr * 64 * ES, p.out_stride * ES, rows_valid, lane);
        stage_store_pass<E, 2, 128>(stg, zg, orow + ((int64_t)NT * 32 + pr * 64) * ES, p.out_stride * ES, rows_valid, lane);
      }
    } else {
      f32x16 za[NT], zg[NT];
      stage_unpack_tiles<E, NT>(stg, za, pre_a, lane);
      stage_unpack_tiles<E, NT>(stg, zg, pre_b, lane);
#pragma unroll
      for (int m = 0; m < NT; ++m) gate_bwd(za[m], zg[m], acc[m]);
      stage_store_tiles<E, NT>(stg, za, orow, p.out_stride * ES, rows_valid, lane);
      stage_store_tiles<E, NT>(stg, zg, orow + (int64_t)NT * 32 * ES, p.out_stride * ES, rows_valid, lane);
    }
  }
}

static int g_tm_occ = 2 | 4;   // bit 1: gate-backward launches, bit 2: residual launches run two workgroups per CU
extern "C" void wae_debug_set_tm_occ(int occ) { g_tm_occ = occ; }

template <typename E, int NT, int MODE, int OCC>
static int launch_tm_occ(const TmArgs& a, hipStream_t st) {
  constexpr int CHB = NT * 4 * 1024;
  constexpr bool PAIRED = OCC == 2 && MODE != TM_PLAIN && NT % 2 == 0 && sizeof(E) == 2;
  const size_t lds = PAIRED ? 2 * CHB : 2 * CHB + 4 * STG_BYTES;
  static size_t attr_done = 0;
  if (attr_done < lds) {
    if (hipFuncSetAttribute((const void*)gemm_tm_kernel<E, NT, MODE, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      wae_set_error("gemm_tm: cannot raise dynamic LDS to %zu", lds);
      return WAE_EHIP;
    }
    attr_done = lds;
  }
  const int tiles = (a.T + 127) / 128;
  hipLaunchKernelGGL((gemm_tm_kernel<E, NT, MODE, OCC>), dim3(a.B * tiles), dim3(256), lds, st, a);
  return wae_check_launch("gemm_tm");
}
template <typename E, int NT, int MODE>
static int launch_tm(const TmArgs& a, hipStream_t st) {
  if constexpr (sizeof(E) == 2 && NT % 2 == 0 && ((MODE == TM_GATE_BWD && NT <= 6) || MODE == TM_RESIDUAL)) {
    if (g_tm_occ & (MODE == TM_GATE_BWD ? 2 : 4)) return launch_tm_occ<E, NT, MODE, 2>(a, st);
  }
  return launch_tm_occ<E, NT, MODE, 1>(a, st);
}

template <typename E, int MODE>
static int dispatch_nt(int nt, const TmArgs& a, hipStream_t st) {
  switch (nt) {
    case 1: return launch_tm<E, 1, MODE>(a, st);
    case 2: return launch_tm<E, 2, MODE>(a, st);
    case 3: return launch_tm<E, 3, MODE>(a, st);
    case 4: return launch_tm<E, 4, MODE>(a, st);
    case 6: return launch_tm<E, 6, MODE>(a, st);
    case 8: return launch_tm<E, 8, MODE>(a, st);
    default:
      wae_set_error("gemm_tm: unsupported M=%d (M/32 must be 1,2,3,4,6 or 8)", nt * 32);
      return WAE_EUNSUPPORTED;
  }
}

extern "C" int wae_gemm_tm(const wae_tm_desc* d, const void* const* src, const int64_t* src_stride, const int32_t* src_cols,
                           const int32_t* src_shift, const void* w_packed, void* out, int64_t out_stride, const void* aux,
                           int64_t aux_stride, void* stream) {
  WAE_REQUIRE(d && src && src_stride && src_cols && src_shift && w_packed && out, "gemm_tm: null pointer argument");
  WAE_REQUIRE(d->dtype == WAE_F32 || d->dtype == WAE_BF16, "gemm_tm: bad dtype");
  WAE_REQUIRE(d->B > 0 && d->T > 0 && d->M > 0 && d->M % 32 == 0, "gemm_tm: bad sizes");
  WAE_REQUIRE(d->nsrc >= 1 && d->nsrc <= TM_MAX_SRC, "gemm_tm: 1..%d sources", TM_MAX_SRC);
  WAE_REQUIRE(d->mode >= 0 && d->mode <= 2, "gemm_tm: bad mode");
  WAE_REQUIRE(d->mode == TM_PLAIN || aux, "gemm_tm: this mode needs aux");
  const int ck = d->dtype == WAE_BF16 ? 64 : 32;
  TmArgs a;
  for (int s = 0; s < TM_MAX_SRC; ++s) {
    const bool on = s < d->nsrc;
    a.src[s] = on ? (const char*)src[s] : nullptr;
    a.src_stride[s] = on ? src_stride[s] : 0;
    a.src_cols[s] = on ? src_cols[s] : 0;
    a.src_shift[s] = on ? src_shift[s] : 0;
    WAE_REQUIRE(!on || (src[s] && src_cols[s] > 0 && src_cols[s] % ck == 0), "gemm_tm: source %d: cols must be a multiple of %d", s, ck);
  }
  a.nsrc = d->nsrc; a.w = (const char*)w_packed; a.out = (char*)out; a.out_stride = out_stride; a.aux = (const char*)aux;
  a.aux_stride = aux_stride; a.alpha = d->alpha; a.B = d->B; a.T = d->T; a.mode = d->mode;
  hipStream_t st = as_stream(stream);
  const int nt = d->M / 32;
  if (d->dtype == WAE_BF16) {
    if (d->mode == TM_PLAIN) return dispatch_nt<__bf16, TM_PLAIN>(nt, a, st);
    if (d->mode == TM_RESIDUAL) return dispatch_nt<__bf16, TM_RESIDUAL>(nt, a, st);
    return dispatch_nt<__bf16, TM_GATE_BWD>(nt, a, st);
  }
  if (d->mode == TM_PLAIN) return dispatch_nt<float, TM_PLAIN>(nt, a, st);
  if (d->mode == TM_RESIDUAL) return dispatch_nt<float, TM_RESIDUAL>(nt, a, st);
  return dispatch_nt<float, TM_GATE_BWD>(nt, a, st);
}
