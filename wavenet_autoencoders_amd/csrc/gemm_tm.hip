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
#include "gemm_tm.hpp"

// timing-only ablation (tools/ablate_tm.sh): -DWAE_TM_ABLATE=bits; 1 no operand loads, 2 no weight DMA, 4 no MFMA, 8 no epilogue
#ifndef WAE_TM_ABLATE
#define WAE_TM_ABLATE 0
#endif
#define TM_ABL(bit) ((WAE_TM_ABLATE & (bit)) != 0)
#ifdef WAE_GLU_PLAIN_LOADS
#define TM_ASM_B 0
#endif
#ifndef TM_ASM_B
#define TM_ASM_B 1     // bf16: inline-asm operand requests two chunks ahead (0: the plain-load loop, one chunk ahead)
#endif

#ifdef WAE_TM_STAMPS
// diagnostic build (tools/stamps_tm.py): per workgroup 8 x u64: [0] life (s_memtime), [1] life (s_memrealtime, 100 MHz), [2] chunk loop,
// [3] epilogue, [4] sum of counted waits of wave 0, [5] sum of its barrier waits, [6] chunks, [7] start (s_memrealtime)
static unsigned long long* g_tm_stamps = nullptr;
extern "C" void wae_debug_set_tm_stamps(unsigned long long* dev_buf) { g_tm_stamps = dev_buf; }
#define TM_TICK() (__builtin_amdgcn_sched_barrier(0), __builtin_amdgcn_s_memtime())
#define TM_STAMP(v) const unsigned long long v = TM_TICK()
#else
#define TM_STAMP(v) do { } while (0)
#endif
// NW = waves per workgroup (each owns 32 time columns).  (Round 3 also built an 8-wave, one-workgroup-per-CU shape whose operand went
// through LDS -- half the weight bytes per column, coalesced operand requests; bit-identical, measured not faster, removed in round 5:
// profiles/EXPERIMENT_LOG.md.)
template <typename E, int NT, int MODE, int OCC, int NW = 4>
__global__ void __launch_bounds__(NW * 64, OCC) gemm_tm_kernel(TmArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  constexpr int CHB = NT * 4 * 1024;
  constexpr int ES = sizeof(E);
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef WAE_TM_STAMPS
  const unsigned long long k_t0 = TM_TICK(), k_w0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long k_wait = 0, k_bar = 0, k_loop0 = 0, k_loop1 = 0, k_n = 0;
#endif
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  constexpr int TW = NW * 32;
  const int tiles_per_b = (p.T + TW - 1) / TW;
  const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int b = tile_id / tiles_per_b;
  const int t0w = (tile_id % tiles_per_b) * TW + wave * 32;
  const int t = t0w + n;
  const bool tvalid = t < p.T;
  const int rows_valid = min(max(p.T - t0w, 0), 32);
  // M above 256 rows: blockIdx.y picks a slice of NT tiles; the weight stream is [slice][chunk], out / aux columns follow
  const int slice = blockIdx.y;
  const int64_t col0 = (int64_t)slice * NT * 32;
  constexpr bool RES_LIKE = MODE == TM_RESIDUAL || MODE == TM_RELU_BWD;
  constexpr bool BIASED = MODE == TM_BIAS_RELU || MODE == TM_CE || MODE == TM_CE_BWD;
  // PAIRED: two workgroups per CU; the gate-backward epilogue then walks the tiles two at a time (fetch of the next
  // pair under the math of this one) so that it fits 256 registers, and stages through 4 KiB per wave
  constexpr bool PAIRED = OCC == 2 && (MODE == TM_GATE_BWD || MODE == TM_RESIDUAL || MODE == TM_RELU_BWD || MODE == TM_BIAS_RELU) && NT % 2 == 0 && sizeof(E) == 2;
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
    if (p.interleave) {
      s = q % p.nsrc;
      q0 = q - q / p.nsrc;       // q - q0 = column block q / nsrc
    } else {
#pragma unroll
      for (int i = 0; i < TM_MAX_SRC - 1; ++i)
        if (q >= qend[i] && i + 1 < p.nsrc) { s = i + 1; q0 = qend[i]; }
    }
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

  const char* wbase = p.w + (int64_t)slice * nq * CHB;
  f32x16 acc[NT];
#pragma unroll
  for (int m = 0; m < NT; ++m) {
    if constexpr (BIASED) {
      init_rows(acc[m], (const float*)p.aux + col0 + 32 * m, h);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    }
  }

  // rows the epilogue needs (residual, or the saved pre-activations z): fetched now, they arrive under the MFMAs
  constexpr int NAUX = PAIRED ? 1 : (MODE == TM_GATE_BWD ? 2 * NT : (RES_LIKE ? NT : 1));
  constexpr int NPASS_AUX = StagePasses<NAUX, E>::N;
  [[maybe_unused]] f32x4 fa[8], fb[8];
  f32x4 pre_a[NPASS_AUX][8];
  [[maybe_unused]] f32x4 pre_b[NPASS_AUX][8];
  if constexpr (PAIRED && RES_LIKE) {
    if (rows_valid > 0 && !TM_ABL(8))
      stage_fetch_pass<E, 2>(fa, p.aux + (((int64_t)b * p.T + t0w) * p.aux_stride + col0) * ES, p.aux_stride * ES, rows_valid, lane);
  } else if constexpr (RES_LIKE) {
    if (rows_valid > 0 && !TM_ABL(8))
      stage_fetch_tiles<E, NT>(pre_a, p.aux + (((int64_t)b * p.T + t0w) * p.aux_stride + col0) * ES, p.aux_stride * ES, rows_valid, lane);
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

  if constexpr (sizeof(E) == 2 && TM_ASM_B) {
    // bf16: operand fragments of chunk q live in group q % 3, requested TWO chunks ahead by inline-asm loads (clamped address,
    // zero fill at use) and retired by a counted wait that leaves the youngest request in flight.  With plain loads hipcc
    // guards the first use of a loop-carried fragment with s_waitcnt vmcnt(0), i.e. every chunk waited for everything it had
    // just requested (an earlier attempt at a deeper pipeline "measured the same" for that reason).  VMEM order per chunk q:
    // [DMA(q+1): the weight ring has two slots][B(q+2)]; at the top of chunk q only B(q+1) may be outstanding.
    auto b_addr = [&](int q, bool& ok) -> const char* {
      int s_ = 0, q0 = 0;
      if (p.interleave) {
        s_ = q % p.nsrc;
        q0 = q - q / p.nsrc;
      } else {
#pragma unroll
        for (int i = 0; i < TM_MAX_SRC - 1; ++i)
          if (q >= qend[i] && i + 1 < p.nsrc) { s_ = i + 1; q0 = qend[i]; }
      }
      const int ts = t + p.src_shift[s_];
      ok = tvalid && ts >= 0 && ts < p.T;
      return p.src[s_] + (((int64_t)b * p.T + (ok ? ts : 0)) * p.src_stride[s_]) * ES + (q - q0) * 128 + h * 16;
    };
    frag G0[4], G1[4], G2[4];
    bool k0, k1, k2;
    auto request_B = [&](int q, frag (&G)[4], bool& ok) {
      const char* src = b_addr(q, ok);
      if (!TM_ABL(1)) { gload_async<0>(G[0], src); gload_async<32>(G[1], src); gload_async<64>(G[2], src); gload_async<96>(G[3], src); }
    };
    if (!TM_ABL(2)) dma_chunk(wbase, smem, CHB, wave, lane);
    request_B(0, G0, k0);
    request_B(min(1, nq - 1), G1, k1);
    auto step = [&](int q, frag (&Gc)[4], bool kc, frag (&Gl)[4], bool& kl) {
      TM_STAMP(s0);
      if (TM_ABL(1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else wait_vmcnt_frags<4>(Gc);
      TM_STAMP(s1);
      __builtin_amdgcn_s_barrier();     // chunk q visible; every wave is past its reads of chunk q-1, whose slot is refilled now
      TM_STAMP(s2);
#ifdef WAE_TM_STAMPS
      k_wait += s1 - s0; k_bar += s2 - s1; ++k_n;
#endif
      {
        const frag z = {};
#pragma unroll
        for (int i = 0; i < 4; ++i) Gc[i] = (kc && !TM_ABL(1)) ? Gc[i] : z;
      }
      // past the end the last chunk is requested again (into the slot / group nobody reads): the wait count never changes
      const int qd = min(q + 1, nq - 1);
      const char* buf = smem + (q & 1) * CHB + lane * 16;
      // (the same requests spread between the chunk's MFMAs -- weight pieces first, then the operand fragments, as the 8-wave shape
      //  issues them -- measured 0.4-1 % SLOWER over the train step, round 4: profiles/EXPERIMENT_LOG.md)
      if (!TM_ABL(2)) dma_chunk(wbase + (int64_t)qd * CHB, smem + ((q + 1) & 1) * CHB, CHB, wave, lane);
      request_B(min(q + 2, nq - 1), Gl, kl);
      if (!TM_ABL(4)) gemm_chunk<4 * NT, NT, 4>(buf, Gc, acc);
      else asm volatile("" : "+v"(Gc[0]), "+v"(Gc[1]), "+v"(Gc[2]), "+v"(Gc[3]));
    };
#ifdef WAE_TM_STAMPS
    k_loop0 = TM_TICK();
#endif
    for (int q = 0; q < nq; q += 3) {
      step(q, G0, k0, G2, k2);
      if (q + 1 < nq) step(q + 1, G1, k1, G0, k0);
      if (q + 2 < nq) step(q + 2, G2, k2, G1, k1);
    }
    // the redundant tail requests must not outlive the ring -- nor their registers: the groups are operands of the wait, or hipcc,
    // to which a request that is never consumed is a dead definition, may move epilogue arithmetic into them above it (round 5:
    // csrc/glu_bwd.hip faulted that way; tools/check_asm_regs.py)
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(G0[0]), "+v"(G0[1]), "+v"(G0[2]), "+v"(G0[3]), "+v"(G1[0]), "+v"(G1[1]), "+v"(G1[2]), "+v"(G1[3]), "+v"(G2[0]),
                   "+v"(G2[1]), "+v"(G2[2]), "+v"(G2[3])
                 :
                 : "memory");
    __syncthreads();
  } else {
  if (!TM_ABL(2)) dma_chunk(wbase, smem, CHB, wave, lane);
  load_B(0, Bn);
  for (int q = 0; q < nq; ++q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) Bc[i] = Bn[i];
    if (q + 1 < nq) {
      if (!TM_ABL(2)) dma_chunk(wbase + (int64_t)(q + 1) * CHB, smem + ((q + 1) & 1) * CHB, CHB, wave, lane);
      load_B(q + 1, Bn);
    }
    const char* buf = smem + (q & 1) * CHB + lane * 16;
    if (!TM_ABL(4)) gemm_chunk<4 * NT, NT, 4>(buf, Bc, acc);
    else asm volatile("" : "+v"(Bc[0]), "+v"(Bc[1]), "+v"(Bc[2]), "+v"(Bc[3]));
  }
  }
  if constexpr (PAIRED) __syncthreads();   // every wave is done with the weight ring: it becomes the staging area
#ifdef WAE_TM_STAMPS
  k_loop1 = TM_TICK();
  struct TmStampOut {
    unsigned long long *o, t0, w0, l0, l1, wt, br, n; int on;
    __device__ ~TmStampOut() {
      if (on) { o[0] = TM_TICK() - t0; o[1] = __builtin_amdgcn_s_memrealtime() - w0; o[2] = l1 - l0; o[3] = TM_TICK() - l1; o[4] = wt; o[5] = br; o[6] = n; o[7] = w0; }
    }
  } stamp_out{p.stamps ? p.stamps + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 : nullptr, k_t0, k_w0, k_loop0, k_loop1, k_wait, k_bar, k_n,
              p.stamps != nullptr && threadIdx.x == 0};
#endif
  if (rows_valid <= 0) return;
  if (TM_ABL(8)) {
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < NT; ++m) s += acc[m][0] + acc[m][7];
    if (s == 123.456f) p.out[0] = 1;
    return;
  }

  [[maybe_unused]] auto combine = [&](float a, float r) -> float {
    if constexpr (MODE == TM_RELU_BWD) return r > 0.f ? p.alpha * a : 0.f;
    else return p.alpha * (a + r);
  };
  if constexpr (MODE == TM_PLAIN) {
    char* orow = p.out + (((int64_t)b * p.T + t0w) * p.out_stride + col0) * ES;
    stage_store_tiles<E, NT>(stg, acc, orow, p.out_stride * ES, rows_valid, lane);
  } else if constexpr (MODE == TM_BIAS_RELU) {
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = fmaxf(acc[m][r] * p.alpha, 0.f);
    char* orow = p.out + (((int64_t)b * p.T + t0w) * p.out_stride + col0) * ES;
    stage_store_tiles<E, NT, PAIRED ? 128 : 256>(stg, acc, orow, p.out_stride * ES, rows_valid, lane);
  } else if constexpr (MODE == TM_CE) {
    // logits (B,O,T) and / or the shifted cross-entropy: same arithmetic as csrc/head_fwd.hip, all O tiles at once
    const TmCe& c = p.ce;
    const bool want_ce = c.target != nullptr && c.nll != nullptr;
    int tgt = -1;
    if (want_ce && tvalid && t + 1 < p.T) tgt = c.target[(int64_t)b * p.T + t + 1];
    float run_m = -INFINITY, run_s = 0.f, picked = 0.f;
    if (tvalid) {
#pragma unroll
      for (int m = 0; m < NT; ++m) {
        float tile_m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cls = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (cls < c.O) {
            if (c.logits) c.logits[((int64_t)b * c.O + cls) * p.T + t] = acc[m][r];
            tile_m = fmaxf(tile_m, acc[m][r]);
            if (cls == tgt) picked = acc[m][r];
          }
        }
        if (want_ce && tile_m > -INFINITY) {
          const float nm = fmaxf(run_m, tile_m);
          float s = run_s * __expf(run_m - nm);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cls = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cls < c.O) s += __expf(acc[m][r] - nm);
          }
          run_m = nm;
          run_s = s;
        }
      }
    }
    if (want_ce) {
      const float om = __shfl_xor(run_m, 32), os = __shfl_xor(run_s, 32), op = __shfl_xor(picked, 32);
      const float nm = fmaxf(run_m, om);
      float s = 0.f;
      if (run_m > -INFINITY) s += run_s * __expf(run_m - nm);
      if (om > -INFINITY) s += os * __expf(om - nm);
      if (tvalid && h == 0) {
        float v = 0.f;
        if (tgt >= 0) v = (nm + __logf(s)) - (picked + op);
        c.nll[(int64_t)b * p.T + t] = v;
        if (c.lse) c.lse[(int64_t)b * p.T + t] = nm + __logf(s);
      }
    }
  } else if constexpr (MODE == TM_CE_BWD) {
    const TmCe& c = p.ce;
    const float lse = tvalid ? c.lse[(int64_t)b * p.T + t] : 0.f;
    int tgt = -1;
    float wt = 0.f;
    if (tvalid && t + 1 < p.T) {
      const int len = c.lengths ? min(c.lengths[b], p.T) : p.T;
      if (t < len - 1) { wt = c.inv_count; tgt = c.target[(int64_t)b * p.T + t + 1]; }
    }
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cls = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
        float g = 0.f;
        if (cls < c.O) g = (expf(acc[m][r] - lse) - (cls == tgt ? 1.f : 0.f)) * wt;
        acc[m][r] = g;
      }
    char* orow = p.out + ((int64_t)b * p.T + t0w) * p.out_stride * ES;
    stage_store_tiles<E, NT>(stg, acc, orow, p.out_stride * ES, rows_valid, lane);
  } else if constexpr (RES_LIKE && PAIRED) {
    const char* arow = p.aux + (((int64_t)b * p.T + t0w) * p.aux_stride + col0) * ES;
    char* orow = p.out + (((int64_t)b * p.T + t0w) * p.out_stride + col0) * ES;
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
      f32x16 res[2];
      stage_unpack_pass<E, 2, 128>(stg, res, fa, lane);
      if (pr + 1 < NT / 2) stage_fetch_pass<E, 2>(fa, arow + (pr + 1) * 64 * ES, p.aux_stride * ES, rows_valid, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[2 * pr + i][r] = combine(acc[2 * pr + i][r], res[i][r]);
      stage_store_pass<E, 2, 128>(stg, &acc[2 * pr], orow + pr * 64 * ES, p.out_stride * ES, rows_valid, lane);
    }
  } else if constexpr (RES_LIKE) {
    f32x16 res[NT];
    stage_unpack_tiles<E, NT>(stg, res, pre_a, lane);
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = combine(acc[m][r], res[m][r]);
    char* orow = p.out + (((int64_t)b * p.T + t0w) * p.out_stride + col0) * ES;
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

// gate-backward / residual / ReLU-backward launches run two workgroups per CU unless the caller sets WAE_TM_ONE_WG in
// wae_tm_desc.flags (A/B measurements; the one-workgroup shape is also what fp32 and odd tile counts use)

template <typename E, int NT, int MODE, int OCC>
static int launch_tm_occ(const TmArgs& a, int nslices, hipStream_t st) {
  constexpr int CHB = NT * 4 * 1024;
  constexpr bool PAIRED = OCC == 2 && (MODE == TM_GATE_BWD || MODE == TM_RESIDUAL || MODE == TM_RELU_BWD || MODE == TM_BIAS_RELU) && NT % 2 == 0 && sizeof(E) == 2;
  const size_t lds = PAIRED ? 2 * CHB : 2 * CHB + 4 * STG_BYTES;
  static WaeLdsCache lds_cache;
  if (int rc = wae_ensure_lds((const void*)gemm_tm_kernel<E, NT, MODE, OCC>, lds_cache, lds, "gemm_tm"); rc != WAE_OK) return rc;
  const int tiles = (a.T + 127) / 128;
  hipLaunchKernelGGL((gemm_tm_kernel<E, NT, MODE, OCC>), dim3(a.B * tiles, nslices), dim3(256), lds, st, a);
  return wae_check_launch("gemm_tm");
}
template <typename E, int NT, int MODE>
static int launch_tm(const TmArgs& a, int nslices, hipStream_t st) {
  if constexpr (sizeof(E) == 2 && NT % 2 == 0 && ((MODE == TM_GATE_BWD && NT <= 6) || MODE == TM_RESIDUAL || MODE == TM_RELU_BWD || MODE == TM_BIAS_RELU)) {
    if (!(a.flags & WAE_TM_ONE_WG)) return launch_tm_occ<E, NT, MODE, 2>(a, nslices, st);
  }
  return launch_tm_occ<E, NT, MODE, 1>(a, nslices, st);
}

// M/32 tiles -> (tiles per slice, slices): up to 8 tiles in one slice, wider outputs in equal slices of 8, 6 or 4 tiles
static bool tm_slicing(int nt, int* nts, int* nslices) {
  if (nt == 1 || nt == 2 || nt == 3 || nt == 4 || nt == 6 || nt == 8) { *nts = nt; *nslices = 1; return true; }
  for (int c : {8, 6, 4})
    if (nt % c == 0) { *nts = c; *nslices = nt / c; return true; }
  return false;
}

template <typename E, int MODE>
static int dispatch_nt(int nt, const TmArgs& a, hipStream_t st) {
  int nts = 0, nsl = 0;
  if (!tm_slicing(nt, &nts, &nsl) || (nsl > 1 && (MODE == TM_GATE_BWD || MODE == TM_CE || MODE == TM_CE_BWD))) {
    wae_set_error("gemm_tm: unsupported M=%d for mode %d (M/32 must be 1,2,3,4,6,8 or, modes 0/1/3/4, a multiple of 4)", nt * 32, MODE);
    return WAE_EUNSUPPORTED;
  }
  if constexpr (MODE == TM_CE || MODE == TM_CE_BWD) {   // M = Op: 128 or 256 classes (padded)
    if (nts == 4) return launch_tm<E, 4, MODE>(a, nsl, st);
    if (nts == 8) return launch_tm<E, 8, MODE>(a, nsl, st);
    wae_set_error("gemm_tm: modes 5/6 need M = 128 or 256 (got %d)", nt * 32);
    return WAE_EUNSUPPORTED;
  } else if constexpr (MODE == TM_BIAS_RELU || MODE == TM_RELU_BWD) {
    switch (nts) {
      case 4: return launch_tm<E, 4, MODE>(a, nsl, st);
      case 6: return launch_tm<E, 6, MODE>(a, nsl, st);
      case 8: return launch_tm<E, 8, MODE>(a, nsl, st);
      default:
        wae_set_error("gemm_tm: modes 3/4 need M a multiple of 128 or 192 (got %d)", nt * 32);
        return WAE_EUNSUPPORTED;
    }
  } else {
    switch (nts) {
      case 1: return launch_tm<E, 1, MODE>(a, nsl, st);
      case 2: return launch_tm<E, 2, MODE>(a, nsl, st);
      case 3: return launch_tm<E, 3, MODE>(a, nsl, st);
      case 4: return launch_tm<E, 4, MODE>(a, nsl, st);
      case 6: return launch_tm<E, 6, MODE>(a, nsl, st);
      default: return launch_tm<E, 8, MODE>(a, nsl, st);
    }
  }
}

template <typename E>
static int dispatch_mode(int mode, int nt, const TmArgs& a, hipStream_t st) {
  switch (mode) {
    case TM_PLAIN: return dispatch_nt<E, TM_PLAIN>(nt, a, st);
    case TM_RESIDUAL: return dispatch_nt<E, TM_RESIDUAL>(nt, a, st);
    case TM_GATE_BWD: return dispatch_nt<E, TM_GATE_BWD>(nt, a, st);
    case TM_BIAS_RELU: return dispatch_nt<E, TM_BIAS_RELU>(nt, a, st);
    case TM_RELU_BWD: return dispatch_nt<E, TM_RELU_BWD>(nt, a, st);
    case TM_CE: return dispatch_nt<E, TM_CE>(nt, a, st);
    default: return dispatch_nt<E, TM_CE_BWD>(nt, a, st);
  }
}

static int tm_run(const wae_tm_desc* d, const void* const* src, const int64_t* src_stride, const int32_t* src_cols,
                  const int32_t* src_shift, const void* w_packed, void* out, int64_t out_stride, const void* aux,
                  int64_t aux_stride, const wae_tm_ce* ce, void* stream) {
  WAE_REQUIRE(d && src && src_stride && src_cols && src_shift && w_packed, "gemm_tm: null pointer argument");
  WAE_REQUIRE(wae_dtype_ok(d->dtype), "gemm_tm: bad dtype");
  WAE_REQUIRE(d->B > 0 && d->T > 0 && d->M > 0 && d->M % 32 == 0, "gemm_tm: bad sizes");
  WAE_REQUIRE(d->nsrc >= 1 && d->nsrc <= TM_MAX_SRC, "gemm_tm: 1..%d sources", TM_MAX_SRC);
  WAE_REQUIRE(d->mode >= 0 && d->mode <= TM_CE_BWD, "gemm_tm: bad mode");
  WAE_REQUIRE(d->mode == TM_PLAIN || aux, "gemm_tm: this mode needs aux");
  WAE_REQUIRE(d->mode == TM_CE || out, "gemm_tm: null output");
  WAE_REQUIRE((d->mode != TM_CE && d->mode != TM_CE_BWD) || ce, "gemm_tm: modes 5/6 go through wae_gemm_tm_ce");
  const int ck = wae_is16(d->dtype) ? 64 : 32;
  TmArgs a;
  for (int s = 0; s < TM_MAX_SRC; ++s) {
    const bool on = s < d->nsrc;
    a.src[s] = on ? (const char*)src[s] : nullptr;
    a.src_stride[s] = on ? src_stride[s] : 0;
    a.src_cols[s] = on ? src_cols[s] : 0;
    a.src_shift[s] = on ? src_shift[s] : 0;
    WAE_REQUIRE(!on || (src[s] && src_cols[s] > 0 && src_cols[s] % ck == 0), "gemm_tm: source %d: cols must be a multiple of %d", s, ck);
  }
#ifdef WAE_TM_STAMPS
  a.stamps = g_tm_stamps;
#else
  a.stamps = nullptr;
#endif
  a.nsrc = d->nsrc; a.w = (const char*)w_packed; a.out = (char*)out; a.out_stride = out_stride; a.aux = (const char*)aux;
  a.aux_stride = aux_stride; a.alpha = d->alpha; a.B = d->B; a.T = d->T; a.mode = d->mode;
  a.interleave = (d->flags & WAE_TM_INTERLEAVE) ? 1 : 0;
  a.flags = d->flags;
  a.nslices = 0;
  if (a.interleave)
    for (int s = 1; s < d->nsrc; ++s) WAE_REQUIRE(src_cols[s] == src_cols[0], "gemm_tm: interleaved sources must be equally wide");
  a.ce = TmCe{nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, 0};
  if (ce) {
    WAE_REQUIRE(ce->O > 0 && ce->O <= d->M, "gemm_tm: 0 < O <= M");
    if (d->mode == TM_CE) WAE_REQUIRE(ce->logits || (ce->target && ce->nll), "gemm_tm: mode 5 produces logits and / or nll");
    if (d->mode == TM_CE_BWD) WAE_REQUIRE(ce->lse && ce->target, "gemm_tm: mode 6 needs lse and target");
    a.ce = TmCe{ce->logits, ce->target, ce->nll, ce->lse, ce->lengths, ce->inv_count, ce->O};
  }
  hipStream_t st = as_stream(stream);
  const int nt = d->M / 32;
  {   // the long-K, one-source contractions (the head's skip GEMM) on the static 8-wave schedule: csrc/gemm_tm8.hip
    bool handled = false;
    const int rc = wae_gemm_tm8_launch(a, d->dtype, d->M, st, &handled);
    if (rc != WAE_OK || handled) return rc;
  }
  if (d->dtype == WAE_BF16) return dispatch_mode<__bf16>(d->mode, nt, a, st);
  if (d->dtype == WAE_F16) return dispatch_mode<f16>(d->mode, nt, a, st);
  return dispatch_mode<float>(d->mode, nt, a, st);
}

extern "C" int wae_gemm_tm(const wae_tm_desc* d, const void* const* src, const int64_t* src_stride, const int32_t* src_cols,
                           const int32_t* src_shift, const void* w_packed, void* out, int64_t out_stride, const void* aux,
                           int64_t aux_stride, void* stream) {
  return tm_run(d, src, src_stride, src_cols, src_shift, w_packed, out, out_stride, aux, aux_stride, nullptr, stream);
}

extern "C" int wae_gemm_tm_ce(const wae_tm_desc* d, const void* const* src, const int64_t* src_stride, const int32_t* src_cols,
                              const int32_t* src_shift, const void* w_packed, void* out, int64_t out_stride, const float* bias,
                              const wae_tm_ce* ce, void* stream) {
  WAE_REQUIRE(d && (d->mode == TM_CE || d->mode == TM_CE_BWD) && ce && bias, "gemm_tm_ce: mode 5 or 6 with ce and bias");
  return tm_run(d, src, src_stride, src_cols, src_shift, w_packed, out, out_stride, bias, 0, ce, stream);
}
