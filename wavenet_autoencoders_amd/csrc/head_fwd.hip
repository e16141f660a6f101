// wae_head_fwd: skip contraction + decoder head + fused softmax cross-entropy
// (reference: modules.py:157 conv1x1_skip, wavenet.py:204-214; vqwae_train.py:363-379 with the shift of :764).
//
//   skips = sum_l b_skip_l + [W_skip_0 .. W_skip_{L-1}][Sp, Ku] . u[Ku, t]     (GEMM 0, K = all layers' gated
//   h0 = relu(skips * sqrt(1/L))                                                activations; replaces L passes of
//   h1 = relu(b1 + W1[Sp,Sp] . h0)                                (GEMM 1)      `skips += h`, wavenet.py:206)
//   y  = b3 + W3[Op,Sp] . h1                                      (GEMM 2) -> logits (B,O,T) fp32 and/or
//                                                                 nll[b,t] = lse(y) - y[target[t+1]]
//
// Same decomposition as glu_fwd.hip: 128 time steps per workgroup, one wave per 32 time columns, weights in
// A-fragment order through a double-buffered LDS ring, accumulator tiles reused as the next MFMA's B operand,
// so skips, h0 and h1 never leave the register file.
#include "wae_common.hpp"

struct HeadArgs {
  const char* u;
  const char* w;
  const float* bias;  // [Sp skip-bias sum | Sp b1 | Op b3]
  float* logits;
  const int32_t* target;
  float* nll;
  float* lse;
  char* h0_save;
  char* h1_save;
  int B, T, Ku, Sp, Op, O;
  float scale;
};

// FROM_H0: GEMM 0 ran as its own launch (wae_gemm_tm, mode BIAS_RELU: two workgroups per CU hide the long K loop's memory trips
// behind each other, which this kernel's one workgroup per CU cannot); p.u then is h0 (B,T,Sp) and p.w starts at GEMM 1's chunks.
// NW waves per workgroup (32 time columns each), NSL ring slots for the chunks behind GEMM 0.  The FROM_H0 launches of 16-bit engines
// run 8 waves x 256 columns on a three-slot ring, weights two chunks ahead under a counted wait (round 6): the 4-wave form waited
// for every chunk's DMA in full right after requesting it (one chunk ahead, vmcnt(0)) on two rounds of workgroups per CU -- 47 us at C2
// for 9 us of MFMAs and 5 us of HBM.
template <typename E, int NT, bool FROM_H0 = false, int NW = 4, int NSL = 2>
__global__ void __launch_bounds__(NW * 64, 1) head_fwd_kernel(HeadArgs p) {
  using T_ = ET<E>;
  using frag = typename T_::frag;
  constexpr int CHB = NT * 4 * 1024;
  constexpr int ES = sizeof(E);
  constexpr int KBU = T_::KBU;
  constexpr int NKB = NT * KBU;
  constexpr int MT2 = 4 / KBU;  // bf16: 2, f32: 1  (MT2 * NKB KiB == CHB)
  constexpr int NQ1 = NT / MT2;  // chunks of GEMM 1 (Sp output tiles)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int n = lane & 31, h = lane >> 5;
  static_assert(NW == 4 || (FROM_H0 && NW == 8), "the 8-wave shape exists for the FROM_H0 launches");
  static_assert(NSL == 2 || FROM_H0, "deeper rings behind GEMM 0: FROM_H0 only");
  constexpr int TW = NW * 32;
  constexpr int PPW = CHB / NW / 1024;         // LDS-DMA pieces per wave and chunk
  const int tiles_per_b = (p.T + TW - 1) / TW;
  const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
  const int b = tile_id / tiles_per_b;
  const int t0w = (tile_id % tiles_per_b) * TW + wave * 32;
  const int t = t0w + n;
  const bool tvalid = t < p.T;
  const int rows_valid = min(max(p.T - t0w, 0), 32);

  const int nq0 = FROM_H0 ? 0 : p.Ku / T_::CK;
  const int nq2 = (p.Op >> 5) / MT2;
  const int nq_total = nq0 + NQ1 + nq2;
  const char* urow = p.u + ((int64_t)b * p.T + (tvalid ? t : 0)) * p.Ku * ES + h * 16;
  constexpr int RSL = FROM_H0 ? NSL : 3;       // ring slots allocated (GEMM 0 runs a three-slot ring)
  constexpr int STGW = NW == 8 ? 4096 : STG_BYTES, SPITCH = STGW / 32;      // staging tile per wave (8 waves: row pitch 128 B)
  char* stg = smem + RSL * CHB + wave * STGW;
  float* bias_lds = (float*)(smem + RSL * CHB + NW * STGW);

  // biases -> LDS once (accumulator inits then never touch vmcnt)
  for (int i = threadIdx.x * 4; i < 2 * p.Sp + p.Op; i += NW * 256) *(f32x4*)(bias_lds + i) = *(const f32x4*)(p.bias + i);
  // chunk qi (counted from GEMM 1's first) -> slot qi % NSL, requested NSL - 1 chunks ahead; the wait in front of chunk qi leaves the
  // NSL - 2 younger chunks' pieces in flight (stores issued in between only make it stricter)
  [[maybe_unused]] auto dma_tail = [&](int qt) {
    if (nq0 + qt < nq_total) dma_chunk<NW>(p.w + (int64_t)(nq0 + qt) * CHB, smem + (qt % NSL) * CHB, CHB, wave, lane);
  };
  [[maybe_unused]] auto wait_tail = [&]() {
    if constexpr (NSL == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSL - 2) * PPW) : "memory");
  };

  // ---- GEMM 0: skip contraction over all layers' u --------------------------------------------------------
  // K = Ku is long (72 chunks at C2) and its operand comes from HBM: with the operand one chunk ahead and one workgroup
  // per CU every chunk waited a full memory round trip (3.3 us against 0.43 us of MFMAs: 0.29 ms per launch).  Now the
  // operand fragments of chunk q live in group q % 4, requested THREE chunks ahead by inline-asm loads, the weight chunks
  // go through a 3-slot ring two chunks ahead, the VMEM issue of a chunk is spread over its MFMAs, and the wait at the top
  // of chunk q is counted: it leaves B(q+2) and DMA(q+1) in flight.
  f32x16 acc[NT];
  frag uf[NKB];
  if constexpr (FROM_H0) {
#pragma unroll
    for (int qt = 0; qt < NSL - 1; ++qt) dma_tail(qt);
    // h0 rows of this wave's 32 time columns -> accumulator layout (coalesced rows through the staging tile) -> operand fragments
#pragma unroll
    for (int m = 0; m < NT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    if (rows_valid > 0) stage_load_tiles<E, NT, SPITCH>(stg, acc, p.u + ((int64_t)b * p.T + t0w) * p.Sp * ES, (int64_t)p.Sp * ES, rows_valid, lane);
#pragma unroll
    for (int m = 0; m < NT; ++m) {
      frag tmp[KBU];
      acc_to_frags(acc[m], tmp);
#pragma unroll
      for (int s = 0; s < KBU; ++s) uf[m * KBU + s] = tmp[s];
    }
  } else {
#pragma unroll
  for (int m = 0; m < NT; ++m) init_rows(acc[m], p.bias + 32 * m, h);
  {
    constexpr int PPW = NT;                      // 1-KiB DMA pieces per wave and chunk
    frag G0[4], G1[4], G2[4], G3[4], Bc[4];
    auto request_B = [&](int q, frag (&G)[4]) {
      const char* s0 = urow + (int64_t)q * 128;
      gload_async<0>(G[0], s0); gload_async<32>(G[1], s0); gload_async<64>(G[2], s0); gload_async<96>(G[3], s0);
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // bias / init loads retired: from here on VMEM ops are counted by hand
    request_B(0, G0);
    request_B(min(1, nq0 - 1), G1);
    dma_chunk(p.w, smem, CHB, wave, lane);
    request_B(min(2, nq0 - 1), G2);
    dma_chunk(p.w + (int64_t)min(1, nq0 - 1) * CHB, smem + CHB, CHB, wave, lane);
    int slot = 0;
    auto step = [&](int q, frag (&Gc)[4], frag (&Gl)[4]) {
      wait_vmcnt_frags<4 + PPW>(Gc);
      __builtin_amdgcn_s_barrier();   // chunk q visible; every wave is past its reads of chunk q-1, whose slot is refilled now
      {
        const frag z = {};
#pragma unroll
        for (int i = 0; i < 4; ++i) Bc[i] = tvalid ? Gc[i] : z;
      }
      // past the end the last chunk is requested again (into a group / slot nobody reads): the issue pattern never changes
      const char* s0 = urow + (int64_t)min(q + 3, nq0 - 1) * 128;
      int slot_d = slot + 2; if (slot_d >= 3) slot_d -= 3;
      const char* dsrc = p.w + (int64_t)min(q + 2, nq0 - 1) * CHB + wave * (CHB / 4) + lane * 16;
      char* ddst = smem + slot_d * CHB + wave * (CHB / 4);
      auto filler = [&](auto ic) {
        constexpr int I = decltype(ic)::value;
        constexpr int SP = (4 * NT) / (4 + PPW);
        if constexpr (I % SP == 0 && I / SP < 4 + PPW) {
          constexpr int k = I / SP;
          if constexpr (k < 4) gload_async<k * 32>(Gl[k], s0);
          else dma_piece(dsrc + (k - 4) * 1024, ddst + (k - 4) * 1024);
        }
      };
      gemm_chunk_fill<4 * NT, NT, 4>(smem + slot * CHB + lane * 16, Bc, acc, filler);
      slot = slot + 1 == 3 ? 0 : slot + 1;
    };
    for (int q = 0; q < nq0; q += 4) {
      step(q, G0, G3);
      if (q + 1 < nq0) step(q + 1, G1, G0);
      if (q + 2 < nq0) step(q + 2, G2, G1);
      if (q + 3 < nq0) step(q + 3, G3, G2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                  // every wave is done with the ring: the remaining chunks alternate between its first two slots
    dma_chunk(p.w + (int64_t)nq0 * CHB, smem, CHB, wave, lane);
  }

  // ---- h0 = relu(skips * scale) -> operand fragments (+ optional save) ------------------------------------
#pragma unroll
  for (int m = 0; m < NT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = fmaxf(acc[m][r] * p.scale, 0.f);
    frag tmp[KBU];
    acc_to_frags(acc[m], tmp);
#pragma unroll
    for (int s = 0; s < KBU; ++s) uf[m * KBU + s] = tmp[s];
  }
  if (p.h0_save && rows_valid > 0)
    stage_store_tiles<E, NT, SPITCH>(stg, acc, p.h0_save + ((int64_t)b * p.T + t0w) * p.Sp * ES, (int64_t)p.Sp * ES, rows_valid, lane);
  }

  // ---- GEMM 1: h1 = relu(b1 + W1 . h0); all NT output tiles stay in registers --------------------------------
#pragma unroll
  for (int q1 = 0; q1 < NQ1; ++q1) {
    const int qi = nq0 + q1;
    if constexpr (FROM_H0) {
      if (q1 == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the h0 rows: plain loads hipcc waits for itself)
      else wait_tail();
      __builtin_amdgcn_s_barrier();
      dma_tail(q1 + NSL - 1);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (qi + 1 < nq_total) dma_chunk(p.w + (int64_t)(qi + 1) * CHB, smem + ((qi - nq0 + 1) & 1) * CHB, CHB, wave, lane);
    }
    const char* buf = smem + (FROM_H0 ? ((qi - nq0) % NSL) : ((qi - nq0) & 1)) * CHB + lane * 16;
    f32x16(&y)[MT2] = *reinterpret_cast<f32x16(*)[MT2]>(&acc[q1 * MT2]);
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) init_rows(y[mt], bias_lds + p.Sp + 32 * (q1 * MT2 + mt), h);
    gemm_chunk<MT2 * NKB, MT2, NKB, true>(buf, uf, y);
  }
#pragma unroll
  for (int m = 0; m < NT; ++m) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = fmaxf(acc[m][r], 0.f);
    frag tmp[KBU];
    acc_to_frags(acc[m], tmp);
#pragma unroll
    for (int s = 0; s < KBU; ++s) uf[m * KBU + s] = tmp[s];
  }
  if (p.h1_save && rows_valid > 0)
    stage_store_tiles<E, NT, SPITCH>(stg, acc, p.h1_save + ((int64_t)b * p.T + t0w) * p.Sp * ES, (int64_t)p.Sp * ES, rows_valid, lane);

  // ---- GEMM 2 + logits store + online log-sum-exp ------------------------------------------------------------
  const bool want_ce = p.target != nullptr && p.nll != nullptr;
  int tgt = -1;
  if (want_ce && tvalid && t + 1 < p.T) tgt = p.target[(int64_t)b * p.T + t + 1];
  float run_m = -INFINITY, run_s = 0.f, picked = 0.f;
  const float* b3 = bias_lds + 2 * p.Sp;
  for (int q2 = 0; q2 < nq2; ++q2) {
    const int qi = nq0 + NQ1 + q2;
    if constexpr (FROM_H0) {
      wait_tail();
      __builtin_amdgcn_s_barrier();
      dma_tail(NQ1 + q2 + NSL - 1);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (qi + 1 < nq_total) dma_chunk(p.w + (int64_t)(qi + 1) * CHB, smem + ((qi - nq0 + 1) & 1) * CHB, CHB, wave, lane);
    }
    const char* buf = smem + (FROM_H0 ? ((qi - nq0) % NSL) : ((qi - nq0) & 1)) * CHB + lane * 16;
    f32x16 y[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) init_rows(y[mt], b3 + 32 * (q2 * MT2 + mt), h);
    gemm_chunk<MT2 * NKB, MT2, NKB, true>(buf, uf, y);
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
      const int gm = q2 * MT2 + mt;
      if (tvalid) {
        float tile_m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cls = 32 * gm + (r & 3) + 8 * (r >> 2) + 4 * h;
          if (cls < p.O) {
            if (p.logits) p.logits[((int64_t)b * p.O + cls) * p.T + t] = y[mt][r];
            tile_m = fmaxf(tile_m, y[mt][r]);
            if (cls == tgt) picked = y[mt][r];
          }
        }
        if (want_ce && tile_m > -INFINITY) {
          const float nm = fmaxf(run_m, tile_m);
          float s = run_s * __expf(run_m - nm);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cls = 32 * gm + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cls < p.O) s += __expf(y[mt][r] - nm);
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
      if (p.lse) p.lse[(int64_t)b * p.T + t] = nm + __logf(s);
    }
  }
}

template <typename E, int NT, bool FROM_H0 = false>
static int launch_head(const HeadArgs& a, hipStream_t st) {
  constexpr int CHB = NT * 4 * 1024;
  // FROM_H0 in 16-bit storage: 8 waves x 256 columns on a three-slot ring, 4-KiB staging tiles (3 x 32 + 8 x 4 + 3 KiB at Sp = 256)
  constexpr bool WIDE8 = FROM_H0 && sizeof(E) == 2;
  constexpr int NW = WIDE8 ? 8 : 4, NSL = WIDE8 ? 3 : 2;
  const size_t lds = (size_t)(FROM_H0 ? NSL : 3) * CHB + NW * (WIDE8 ? 4096 : STG_BYTES) + (size_t)(2 * a.Sp + a.Op) * 4;
  static WaeLdsCache lds_cache;
  auto kern = head_fwd_kernel<E, NT, FROM_H0, NW, NSL>;
  if (int rc = wae_ensure_lds((const void*)kern, lds_cache, lds, "head_fwd"); rc != WAE_OK) return rc;
  const int tiles = (a.T + NW * 32 - 1) / (NW * 32);
  hipLaunchKernelGGL(kern, dim3(a.B * tiles), dim3(NW * 64), lds, st, a);
  return wae_check_launch("head_fwd");
}

static int head_validate(const wae_head_desc* d) {
  WAE_REQUIRE(d != nullptr, "head: null desc");
  WAE_REQUIRE(wae_dtype_ok(d->dtype), "head: bad dtype %d", d->dtype);
  WAE_REQUIRE(d->B > 0 && d->T > 0 && d->O > 0, "head: B,T,O must be positive");
  WAE_REQUIRE(d->Ku > 0 && d->Ku % 64 == 0, "head: Ku must be a positive multiple of 64 (got %d)", d->Ku);
  WAE_REQUIRE(d->Sp == 128 || d->Sp == 256, "head: Sp must be 128 or 256 (got %d)", d->Sp);
  WAE_REQUIRE(d->Op % 128 == 0 && d->Op >= d->O && d->Op <= 8192, "head: Op must be a multiple of 128 and >= O");
  return WAE_OK;
}

extern "C" int64_t wae_head_packed_bytes(const wae_head_desc* d) {
  if (head_validate(d) != WAE_OK) return WAE_EINVAL;
  const int ck = wae_is16(d->dtype) ? 64 : 32;
  const int mt2 = wae_is16(d->dtype) ? 2 : 1;
  const int64_t chb = (int64_t)(d->Sp / 32) * 4 * 1024;
  return (int64_t)(d->Ku / ck + (d->Sp / 32) / mt2 + (d->Op / 32) / mt2) * chb;
}

extern "C" int wae_head_fwd(const wae_head_desc* d, const void* u, const void* w_packed, const float* bias, float* logits,
                            const int32_t* target, float* nll, float* lse, void* h0_save, void* h1_save, void* stream) {
  int rc = head_validate(d);
  if (rc != WAE_OK) return rc;
  WAE_REQUIRE(u && w_packed && bias, "head: null pointer argument");
  WAE_REQUIRE(logits || (target && nll), "head: nothing to produce (logits and nll both null)");
  HeadArgs a;
  a.u = (const char*)u; a.w = (const char*)w_packed; a.bias = bias; a.logits = logits; a.target = target; a.nll = nll; a.lse = lse;
  a.h0_save = (char*)h0_save; a.h1_save = (char*)h1_save; a.B = d->B; a.T = d->T; a.Ku = d->Ku; a.Sp = d->Sp; a.Op = d->Op;
  a.O = d->O; a.scale = d->scale;
  hipStream_t st = as_stream(stream);
  const int nt = d->Sp / 32;
  if (d->dtype == WAE_BF16) return nt == 4 ? launch_head<__bf16, 4>(a, st) : launch_head<__bf16, 8>(a, st);
  if (d->dtype == WAE_F16) return nt == 4 ? launch_head<f16, 4>(a, st) : launch_head<f16, 8>(a, st);
  return nt == 4 ? launch_head<float, 4>(a, st) : launch_head<float, 8>(a, st);
}

// GEMM 1, GEMM 2 and the fused cross-entropy from a stored h0 = relu(sqrt(1/L) * skips) (B,T,Sp): the second half of wae_head_fwd for
// callers that ran the skip contraction as a wae_gemm_tm launch (mode 3).  w_tail = the packed stream from GEMM 1's first chunk on
// (w_packed + (Ku / CK) * (Sp / 32) * 4096 bytes); bias as in wae_head_fwd (its first Sp entries are not read).
extern "C" int wae_head_fwd_from_h0(const wae_head_desc* d, const void* h0, const void* w_tail, const float* bias, float* logits,
                                    const int32_t* target, float* nll, float* lse, void* h1_save, void* stream) {
  int rc = head_validate(d);
  if (rc != WAE_OK) return rc;
  WAE_REQUIRE(h0 && w_tail && bias, "head_from_h0: null pointer argument");
  WAE_REQUIRE(logits || (target && nll), "head_from_h0: nothing to produce (logits and nll both null)");
  HeadArgs a;
  a.u = (const char*)h0; a.w = (const char*)w_tail; a.bias = bias; a.logits = logits; a.target = target; a.nll = nll; a.lse = lse;
  a.h0_save = nullptr; a.h1_save = (char*)h1_save; a.B = d->B; a.T = d->T; a.Ku = 0; a.Sp = d->Sp; a.Op = d->Op;
  a.O = d->O; a.scale = d->scale;
  hipStream_t st = as_stream(stream);
  const int nt = d->Sp / 32;
  if (d->dtype == WAE_BF16) return nt == 4 ? launch_head<__bf16, 4, true>(a, st) : launch_head<__bf16, 8, true>(a, st);
  if (d->dtype == WAE_F16) return nt == 4 ? launch_head<f16, 4, true>(a, st) : launch_head<f16, 8, true>(a, st);
  return nt == 4 ? launch_head<float, 4, true>(a, st) : launch_head<float, 8, true>(a, st);
}
