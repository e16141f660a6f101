// wae_ar_generate: autoregressive (sample-by-sample) decoding, the MI355X replacement of
// Conv1d.incremental_forward (conv.py:17-62) + WaveNet.incremental_forward (wavenet.py:218-346).
//
// One persistent workgroup (512 threads) per utterance runs the whole T-step loop on the device: no per-sample
// launches, no per-layer launches.  The reference shifts a (1+(k-1)d, R) buffer by one row per layer per sample
// (an O(d) copy, conv.py:39); here every layer owns a ring of (k-1)d+1 rows in HBM/L2 that is written once and
// read at two fixed lags -- O(1) per sample.  Per layer and sample the work is two matrix-vector products
// (z = W1 [x taps ; c] + zb, [x' ; skip] = W2 u + b) streamed from L2 in a k-blocked, row-interleaved layout
// [k/EPL][row][EPL] so that consecutive lanes read consecutive 16-byte weight packets; the k range is split over
// NS thread slices whose partial sums meet in LDS.  Sampling (argmax / inverse-CDF categorical) is fused.
// Utterances are independent: a batch runs as B workgroups (replicas only; no collective).
#include "wae_common.hpp"

#define AR_THREADS 512

struct ArArgs {
  int dtype, B, T, L, R, G, S, O, Cc, Ccp, Hp, ktaps, mode, Rp;
  float scale;
  const int32_t* dil;
  const int64_t* ring_off;  // L+1 offsets (floats) into one utterance's ring arena
  float* ring;
  int64_t ring_total;
  const char* w_layers;
  int64_t layer_stride;  // bytes
  int64_t w2_off;        // bytes from a layer's base to its second matrix
  const float* bias2;    // (L, R+S)
  const float* zb;       // (B, L, 2Hp)
  const float* first_tab;   // (O, Rp)
  const float* first_bias;  // (Rp)
  const char* w_head;       // [S x S | O x S] blocked
  const float* head_bias;   // [S | O]
  const char* c_up;         // (B, T, Ccp) dtype_c
  int c_dtype;
  const int32_t* inputs;  // (B, T) teacher-forced class ids or null
  int n_forced;           // steps t < n_forced consume inputs[t] / inputs_f[t] (wavenet.py:300-305); 0 without inputs
  int init_idx;
  const float* uniforms;  // (B, T) or null
  int32_t* out_idx;       // (B, T)
  float* out_logits;      // (B, O, T) or null
  // scalar-input decoders (wavenet.py:284-285,325-333): the fed-back quantity is a float, drawn from the mixture of logistics
  int scalar;
  const float* inputs_f;  // (B, T) teacher-forced samples or null
  const float* u_mix;     // (B, T, M) uniforms of the mixture pick, or null (then inputs_f must cover every step)
  const float* u_log;     // (B, T) uniforms of the logistic draw
  float* out_f;           // (B, T) drawn samples, or null
  float log_scale_min;
  int clamp_log_scale;
  int psum_floats;        // max over the matrix-vector products of slices x padded rows (>= AR_THREADS)
};

template <typename E>
__device__ __forceinline__ void load_w(const char* p, float (&w)[ET<E>::EPL]);
template <>
__device__ __forceinline__ void load_w<float>(const char* p, float (&w)[4]) {
  const f32x4 v = *(const f32x4*)p;
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
template <>
__device__ __forceinline__ void load_w<__bf16>(const char* p, float (&w)[8]) {
  // bf16 -> fp32 is a 16-bit shift; even elements sit in the low halves of the four dwords
  const f32x4 raw = *(const f32x4*)p;
  const unsigned r[4] = {__float_as_uint(raw.x), __float_as_uint(raw.y), __float_as_uint(raw.z), __float_as_uint(raw.w)};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    w[2 * i] = __uint_as_float(r[i] << 16);
    w[2 * i + 1] = __uint_as_float(r[i] & 0xffff0000u);
  }
}

// y[r] = sum_k W[r][k] v[k] for r < rows; W blocked [k/EPL][rows_pad][EPL]; v in LDS (K padded to EPL, zero filled).
// Threads are laid out as (row = tid % RW, slice = tid / RW); partials go to psum[slice][row] (row pitch rows_pad); caller
// barriers and sums.  More rows than threads (R + S = 1024 at C5): NS = 1 and a thread walks rows tid, tid + RW, ...
template <typename E>
__device__ __forceinline__ void gemv_partial(const char* __restrict__ W, const float* v, float* psum, int rows_pad, int K, int RW,
                                             int NS) {
  constexpr int EPL = ET<E>::EPL;
  const int r0 = threadIdx.x % RW, s = threadIdx.x / RW;
  if (s >= NS) return;
  const int nkb = (K + EPL - 1) / EPL;
  const int per = (nkb + NS - 1) / NS;
  const int kb0 = s * per, kb1 = min(nkb, kb0 + per);
  for (int r = r0; r < rows_pad; r += RW) {
  float acc = 0.f;
  const char* wp = W + ((int64_t)kb0 * rows_pad + r) * 16;
  const int64_t step = (int64_t)rows_pad * 16;
  int kb = kb0;
  for (; kb + 4 <= kb1; kb += 4) {
    float w0[EPL], w1[EPL], w2[EPL], w3[EPL];
    load_w<E>(wp, w0); load_w<E>(wp + step, w1); load_w<E>(wp + 2 * step, w2); load_w<E>(wp + 3 * step, w3);
    wp += 4 * step;
    const float* vv = v + kb * EPL;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc = fmaf(w0[j], vv[j], acc);
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc = fmaf(w1[j], vv[EPL + j], acc);
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc = fmaf(w2[j], vv[2 * EPL + j], acc);
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc = fmaf(w3[j], vv[3 * EPL + j], acc);
  }
  for (; kb < kb1; ++kb) {
    float w0[EPL];
    load_w<E>(wp, w0);
    wp += step;
#pragma unroll
    for (int j = 0; j < EPL; ++j) acc = fmaf(w0[j], v[kb * EPL + j], acc);
  }
  psum[s * rows_pad + r] = acc;
  }
}

__device__ __forceinline__ float psum_total(const float* psum, int r, int rows_pad, int NS) {
  float a = 0.f;
  for (int s = 0; s < NS; ++s) a += psum[s * rows_pad + r];
  return a;
}

__device__ __forceinline__ void layout_rows(int rows, int& rows_pad, int& RW, int& NS) {
  rows_pad = (rows + 63) & ~63;
  RW = rows_pad > AR_THREADS ? AR_THREADS : rows_pad;
  NS = AR_THREADS / RW;
  if (NS < 1) NS = 1;
}

template <>
__device__ __forceinline__ void load_w<f16>(const char* p, float (&w)[8]) {
  const f16x8 v = *(const f16x8*)p;
#pragma unroll
  for (int i = 0; i < 8; ++i) w[i] = (float)v[i];
}

template <typename E>
__global__ void __launch_bounds__(AR_THREADS) ar_kernel(ArArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  constexpr int EPL = ET<E>::EPL;
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int H = p.G / 2;
  const int K1 = p.ktaps * p.R + (p.Cc > 0 ? p.Cc : 0);
  const int K1p = (K1 + EPL - 1) / EPL * EPL;
  const int Hk = (H + EPL - 1) / EPL * EPL;
  const int Sk = (p.S + EPL - 1) / EPL * EPL;
  // LDS carve (floats)
  float* vbuf = sm;                       // K1p
  float* xbuf = vbuf + K1p;               // R
  float* ubuf = xbuf + p.R;               // Hk
  float* skipb = ubuf + Hk;               // Sk   (also the head's h0)
  float* hbuf = skipb + Sk;               // Sk   (h1)
  float* lbuf = hbuf + Sk;                // O    logits
  float* psum = lbuf + ((p.O + 3) & ~3);  // p.psum_floats
  int* ibuf = (int*)(psum + p.psum_floats);     // [0] = current input id

  float* ring = p.ring + (int64_t)b * p.ring_total;
  const float* zb_b = p.zb + (int64_t)b * p.L * 2 * p.Hp;
  int gp, gRW, gNS, wp_, wRW, wNS, sp_, sRW, sNS, op_, oRW, oNS;
  layout_rows(p.G, gp, gRW, gNS);
  layout_rows(p.R + p.S, wp_, wRW, wNS);
  layout_rows(p.S, sp_, sRW, sNS);
  layout_rows(p.O, op_, oRW, oNS);

  for (int i = tid; i < K1p; i += AR_THREADS) vbuf[i] = 0.f;
  for (int i = tid; i < Hk; i += AR_THREADS) ubuf[i] = 0.f;
  for (int i = tid; i < Sk; i += AR_THREADS) { skipb[i] = 0.f; hbuf[i] = 0.f; }
  float* fcur = (float*)(ibuf + 1);   // scalar input: the current input value
  if (tid == 0) {
    ibuf[0] = (p.inputs && p.n_forced > 0) ? p.inputs[(int64_t)b * p.T] : p.init_idx;
    fcur[0] = (p.inputs_f && p.n_forced > 0) ? p.inputs_f[(int64_t)b * p.T] : 0.f;     // wavenet.py:284-285: the start value is zero
  }
  __syncthreads();

  for (int t = 0; t < p.T; ++t) {
    // ---- first conv: one-hot input == column gather (wavenet.py:311); scalar input: w * x + b ----------------
    const int cur = ibuf[0];
    if (cur < 0) {
      // dense feedback (quantize=False, wavenet.py:335-338 skipped): the previous step's probability / logit vector, still in
      // lbuf, is the decoder input -- first_conv on a dense (1, O) row (wavenet.py:311)
      for (int r = tid; r < p.R; r += AR_THREADS) {
        float acc = p.first_bias[r];
        for (int o = 0; o < p.O; ++o) acc = fmaf(p.first_tab[(int64_t)o * p.Rp + r], lbuf[o], acc);
        xbuf[r] = acc;
      }
    } else {
      for (int r = tid; r < p.R; r += AR_THREADS)
        xbuf[r] = p.scalar ? fmaf(p.first_tab[r], fcur[0], p.first_bias[r])
                           : p.first_tab[(int64_t)cur * p.Rp + r] + p.first_bias[r];
    }
    for (int cc = AR_THREADS - 1 - tid; cc < p.Cc; cc += AR_THREADS) {   // local conditioning of this step -> tail of the operand vector
      const int64_t ci = ((int64_t)b * p.T + t) * p.Ccp + cc;
      vbuf[p.ktaps * p.R + cc] = p.c_dtype == WAE_BF16 ? (float)((const __bf16*)p.c_up)[ci] : (p.c_dtype == WAE_F16 ? (float)((const f16*)p.c_up)[ci] : ((const float*)p.c_up)[ci]);
    }
    for (int i = tid; i < p.S; i += AR_THREADS) skipb[i] = 0.f;
    __syncthreads();

    for (int l = 0; l < p.L; ++l) {
      const int d = p.dil[l];
      const int64_t roff = p.ring_off[l];
      const int rlen = (p.ktaps - 1) * d + 1;
      float* rl = ring + roff;
      // ---- assemble [x[t-(k-1)d] .. x[t]] and push x[t] into the layer's ring (conv.py:35-44, O(1)) -------
      for (int i = tid; i < p.ktaps * p.R; i += AR_THREADS) {
        const int tap = i / p.R, ch = i - tap * p.R;
        const int tt = t - (p.ktaps - 1 - tap) * d;
        float v;
        if (tap == p.ktaps - 1) {
          v = xbuf[ch];
          rl[(int64_t)(t % rlen) * p.R + ch] = v;
        } else {
          v = tt >= 0 ? rl[(int64_t)(tt % rlen) * p.R + ch] : 0.f;   // zero history before the clip starts
        }
        vbuf[i] = v;
      }
      __syncthreads();
      const char* wl = p.w_layers + (int64_t)l * p.layer_stride;
      gemv_partial<E>(wl, vbuf, psum, gp, K1, gRW, gNS);
      __syncthreads();
      // ---- gate (modules.py:138-154): thread h owns channel h ------------------------------------------------
      for (int hh = tid; hh < H; hh += AR_THREADS) {
        const float* zbl = zb_b + (int64_t)l * 2 * p.Hp;
        const float a = psum_total(psum, hh, gp, gNS) + zbl[hh];
        const float g = psum_total(psum, H + hh, gp, gNS) + zbl[p.Hp + hh];
        ubuf[hh] = tanhf(a) * (1.f / (1.f + expf(-g)));
      }
      __syncthreads();
      gemv_partial<E>(wl + p.w2_off, ubuf, psum, wp_, H, wRW, wNS);
      __syncthreads();
      // ---- residual + skip (modules.py:157-162, wavenet.py:315) ------------------------------------------------
      const float* b2 = p.bias2 + (int64_t)l * (p.R + p.S);
      for (int i = tid; i < p.R + p.S; i += AR_THREADS) {
        const float y = psum_total(psum, i, wp_, wNS) + b2[i];
        if (i < p.R) xbuf[i] = (y + xbuf[i]) * 0.70710678118654752440f;
        else skipb[i - p.R] += y;
      }
      __syncthreads();
    }
    // ---- head (wavenet.py:316-322) ------------------------------------------------------------------------------
    for (int i = tid; i < p.S; i += AR_THREADS) skipb[i] = fmaxf(skipb[i] * p.scale, 0.f);
    __syncthreads();
    gemv_partial<E>(p.w_head, skipb, psum, sp_, p.S, sRW, sNS);
    __syncthreads();
    for (int i = tid; i < p.S; i += AR_THREADS) hbuf[i] = fmaxf(psum_total(psum, i, sp_, sNS) + p.head_bias[i], 0.f);
    __syncthreads();
    gemv_partial<E>(p.w_head + (int64_t)((p.S + EPL - 1) / EPL) * sp_ * 16, hbuf, psum, op_, p.S, oRW, oNS);
    __syncthreads();
    for (int i = tid; i < p.O; i += AR_THREADS) {
      const float y = psum_total(psum, i, op_, oNS) + p.head_bias[p.S + i];
      lbuf[i] = y;
      if (p.out_logits && p.mode != 3) p.out_logits[((int64_t)b * p.O + i) * p.T + t] = y;
    }
    __syncthreads();
    if (p.mode == 3) {
      // softmax=True, quantize=False: the probability vector is the step's output and the next step's input
      float mx = -INFINITY, den = 0.f;
      for (int i = 0; i < p.O; ++i) mx = fmaxf(mx, lbuf[i]);
      for (int i = 0; i < p.O; ++i) den += expf(lbuf[i] - mx);
      __syncthreads();
      for (int i = tid; i < p.O; i += AR_THREADS) {
        const float pr = expf(lbuf[i] - mx) / den;
        lbuf[i] = pr;
        if (p.out_logits) p.out_logits[((int64_t)b * p.O + i) * p.T + t] = pr;
      }
      __syncthreads();
    }
    // ---- next input: teacher forcing / greedy / categorical draw (wavenet.py:300-338) -------------------------
    if (tid == 0 && p.scalar) {
      // sample_from_discretized_mix_logistic (mixture.py:118-156) on caller-supplied uniforms: Gumbel-max mixture pick,
      // logistic draw, clamp to [-1, 1] -- the arithmetic of dmol_sample_kernel (csrc/loss.hip)
      float xs = 0.f;
      if (p.u_mix) {
        const int M = p.O / 3;
        const float* um = p.u_mix + ((int64_t)b * p.T + t) * M;
        float best = -INFINITY;
        int arg = 0;
        for (int i = 0; i < M; ++i) {
          const float v = lbuf[i] - logf(-logf(um[i]));
          if (v > best) { best = v; arg = i; }
        }
        const float mu = lbuf[M + arg];
        float ls = lbuf[2 * M + arg];
        if (p.clamp_log_scale) ls = fmaxf(ls, p.log_scale_min);
        const float u = p.u_log[(int64_t)b * p.T + t];
        xs = fminf(fmaxf(mu + expf(ls) * (logf(u) - logf(1.f - u)), -1.f), 1.f);
        if (p.out_f) p.out_f[(int64_t)b * p.T + t] = xs;
      }
      fcur[0] = (p.inputs_f && t + 1 < p.n_forced) ? p.inputs_f[(int64_t)b * p.T + t + 1] : xs;
    }
    if (tid == 0 && !p.scalar) {
      int nxt;
      float mx = -INFINITY;
      int am = 0;
      for (int i = 0; i < p.O; ++i)
        if (lbuf[i] > mx) { mx = lbuf[i]; am = i; }
      int produced = am;
      if (p.mode == 2) {
        // softmax in fp32 (F.softmax), then inverse CDF over a double cumulative sum
        float den = 0.f;
        for (int i = 0; i < p.O; ++i) den += expf(lbuf[i] - mx);
        double tot = 0.0;
        for (int i = 0; i < p.O; ++i) tot += (double)(expf(lbuf[i] - mx) / den);
        const double thr = (double)p.uniforms[(int64_t)b * p.T + t] * tot;
        double c = 0.0;
        int cnt = 0;
        for (int i = 0; i < p.O; ++i) {
          c += (double)(expf(lbuf[i] - mx) / den);
          if (c < thr) ++cnt;
        }
        produced = min(cnt, p.O - 1);
      }
      p.out_idx[(int64_t)b * p.T + t] = produced;
      if (p.inputs && t + 1 < p.n_forced) nxt = p.inputs[(int64_t)b * p.T + t + 1];
      else nxt = p.mode >= 3 ? -1 : produced;
      ibuf[0] = nxt;
    }
    __syncthreads();
  }
}

static int ar_launch(const wae_ar_desc* d, const int32_t* dilations, const int64_t* ring_off, float* ring, int64_t ring_total,
                     const void* w_layers, int64_t layer_stride_bytes, int64_t w2_off_bytes, const float* bias2, const float* zb,
                     const float* first_tab, const float* first_bias, const void* w_head, const float* head_bias, const void* c_up,
                     int32_t c_dtype, const int32_t* inputs, const float* uniforms, int32_t* out_idx, float* out_logits,
                     const float* inputs_f, const float* u_mix, const float* u_log, float* out_f, float log_scale_min,
                     int clamp_log_scale, void* stream) {
  ArArgs a;
  a.dtype = d->dtype; a.B = d->B; a.T = d->T; a.L = d->L; a.R = d->R; a.G = d->G; a.S = d->S; a.O = d->O; a.Cc = d->Cc;
  a.Ccp = d->Ccp; a.Hp = d->Hp; a.ktaps = d->ktaps; a.mode = d->mode; a.Rp = d->Rp; a.scale = d->scale; a.dil = dilations;
  a.ring_off = ring_off; a.ring = ring; a.ring_total = ring_total; a.w_layers = (const char*)w_layers;
  a.layer_stride = layer_stride_bytes; a.w2_off = w2_off_bytes; a.bias2 = bias2; a.zb = zb; a.first_tab = first_tab;
  a.first_bias = first_bias; a.w_head = (const char*)w_head; a.head_bias = head_bias; a.c_up = (const char*)c_up;
  a.c_dtype = c_dtype; a.inputs = inputs; a.init_idx = d->init_idx;
  a.n_forced = (inputs || inputs_f) ? (d->n_forced > 0 && d->n_forced < d->T ? d->n_forced : d->T) : 0; a.uniforms = uniforms; a.out_idx = out_idx;
  a.out_logits = out_logits; a.scalar = d->scalar_input ? 1 : 0; a.inputs_f = inputs_f; a.u_mix = u_mix; a.u_log = u_log;
  a.out_f = out_f; a.log_scale_min = log_scale_min; a.clamp_log_scale = clamp_log_scale;
  const int epl = wae_is16(d->dtype) ? 8 : 4;
  const int H = d->G / 2;
  auto ru = [](int x, int m) { return (x + m - 1) / m * m; };
  int psz = AR_THREADS;
  for (int rows : {d->G, d->R + d->S, d->S, d->O}) psz = psz > ru(rows, 64) ? psz : ru(rows, 64);
  a.psum_floats = psz;
  const size_t lds = sizeof(float) * (size_t)(ru(d->ktaps * d->R + (d->Cc > 0 ? d->Cc : 0), epl) + d->R + ru(H, epl) +
                                              2 * ru(d->S, epl) + ru(d->O, 4) + psz + 4);
  hipStream_t st = as_stream(stream);
  if (d->dtype == WAE_BF16) {
    (void)hipFuncSetAttribute((const void*)ar_kernel<__bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ar_kernel<__bf16>, dim3(d->B), dim3(AR_THREADS), lds, st, a);
  } else if (d->dtype == WAE_F16) {
    (void)hipFuncSetAttribute((const void*)ar_kernel<f16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ar_kernel<f16>, dim3(d->B), dim3(AR_THREADS), lds, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)ar_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(ar_kernel<float>, dim3(d->B), dim3(AR_THREADS), lds, st, a);
  }
  return wae_check_launch("ar_generate");
}

extern "C" int wae_ar_generate(const wae_ar_desc* d, const int32_t* dilations, const int64_t* ring_off, float* ring,
                               int64_t ring_total, const void* w_layers, int64_t layer_stride_bytes, int64_t w2_off_bytes,
                               const float* bias2, const float* zb, const float* first_tab, const float* first_bias,
                               const void* w_head, const float* head_bias, const void* c_up, int32_t c_dtype,
                               const int32_t* inputs, const float* uniforms, int32_t* out_idx, float* out_logits,
                               void* stream) {
  WAE_REQUIRE(d && dilations && ring_off && ring && w_layers && bias2 && zb && first_tab && first_bias && w_head && head_bias &&
                  out_idx, "ar_generate: null pointer argument");
  WAE_REQUIRE(wae_dtype_ok(d->dtype), "ar_generate: bad dtype");
  WAE_REQUIRE(d->B > 0 && d->T > 0 && d->L > 0 && d->R > 0 && d->G > 0 && d->G % 2 == 0 && d->S > 0 && d->O > 0,
              "ar_generate: bad sizes");
  WAE_REQUIRE(d->Cc <= 0 || c_up, "ar_generate: Cc > 0 but c_up is null");
  WAE_REQUIRE(d->mode >= 0 && d->mode <= 4,
              "ar_generate: mode must be 0 (logits), 1 (argmax), 2 (sample), 3 (feed probabilities back) or 4 (feed logits back)");
  WAE_REQUIRE(d->mode != 2 || uniforms, "ar_generate: sample mode needs uniforms");
  WAE_REQUIRE(d->mode != 0 || (inputs && (d->n_forced <= 0 || d->n_forced >= d->T)), "ar_generate: mode 0 needs inputs for every step");
  WAE_REQUIRE(d->mode < 3 || out_logits, "ar_generate: modes 3 / 4 return their vectors through out_logits");
  WAE_REQUIRE(!d->scalar_input, "ar_generate: scalar-input decoders go through wae_ar_generate_scalar");
  // the start class indexes the first-conv table: wavenet.py:288 sets class 127, an IndexError there when O <= 127
  WAE_REQUIRE(inputs || (d->init_idx >= 0 && d->init_idx < d->O), "ar_generate: init_idx %d is not a class (O = %d)", d->init_idx, d->O);
  return ar_launch(d, dilations, ring_off, ring, ring_total, w_layers, layer_stride_bytes, w2_off_bytes, bias2, zb, first_tab,
                   first_bias, w_head, head_bias, c_up, c_dtype, inputs, uniforms, out_idx, out_logits, nullptr, nullptr, nullptr,
                   nullptr, -7.0f, 0, stream);
}

extern "C" int wae_ar_generate_scalar(const wae_ar_desc* d, const int32_t* dilations, const int64_t* ring_off, float* ring,
                                      int64_t ring_total, const void* w_layers, int64_t layer_stride_bytes, int64_t w2_off_bytes,
                                      const float* bias2, const float* zb, const float* first_tab, const float* first_bias,
                                      const void* w_head, const float* head_bias, const void* c_up, int32_t c_dtype,
                                      const float* inputs_f, const float* u_mix, const float* u_log, float log_scale_min,
                                      int32_t clamp_log_scale, float* out_samples, float* out_params, void* stream) {
  WAE_REQUIRE(d && dilations && ring_off && ring && w_layers && bias2 && zb && first_tab && first_bias && w_head && head_bias,
              "ar_generate_scalar: null pointer argument");
  WAE_REQUIRE(wae_dtype_ok(d->dtype), "ar_generate_scalar: bad dtype");
  WAE_REQUIRE(d->scalar_input && d->O > 0 && d->O % 3 == 0, "ar_generate_scalar: needs a scalar-input decoder with 3M output channels");
  WAE_REQUIRE(d->B > 0 && d->T > 0 && d->L > 0 && d->R > 0 && d->G > 0 && d->G % 2 == 0 && d->S > 0, "ar_generate_scalar: bad sizes");
  WAE_REQUIRE(d->Cc <= 0 || c_up, "ar_generate_scalar: Cc > 0 but c_up is null");
  WAE_REQUIRE((inputs_f && (d->n_forced <= 0 || d->n_forced >= d->T)) || (u_mix && u_log),
              "ar_generate_scalar: needs teacher-forced inputs for every step or the uniforms of the draws");
  WAE_REQUIRE(!u_mix == !u_log, "ar_generate_scalar: u_mix and u_log come together");
  WAE_REQUIRE(!out_samples || u_mix, "ar_generate_scalar: samples need the uniforms");
  WAE_REQUIRE(out_samples || out_params, "ar_generate_scalar: no output requested");
  return ar_launch(d, dilations, ring_off, ring, ring_total, w_layers, layer_stride_bytes, w2_off_bytes, bias2, zb, first_tab,
                   first_bias, w_head, head_bias, c_up, c_dtype, nullptr, nullptr, nullptr, out_params, inputs_f, u_mix, u_log,
                   out_samples, log_scale_min, clamp_log_scale, stream);
}
