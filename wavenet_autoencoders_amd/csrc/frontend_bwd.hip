// Backward of the low-rate front end: conditioning upsample stages (upsample.py:19-21,39-46), plain / ReLU /
// residual Conv1d blocks of the encoder and conv_in (vqvae_model.py:17-23,50; upsample.py:77-78) and the vector
// quantizer (vector_quantization.py:38-45).  These run at 1/160 .. 1/640 of the audio rate (the last upsample stage
// excepted) and are plain fp32 VALU kernels; weight gradients are accumulated with fp32 atomics into the
// effective-weight gradient arena.
#include "wae_common.hpp"

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- upsample stage: out[t] = sum_j w[j] * in[(t + j - s) / s]  (0 <= t + j - s < Tin*s) ---------------------------------
// din[i] = sum_{u in [is, is+s)} sum_j w[j] dout[u - j + s]: every dout[t], t = is + d with d in [-s, 2s), enters with the
// sum of the taps that reach it, coef[d + s] = sum_{j = max(0, s-d)}^{min(2s, 2s-1-d)} w[j] (formed once per block):
// 3s loads per frame instead of s(2s+1).
#define UPW_MAXTAPS 33
__device__ __forceinline__ void upsample_stage_bwd_in_body(const float* __restrict__ dout, const float* __restrict__ w,
                                                           float* __restrict__ din, int C, int Tin, int s, int bx, int by, int bz) {
  __shared__ float coef[3 * (UPW_MAXTAPS / 2)];
  if (threadIdx.x < 3 * s) {
    const int d = (int)threadIdx.x - s;
    float a = 0.f;
    for (int j = max(0, s - d); j <= min(2 * s, 2 * s - 1 - d); ++j) a += w[j];
    coef[threadIdx.x] = a;
  }
  __syncthreads();
  const int i = bx * 256 + threadIdx.x;
  const int c = by, b = bz;
  if (i >= Tin) return;
  const int Tout = Tin * s;
  const float* dr = dout + ((int64_t)b * C + c) * Tout;
  float acc = 0.f;
  const int tb = i * s - s;
  for (int k = 0; k < 3 * s; ++k) {
    const int t = tb + k;
    if (t >= 0 && t < Tout) acc = fmaf(coef[k], dr[t], acc);
  }
  din[((int64_t)b * C + c) * Tin + i] = acc;
}
// dw[j] = sum_{row,t} dout[row][t] * in[row][(t + j - s) / s]  for 0 <= t + j - s < Tout   (autograd of upsample.py:51-66).
// With t = q s + rem the input index is q - 1 + (rem + j) / s, and (rem + j) / s in {0, 1, 2} does not depend on q: the
// 2s+1 tap sums are regroupings of the 3s correlations A[rem][o] = sum_q dout[q s + rem] in[q - 1 + o] (in[-1] = in[Tin] = 0
// reproduce the zero padding), dw[j] = sum_rem A[rem][(rem + j) / s].  A thread walks frames q: s contiguous dout loads,
// three in loads, 3s FMAs (the tap-by-tap form needed s(2s+1) selects + FMAs and took 0.1 ms per step).  A block walks
// whole (clip, channel) rows; at most 256 blocks leave their 2s+1 sums as atomics.  S = 0: any s <= 16, tap-by-tap.
template <int S>
__device__ __forceinline__ void upsample_stage_bwd_w_body(const float* __restrict__ dout, const float* __restrict__ in,
                                                          float* __restrict__ dw, int BC, int Tin, int s_rt, int bx, int nbx) {
  const int s = S > 0 ? S : s_rt;
  constexpr int NTAP = S > 0 ? 2 * S + 1 : UPW_MAXTAPS;
  const int Tout = Tin * s, ntap = 2 * s + 1;
  float acc[NTAP];
#pragma unroll
  for (int j = 0; j < NTAP; ++j) acc[j] = 0.f;
  if constexpr (S > 0) {
    float A[S][3];
#pragma unroll
    for (int r = 0; r < S; ++r) A[r][0] = A[r][1] = A[r][2] = 0.f;
    for (int row = bx; row < BC; row += nbx) {
      const float* drow = dout + (int64_t)row * Tout;
      const float* irow = in + (int64_t)row * Tin;
      for (int q = threadIdx.x; q < Tin; q += 256) {
        float g[S];
#pragma unroll
        for (int r = 0; r < S; ++r) g[r] = drow[q * S + r];
        const float i0 = q >= 1 ? irow[q - 1] : 0.f, i1 = irow[q], i2 = q + 1 < Tin ? irow[q + 1] : 0.f;
#pragma unroll
        for (int r = 0; r < S; ++r) {
          A[r][0] = fmaf(g[r], i0, A[r][0]);
          A[r][1] = fmaf(g[r], i1, A[r][1]);
          A[r][2] = fmaf(g[r], i2, A[r][2]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
#pragma unroll
      for (int r = 0; r < S; ++r) acc[j] += A[r][(r + j) / S];
  } else {
    for (int row = bx; row < BC; row += nbx) {
      const float* drow = dout + (int64_t)row * Tout;
      const float* irow = in + (int64_t)row * Tin;
      for (int t = threadIdx.x; t < Tout; t += 256) {
        const float g = drow[t];
        const int q = t / s, rem = t - q * s;
        const float i0 = q >= 1 ? irow[q - 1] : 0.f, i1 = irow[q], i2 = q + 1 < Tin ? irow[q + 1] : 0.f;
#pragma unroll
        for (int j = 0; j < NTAP; ++j)
          if (j < ntap) {
            const int k = rem + j;                       // (t + j - s) / s = q - 1 + k / s,  k / s in {0, 1, 2}
            const int u = t + j - s;
            const float v = k < s ? i0 : (k < 2 * s ? i1 : i2);
            if (u >= 0 && u < Tout) acc[j] = fmaf(g, v, acc[j]);
          }
      }
    }
  }
  __shared__ float part[4][UPW_MAXTAPS];
#pragma unroll
  for (int j = 0; j < NTAP; ++j)
    if (j < ntap) {
      const float v = wave_sum_f(acc[j]);
      if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][j] = v;
    }
  __syncthreads();
  if (threadIdx.x < ntap) atomicAdd(dw + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}
// the input gradient and the FIR's weight gradient of a stage read the same dout and depend on nothing of each other: ONE launch, the first
// nxx * C * B workgroups of a flat grid on the input gradient, the rest on the weight gradient (as two launches on one stream they ran one
// after the other, 6-15 us each)
template <int S>
__global__ void __launch_bounds__(256) upsample_stage_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ in,
                                                                const float* __restrict__ w, float* __restrict__ din,
                                                                float* __restrict__ dw, int B, int C, int Tin, int s, int nxx, int nbw) {
  const int id = blockIdx.x, nx = nxx * C * B;
  if (id < nx) upsample_stage_bwd_in_body(dout, w, din, C, Tin, s, id % nxx, (id / nxx) % C, id / (nxx * C));
  else upsample_stage_bwd_w_body<S>(dout, in, dw, B * C, Tin, s, id - nx, nbw);
}
extern "C" int wae_upsample_stage_bwd(const float* dout, const float* in, const float* w, float* din, float* dw, int32_t B,
                                      int32_t C, int32_t Tin, int32_t s, void* stream) {
  WAE_REQUIRE(dout && in && w && din && dw && B > 0 && C > 0 && Tin > 0 && s > 0, "upsample_stage_bwd: bad arguments");
  WAE_REQUIRE(2 * s + 1 <= UPW_MAXTAPS, "upsample_stage_bwd: scale %d > %d is not supported", s, (UPW_MAXTAPS - 1) / 2);
  hipStream_t st = as_stream(stream);
  const int rows = B * C, nbw = rows < 256 ? rows : 256, nxx = (Tin + 255) / 256;
  const dim3 grid(nxx * C * B + nbw);
  if (s == 4) hipLaunchKernelGGL(upsample_stage_bwd_kernel<4>, grid, dim3(256), 0, st, dout, in, w, din, dw, B, C, Tin, s, nxx, nbw);
  else if (s == 5) hipLaunchKernelGGL(upsample_stage_bwd_kernel<5>, grid, dim3(256), 0, st, dout, in, w, din, dw, B, C, Tin, s, nxx, nbw);
  else if (s == 8) hipLaunchKernelGGL(upsample_stage_bwd_kernel<8>, grid, dim3(256), 0, st, dout, in, w, din, dw, B, C, Tin, s, nxx, nbw);
  else hipLaunchKernelGGL(upsample_stage_bwd_kernel<0>, grid, dim3(256), 0, st, dout, in, w, din, dw, B, C, Tin, s, nxx, nbw);
  return wae_check_launch("upsample_stage_bwd");
}

// ---- Conv1d (+ReLU) (+residual) backward ------------------------------------------------------------------------------
// dpre = dy * [relu ? (y - (residual ? x : 0)) > 0 : 1]
__device__ __forceinline__ float conv_dpre(const float* dy, const float* y, const float* x, int64_t yi, int64_t xi_same, int relu,
                                           int residual) {
  float g = dy[yi];
  if (relu) {
    const float act = residual ? y[yi] - x[xi_same] : y[yi];
    if (!(act > 0.f)) g = 0.f;
  }
  return g;
}
__global__ void __launch_bounds__(256) conv_bwd_x_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ y, const float* __restrict__ dy,
                                                         float* __restrict__ dx, int Cin, int Tin, int Cout, int Tout, int k,
                                                         int stride, int pad, int relu, int residual) {
  const int ti = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ci = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (ci >= Cin || ti >= Tin) return;
  const float* xb = x + (int64_t)b * Cin * Tin;
  const float* yb = y ? y + (int64_t)b * Cout * Tout : nullptr;
  const float* dyb = dy + (int64_t)b * Cout * Tout;
  float acc = 0.f;
  for (int j = 0; j < k; ++j) {
    const int num = ti + pad - j;
    if (num < 0 || num % stride != 0) continue;
    const int to = num / stride;
    if (to >= Tout) continue;
    for (int co = 0; co < Cout; ++co) {
      const float g = conv_dpre(dyb, yb, xb, (int64_t)co * Tout + to, (int64_t)co * Tin + to, relu, residual);
      acc = fmaf(w[((int64_t)co * Cin + ci) * k + j], g, acc);
    }
  }
  if (residual) acc += dyb[(int64_t)ci * Tout + ti];
  dx[((int64_t)b * Cin + ci) * Tin + ti] = acc;
}
__global__ void __launch_bounds__(64) conv_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ dy, float* __restrict__ dw,
                                                        float* __restrict__ dbias, int B, int Cin, int Tin, int Cout, int Tout,
                                                        int k, int stride, int pad, int relu, int residual) {
  // one wave per (co, ci, j); lanes stride over (b, to)
  const int j = blockIdx.x % k, ci = (blockIdx.x / k) % Cin, co = blockIdx.x / (k * Cin);
  float acc = 0.f, accb = 0.f;
  for (int e = threadIdx.x; e < B * Tout; e += 64) {
    const int b = e / Tout, to = e % Tout;
    const float* xb = x + (int64_t)b * Cin * Tin;
    const float g = conv_dpre(dy + (int64_t)b * Cout * Tout, y ? y + (int64_t)b * Cout * Tout : nullptr, xb,
                              (int64_t)co * Tout + to, (int64_t)co * Tin + to, relu, residual);
    const int ti = to * stride + j - pad;
    if (ti >= 0 && ti < Tin) acc = fmaf(g, xb[(int64_t)ci * Tin + ti], acc);
    accb += g;
  }
  acc = wave_sum_f(acc);
  accb = wave_sum_f(accb);
  if (threadIdx.x == 0) {
    atomicAdd(dw + ((int64_t)co * Cin + ci) * k + j, acc);
    if (dbias && ci == 0 && j == 0) atomicAdd(dbias + co, accb);
  }
}
// Tiled forms (round 2; see csrc/misc.hip: enc_conv_fwd_tiled_kernel).  The one-thread-per-output kernels above took 170 us
// (input gradient) and 43 us (weight gradient) per encoder block at hps/vqwae.json's training shapes.
#define ECB_T 32
#define ECB_NS 8
#define ECB_CH 256
#define ECB_WB 8
#define ECB_RT 8    // (clip, step tile) pairs staged per round of the weight-gradient kernel
// dx[b][ci][ti] = sum_{co, j} w[co][ci][j] * dpre[b][co][(ti + PAD - j) / S]: block = 32 input channels x 32 input steps of one
// clip, thread = (channel, slice of the co reduction); dpre windows of 256 output channels are staged in LDS.  ti0 is a multiple
// of 32, so which (ti, j) pairs hit a whole output step, and where it sits in the staged window, is known at compile time.
template <int K, int S, int PAD, int ET>
__device__ __forceinline__ void conv_bwd_x_tiled_body(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ y, const float* __restrict__ dy,
                                                      float* __restrict__ dx, int Cin, int Tin, int Cout, int Tout, int relu,
                                                      int residual, int bx, int by, int bz) {
  extern __shared__ float sm[];
  constexpr int R0 = (((PAD - K + 1) % S) + S) % S;        // (ti0 + PAD - K + 1) mod S for ti0 = 0 mod S
  constexpr int BASE = K - 1 + R0;                         // ti_l - j + BASE = S * (to - tb) when that is a whole step
  constexpr int TWIN = (ET - 1 + BASE) / S + 1;
  constexpr int TP = (TWIN + 3) & ~3;
  const int b = bz, ci0 = by * ECB_T, ti0 = bx * ET;
  const int tb = (ti0 + PAD - (K - 1) - R0) / S;           // exact: the numerator is a multiple of S (may be negative)
  const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int ci = ci0 + col;
  float acc[ET];
#pragma unroll
  for (int i = 0; i < ET; ++i) acc[i] = 0.f;
  const float* xb = x + (int64_t)b * Cin * Tin;
  const float* yb = y ? y + (int64_t)b * Cout * Tout : nullptr;
  const float* dyb = dy + (int64_t)b * Cout * Tout;
  for (int c0 = 0; c0 < Cout; c0 += ECB_CH) {
    const int nc = min(ECB_CH, Cout - c0);
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < nc * TP; i0 += 256 * 8) {   // batches of 8 independent loads per thread
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int co = i / TP, tl = i - co * TP, to = tb + tl;
        v[u] = (i < nc * TP && tl < TWIN && to >= 0 && to < Tout)
                   ? conv_dpre(dyb, yb, xb, (int64_t)(c0 + co) * Tout + to, (int64_t)(c0 + co) * Tin + to, relu, residual)
                   : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u * 256 < nc * TP) sm[i0 + u * 256] = v[u];
    }
    __syncthreads();
    const int per = (nc + ECB_NS - 1) / ECB_NS;
    const int ca = sl * per, cb = min(nc, ca + per);
    if (ci < Cin) {
      // weights of ECB_WB output channels requested together, the NEXT batch under this one's FMAs (one exposed round trip per chunk)
      float wv[ECB_WB][K], wn[ECB_WB][K];
      auto fetch = [&](float (&dst)[ECB_WB][K], int cc) {
#pragma unroll
        for (int u = 0; u < ECB_WB; ++u)
#pragma unroll
          for (int j = 0; j < K; ++j) dst[u][j] = cc + u < cb ? w[((int64_t)(c0 + cc + u) * Cin + ci) * K + j] : 0.f;
      };
      if (ca < cb) fetch(wv, ca);
      for (int cc = ca; cc < cb; cc += ECB_WB) {
        if (cc + ECB_WB < cb) fetch(wn, cc + ECB_WB);
#pragma unroll
        for (int u = 0; u < ECB_WB; ++u) {
          const f32x4* gr4 = (const f32x4*)(sm + min(cc + u, cb - 1) * TP);
          float gw[TP];
#pragma unroll
          for (int q = 0; q < TP / 4; ++q) {
            const f32x4 v = gr4[q];
            gw[4 * q] = v.x; gw[4 * q + 1] = v.y; gw[4 * q + 2] = v.z; gw[4 * q + 3] = v.w;
          }
#pragma unroll
          for (int j = 0; j < K; ++j)
#pragma unroll
            for (int t = 0; t < ET; ++t)   // (t, j, BASE, S are compile-time after unrolling: the test folds away)
              if ((t - j + BASE) >= 0 && (t - j + BASE) % S == 0) acc[t] = fmaf(wv[u][j], gw[(t - j + BASE) / S], acc[t]);
        }
#pragma unroll
        for (int u = 0; u < ECB_WB; ++u)
#pragma unroll
          for (int j = 0; j < K; ++j) wv[u][j] = wn[u][j];
      }
    }
  }
  __syncthreads();
  float* red = sm;
#pragma unroll
  for (int t = 0; t < ET; ++t) red[(sl * ET + t) * 33 + col] = acc[t];
  __syncthreads();
  for (int o = threadIdx.x; o < ET * ECB_T; o += 256) {
    const int tl = o % ET, cl = o / ET;
    const int oc = ci0 + cl, ot = ti0 + tl;
    if (oc >= Cin || ot >= Tin) continue;
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < ECB_NS; ++q) v += red[(q * ET + tl) * 33 + cl];
    if (residual) v += dyb[(int64_t)oc * Tout + ot];
    dx[((int64_t)b * Cin + oc) * Tin + ot] = v;
  }
}
// dw[co][ci][j] += sum_{b, to} dpre[b][co][to] * x[b][ci][to*S + j - pad] (+ dbias[co] += sum dpre): block = 32 output x 32 input
// channels over every clip and step (the unique owner of its outputs: plain read-modify-write), thread = (co, 4 input channels).
template <int K, int S, int ET>
__device__ __forceinline__ void conv_bwd_w_tiled_body(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ dy, float* __restrict__ dw,
                                                      float* __restrict__ dbias, int B, int Cin, int Tin, int Cout, int Tout,
                                                      int pad, int relu, int residual, int bx, int by) {
  constexpr int WIN = (ET - 1) * S + K;
  constexpr int WP = (WIN + 3) & ~3, GP = ET + 4;   // row pitches: whole 16-byte reads
  constexpr int GSZ = ECB_T * GP, XSZ = ECB_T * WP;
  extern __shared__ float sm[];     // ECB_RT x (dpre tile [32 co][GP] + x window [32 ci][WP]): one round of (clip, step tile) pairs
  const int co0 = bx * ECB_T, ci0 = by * ECB_T;
  const int col = threadIdx.x & 31, cg = threadIdx.x >> 5;
  float acc[4][K];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < K; ++j) acc[c][j] = 0.f;
  float accb = 0.f;
  const int tpc = (Tout + ET - 1) / ET;      // step tiles per clip
  const int ntile = B * tpc;
  for (int r0 = 0; r0 < ntile; r0 += ECB_RT) {
    const int nr = min(ECB_RT, ntile - r0);
    __syncthreads();
    // every global load of the round is issued before the first LDS store of the round is needed: one round trip per round
    for (int i0 = threadIdx.x; i0 < nr * ECB_T * ET; i0 += 256 * 8) {   // batches of 8 independent loads per thread
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int r = i / (ECB_T * ET), e = i - r * (ECB_T * ET);
        const int tl = e % ET, cl = e / ET;
        const int b = min((r0 + r) / tpc, B - 1), to = ((r0 + r) % tpc) * ET + tl, co = co0 + cl;
        const float* xb = x + (int64_t)b * Cin * Tin;
        v[u] = (i < nr * ECB_T * ET && co < Cout && to < Tout)
                   ? conv_dpre(dy + (int64_t)b * Cout * Tout, y ? y + (int64_t)b * Cout * Tout : nullptr, xb, (int64_t)co * Tout + to,
                               (int64_t)co * Tin + to, relu, residual)
                   : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int r = i / (ECB_T * ET), e = i - r * (ECB_T * ET);
        if (i < nr * ECB_T * ET) sm[r * (GSZ + XSZ) + (e / ET) * GP + (e % ET)] = v[u];
      }
    }
    for (int i0 = threadIdx.x; i0 < nr * XSZ; i0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int r = i / XSZ, e = i - r * XSZ;
        const int cl = e / WP, wv = e - cl * WP;
        const int b = min((r0 + r) / tpc, B - 1), to0 = ((r0 + r) % tpc) * ET;
        const int ci = ci0 + cl, ti = to0 * S - pad + wv;
        v[u] = (i < nr * XSZ && wv < WIN && ci < Cin && ti >= 0 && ti < Tin) ? x[((int64_t)b * Cin + ci) * Tin + ti] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u * 256;
        const int r = i / XSZ, e = i - r * XSZ;
        if (i < nr * XSZ) sm[r * (GSZ + XSZ) + GSZ + e] = v[u];
      }
    }
    __syncthreads();
    for (int r = 0; r < nr; ++r) {
      const float* gs = sm + r * (GSZ + XSZ);
      const float* xs = gs + GSZ;
      float g[ET];   // this thread's output channel, the tile's ET steps
#pragma unroll
      for (int q = 0; q < ET / 4; ++q) {
        const f32x4 v = *(const f32x4*)(gs + col * GP + 4 * q);
        g[4 * q] = v.x; g[4 * q + 1] = v.y; g[4 * q + 2] = v.z; g[4 * q + 3] = v.w;
        accb += (v.x + v.y) + (v.z + v.w);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const f32x4* xr4 = (const f32x4*)(xs + (cg * 4 + c) * WP);
        float xw[WP];
#pragma unroll
        for (int q = 0; q < WP / 4; ++q) {
          const f32x4 v = xr4[q];
          xw[4 * q] = v.x; xw[4 * q + 1] = v.y; xw[4 * q + 2] = v.z; xw[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
          for (int t = 0; t < ET; ++t) acc[c][j] = fmaf(g[t], xw[t * S + j], acc[c][j]);
      }
    }
  }
  const int co = co0 + col;
  if (co < Cout) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ci = ci0 + cg * 4 + c;
      if (ci < Cin)
#pragma unroll
        for (int j = 0; j < K; ++j) dw[((int64_t)co * Cin + ci) * K + j] += acc[c][j];
    }
    if (dbias && cg == 0 && by == 0) dbias[co] += accb;
  }
}
// ET: the time tile of both kernels (32, or 16 / 8 for the short sequences behind the encoder's strided blocks: at the reference's
// 8 x 5120-sample shard eight of the ten blocks see 8 frames, and a 32-step tile there is 75 % padding)
template <int K, int S, int PAD, int ET>
__global__ void __launch_bounds__(256) conv_bwd_x_tiled_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ y, const float* __restrict__ dy,
                                                               float* __restrict__ dx, int Cin, int Tin, int Cout, int Tout, int relu,
                                                               int residual) {
  conv_bwd_x_tiled_body<K, S, PAD, ET>(x, w, y, dy, dx, Cin, Tin, Cout, Tout, relu, residual, blockIdx.x, blockIdx.y, blockIdx.z);
}
template <int K, int S, int ET>
__global__ void __launch_bounds__(256) conv_bwd_w_tiled_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                               const float* __restrict__ dy, float* __restrict__ dw,
                                                               float* __restrict__ dbias, int B, int Cin, int Tin, int Cout, int Tout,
                                                               int pad, int relu, int residual) {
  conv_bwd_w_tiled_body<K, S, ET>(x, y, dy, dw, dbias, B, Cin, Tin, Cout, Tout, pad, relu, residual, blockIdx.x, blockIdx.y);
}
// Input and weight gradient of one block in ONE launch: the two read the same dy / x / w and depend on nothing of each other, but as two
// launches on one stream they ran one after the other -- each on 64 of the 256 CUs at the reference preset's shard (8 clips x 8-32
// frames).  The first nxx * nxy * nxz workgroups of a flat grid run the input-gradient body, the rest the weight-gradient body (the
// same arithmetic in the same order as the two kernels above).
template <int K, int S, int PAD, int ET>
__global__ void __launch_bounds__(256) conv_bwd_xw_tiled_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ y, const float* __restrict__ dy,
                                                                float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ dbias,
                                                                int B, int Cin, int Tin, int Cout, int Tout, int relu, int residual,
                                                                int nxx, int nxy, int nxz, int nwx) {
  const int id = blockIdx.x, nx = nxx * nxy * nxz;
  if (id < nx) {
    conv_bwd_x_tiled_body<K, S, PAD, ET>(x, w, y, dy, dx, Cin, Tin, Cout, Tout, relu, residual, id % nxx, (id / nxx) % nxy, id / (nxx * nxy));
  } else {
    const int iw = id - nx;
    conv_bwd_w_tiled_body<K, S, ET>(x, y, dy, dw, dbias, B, Cin, Tin, Cout, Tout, PAD, relu, residual, iw % nwx, iw / nwx);
  }
}

template <int K, int S, int PAD, int ET>
static int launch_conv_bwd_tiled_et(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* dbias,
                                    int B, int Cin, int Tin, int Cout, int Tout, int relu, int residual, hipStream_t st) {
  if (dx) {
    constexpr int R0 = (((PAD - K + 1) % S) + S) % S;
    constexpr int TWIN = (ET - 1 + K - 1 + R0) / S + 1;
    constexpr int TP = (TWIN + 3) & ~3;
    const size_t a = (size_t)(Cout < ECB_CH ? Cout : ECB_CH) * TP, r = (size_t)ECB_NS * ET * 33;
    const size_t lds_x = (a > r ? a : r) * sizeof(float);
    constexpr int WIN = (ET - 1) * S + K;
    constexpr int WP = (WIN + 3) & ~3;
    const size_t lds_w = (size_t)ECB_RT * (ECB_T * (ET + 4) + ECB_T * WP) * sizeof(float);
    const size_t lds = lds_x > lds_w ? lds_x : lds_w;
    const int nxx = (Tin + ET - 1) / ET, nxy = (Cin + ECB_T - 1) / ECB_T, nwx = (Cout + ECB_T - 1) / ECB_T, nwy = (Cin + ECB_T - 1) / ECB_T;
    static WaeLdsCache cache;
    if (int rc = wae_ensure_lds((const void*)conv_bwd_xw_tiled_kernel<K, S, PAD, ET>, cache, lds, "enc_conv_bwd"); rc != WAE_OK) return rc;
    hipLaunchKernelGGL((conv_bwd_xw_tiled_kernel<K, S, PAD, ET>), dim3(nxx * nxy * B + nwx * nwy), dim3(256), lds, st, x, w, y, dy, dx, dw,
                       dbias, B, Cin, Tin, Cout, Tout, relu, residual, nxx, nxy, B, nwx);
    return WAE_OK;
  }
  {
    constexpr int WIN = (ET - 1) * S + K;
    constexpr int WP = (WIN + 3) & ~3;
    const size_t lds = (size_t)ECB_RT * (ECB_T * (ET + 4) + ECB_T * WP) * sizeof(float);
    static WaeLdsCache cache;
    if (int rc = wae_ensure_lds((const void*)conv_bwd_w_tiled_kernel<K, S, ET>, cache, lds, "enc_conv_bwd"); rc != WAE_OK) return rc;
    hipLaunchKernelGGL((conv_bwd_w_tiled_kernel<K, S, ET>), dim3((Cout + ECB_T - 1) / ECB_T, (Cin + ECB_T - 1) / ECB_T), dim3(256), lds, st, x,
                       y, dy, dw, dbias, B, Cin, Tin, Cout, Tout, PAD, relu, residual);
  }
  return WAE_OK;
}
template <int K, int S, int PAD>
static int launch_conv_bwd_tiled(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw, float* dbias,
                                 int B, int Cin, int Tin, int Cout, int Tout, int relu, int residual, hipStream_t st) {
  // one tile size for both kernels, chosen by the longer of the two sequences (Tin >= Tout)
  if (Tin <= 8) return launch_conv_bwd_tiled_et<K, S, PAD, 8>(x, w, y, dy, dx, dw, dbias, B, Cin, Tin, Cout, Tout, relu, residual, st);
  if (Tin <= 16) return launch_conv_bwd_tiled_et<K, S, PAD, 16>(x, w, y, dy, dx, dw, dbias, B, Cin, Tin, Cout, Tout, relu, residual, st);
  return launch_conv_bwd_tiled_et<K, S, PAD, 32>(x, w, y, dy, dx, dw, dbias, B, Cin, Tin, Cout, Tout, relu, residual, st);
}

extern "C" int wae_enc_conv_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                                float* dbias, int32_t B, int32_t Cin, int32_t Tin, int32_t Cout, int32_t k, int32_t stride,
                                int32_t pad, int32_t relu, int32_t residual, void* stream) {
  WAE_REQUIRE(x && w && dy && dw && B > 0 && Cin > 0 && Tin > 0 && Cout > 0 && k > 0 && stride > 0 && pad >= 0,
              "enc_conv_bwd: bad arguments");
  WAE_REQUIRE(!relu || y, "enc_conv_bwd: relu needs the forward output y");
  const int Tout = (Tin + 2 * pad - k) / stride + 1;
  hipStream_t st = as_stream(stream);
  int rc = WAE_OK;   // a failed LDS opt-in skips the launch: the caller must see that
#define WAE_ECB(K_, S_, P_) rc = launch_conv_bwd_tiled<K_, S_, P_>(x, w, y, dy, dx, dw, dbias, B, Cin, Tin, Cout, Tout, relu, residual, st)
  if (k == 1 && stride == 1 && pad == 0) WAE_ECB(1, 1, 0);
  else if (k == 3 && stride == 1 && pad == 1) WAE_ECB(3, 1, 1);
  else if (k == 3 && stride == 1 && pad == 0) WAE_ECB(3, 1, 0);
  else if (k == 5 && stride == 2 && pad == 2) WAE_ECB(5, 2, 2);
  else if (k == 5 && stride == 1 && pad == 2) WAE_ECB(5, 1, 2);
  else if (k == 5 && stride == 1 && pad == 0) WAE_ECB(5, 1, 0);
  else {   // any other shape: the plain one-thread-per-output kernels
    if (dx)
      hipLaunchKernelGGL(conv_bwd_x_kernel, dim3((Tin + 63) / 64, (Cin + 3) / 4, B), dim3(256), 0, st, x, w, y, dy, dx, Cin, Tin,
                         Cout, Tout, k, stride, pad, relu, residual);
    hipLaunchKernelGGL(conv_bwd_w_kernel, dim3(Cout * Cin * k), dim3(64), 0, st, x, y, dy, dw, dbias, B, Cin, Tin, Cout, Tout, k,
                       stride, pad, relu, residual);
  }
#undef WAE_ECB
  if (rc != WAE_OK) return rc;
  return wae_check_launch("enc_conv_bwd");
}

// ---- VQ backward: straight-through + both halves of vq_loss (vector_quantization.py:41-45) ---------------------------------
// dlat = dquant + c_x*(x - q) ;  demb[idx] += c_e*(q - x)  over channels [d0, d0+D) of (B, Dtot, Tq) tensors; demb (K, D) of
// the slice's own codebook, or NULL (EMA codebooks get no gradient, vector_quantization.py:217,:292).
__global__ void __launch_bounds__(256) vq_bwd_kernel(const float* __restrict__ lat, const float* __restrict__ quant,
                                                     const int64_t* __restrict__ idx, const float* __restrict__ dquant,
                                                     float* __restrict__ dlat, float* __restrict__ demb, int Dtot, int d0, int D,
                                                     int Tq, float c_x, float c_e, int64_t total) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int t = (int)(e % Tq);
  const int dch = (int)((e / Tq) % D);
  const int b = (int)(e / ((int64_t)Tq * D));
  const int64_t at = ((int64_t)b * Dtot + d0 + dch) * Tq + t;
  const float diff = lat[at] - quant[at];
  dlat[at] = (dquant ? dquant[at] : 0.f) + c_x * diff;
  if (demb) atomicAdd(demb + idx[(int64_t)b * Tq + t] * D + dch, -c_e * diff);
}
extern "C" int wae_vq_slice_bwd(const float* lat, const float* quant, const int64_t* idx, const float* dquant, float* dlat,
                                float* demb, int32_t B, int32_t Dtot, int32_t d0, int32_t D, int32_t Tq, float c_lat, float c_emb,
                                void* stream) {
  WAE_REQUIRE(lat && quant && idx && dlat && B > 0 && D > 0 && Tq > 0 && d0 >= 0 && d0 + D <= Dtot, "vq_slice_bwd: bad arguments");
  const int64_t total = (int64_t)B * D * Tq;
  hipLaunchKernelGGL(vq_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), lat, quant, idx, dquant,
                     dlat, demb, Dtot, d0, D, Tq, c_lat, c_emb, total);
  return wae_check_launch("vq_slice_bwd");
}
extern "C" int wae_vq_bwd(const float* lat, const float* quant, const int64_t* idx, const float* dquant, float* dlat, float* demb,
                          int32_t B, int32_t D, int32_t Tq, float beta, float loss_scale, void* stream) {
  WAE_REQUIRE(demb, "vq_bwd: bad arguments");
  const float n = (float)((int64_t)B * D * Tq);
  return wae_vq_slice_bwd(lat, quant, idx, dquant, dlat, demb, B, D, 0, D, Tq, loss_scale * 2.f * beta / n, loss_scale * 2.f / n,
                          stream);
}
