// Discretized mixture of logistics: loss (+ gradient wrt the network output) and sampler.
// Reference: wavenet_vocoder/mixture.py:26-106 (loss), :118-156 (sampler); wrapper vqwae_train.py:382-401, shift :766.
// One lane per (clip, time step); the mixture dimension (M = out_channels/3, 10 in every preset) is a register
// loop, so the (B,3M,T) logits are read once, coalesced along T, and nothing but the per-sample loss is written.
#include "wae_common.hpp"

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }  // F.softplus, threshold 20
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

#define DMOL_MAX_M 32

__global__ void __launch_bounds__(256) dmol_loss_kernel(const float* __restrict__ y_hat, const float* __restrict__ y,
                                                        float* __restrict__ nll, float* __restrict__ dy_hat, int M, int T,
                                                        float half_bin, float log_half_classes, float log_scale_min,
                                                        int shift) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  const float* yh = y_hat + (int64_t)b * 3 * M * T + t;
  float* dyh = dy_hat ? dy_hat + (int64_t)b * 3 * M * T + t : nullptr;
  const int ty = t + shift;
  if (ty >= T) {  // no target for the last `shift` positions
    nll[(int64_t)b * T + t] = 0.f;
    if (dyh)
      for (int i = 0; i < 3 * M; ++i) dyh[(int64_t)i * T] = 0.f;
    return;
  }
  const float yy = y[(int64_t)b * T + ty];
  float lp[DMOL_MAX_M], dmu[DMOL_MAX_M], dls[DMOL_MAX_M];
  // log_softmax over the mixture logits (mixture.py:101)
  float lmax = -INFINITY;
  for (int i = 0; i < M; ++i) lmax = fmaxf(lmax, yh[(int64_t)i * T]);
  float lsum = 0.f;
  for (int i = 0; i < M; ++i) lsum += expf(yh[(int64_t)i * T] - lmax);
  const float llse = lmax + logf(lsum);
  float m = -INFINITY;
  for (int i = 0; i < M; ++i) {
    const float logit = yh[(int64_t)i * T];
    const float mu = yh[(int64_t)(M + i) * T];
    const float raw = yh[(int64_t)(2 * M + i) * T];
    const float ls = fmaxf(raw, log_scale_min);                       // :53
    const float pass = raw >= log_scale_min ? 1.f : 0.f;               // clamp gradient
    const float inv = expf(-ls);
    const float cen = yy - mu;
    const float plus = inv * (cen + half_bin), mn = inv * (cen - half_bin), mid = inv * cen;
    const float sp = sigmoid_f(plus), sm = sigmoid_f(mn);
    const float delta = sp - sm;
    float tval, tmu, tls;
    if (yy < -0.999f) {                                                 // :99, log cdf of the first bin
      tval = plus - softplus_f(plus);
      tmu = -(1.f - sp) * inv;
      tls = -(1.f - sp) * plus;
    } else if (yy > 0.999f) {                                           // :97, last bin
      tval = -softplus_f(mn);
      tmu = sm * inv;
      tls = sm * mn;
    } else if (delta > 1e-5f) {                                         // :91-95
      tval = logf(fmaxf(delta, 1e-12f));
      const float dp = sp * (1.f - sp), dm = sm * (1.f - sm);
      tmu = -inv * (dp - dm) / delta;
      tls = (-dp * plus + dm * mn) / delta;
    } else {                                                            // :79,:95 centre-of-bin density
      const float smid = sigmoid_f(mid);
      tval = mid - ls - 2.f * softplus_f(mid) - log_half_classes;
      tmu = -(1.f - 2.f * smid) * inv;
      tls = -(1.f - 2.f * smid) * mid - 1.f;
    }
    lp[i] = tval + (logit - llse);
    dmu[i] = tmu;
    dls[i] = tls * pass;
    m = fmaxf(m, lp[i]);
  }
  float s = 0.f;
  for (int i = 0; i < M; ++i) s += expf(lp[i] - m);
  const float lse = m + logf(s);                                        // :17-23
  nll[(int64_t)b * T + t] = -lse;
  if (dyh) {
    for (int i = 0; i < M; ++i) {
      const float w = expf(lp[i] - lse);                                // posterior responsibility
      const float pi = expf(yh[(int64_t)i * T] - llse);
      dyh[(int64_t)i * T] = pi - w;
      dyh[(int64_t)(M + i) * T] = -w * dmu[i];
      dyh[(int64_t)(2 * M + i) * T] = -w * dls[i];
    }
  }
}

extern "C" int wae_dmol_loss_fwd(const float* y_hat, const float* y, float* nll, float* dy_hat, int32_t B, int32_t M,
                                 int32_t T, int32_t num_classes, float log_scale_min, int32_t shift, void* stream) {
  WAE_REQUIRE(y_hat && y && nll && B > 0 && T > 0, "dmol_loss: bad arguments");
  WAE_REQUIRE(M > 0 && M <= DMOL_MAX_M, "dmol_loss: mixtures must be in 1..%d (got %d)", DMOL_MAX_M, M);
  WAE_REQUIRE(num_classes > 1 && shift >= 0, "dmol_loss: bad num_classes/shift");
  hipLaunchKernelGGL(dmol_loss_kernel, dim3((T + 255) / 256, B), dim3(256), 0, as_stream(stream), y_hat, y, nll, dy_hat, M, T,
                     1.0f / (float)(num_classes - 1), logf((float)(num_classes - 1) * 0.5f), log_scale_min, shift);
  return wae_check_launch("dmol_loss_fwd");
}

// sampler with caller-supplied uniforms in (1e-5, 1-1e-5): Gumbel-max mixture pick (mixture.py:138-140), logistic
// draw (:151-152), clamp to [-1,1] (:154).  y (B,3M,Tn), u_mix (B,Tn,M), u_log (B,Tn) -> out (B,Tn).
__global__ void __launch_bounds__(256) dmol_sample_kernel(const float* __restrict__ y, const float* __restrict__ u_mix,
                                                          const float* __restrict__ u_log, float* __restrict__ out, int M,
                                                          int Tn, float log_scale_min, int clamp_log_scale) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= Tn) return;
  const float* yh = y + (int64_t)b * 3 * M * Tn + t;
  const float* um = u_mix + ((int64_t)b * Tn + t) * M;
  float best = -INFINITY;
  int arg = 0;
  for (int i = 0; i < M; ++i) {
    const float v = yh[(int64_t)i * Tn] - logf(-logf(um[i]));
    if (v > best) { best = v; arg = i; }
  }
  const float mu = yh[(int64_t)(M + arg) * Tn];
  float ls = yh[(int64_t)(2 * M + arg) * Tn];
  if (clamp_log_scale) ls = fmaxf(ls, log_scale_min);
  const float u = u_log[(int64_t)b * Tn + t];
  const float x = mu + expf(ls) * (logf(u) - logf(1.f - u));
  out[(int64_t)b * Tn + t] = fminf(fmaxf(x, -1.f), 1.f);
}

extern "C" int wae_dmol_sample(const float* y, const float* u_mix, const float* u_log, float* out, int32_t B, int32_t M,
                               int32_t Tn, float log_scale_min, int32_t clamp_log_scale, void* stream) {
  WAE_REQUIRE(y && u_mix && u_log && out && B > 0 && M > 0 && Tn > 0, "dmol_sample: bad arguments");
  hipLaunchKernelGGL(dmol_sample_kernel, dim3((Tn + 255) / 256, B), dim3(256), 0, as_stream(stream), y, u_mix, u_log, out, M,
                     Tn, log_scale_min, clamp_log_scale);
  return wae_check_launch("dmol_sample");
}

// ---------------------------------------------------------------------------------------------------
// K15: clip_grad_norm_ + Adam + EMA over the flat parameter arena (vqwae_train.py:776-787, :339-350)
//   pass 1 (grad_sqnorm): partial sums of g^2 per workgroup -> one fp64 atomic per workgroup
//   pass 2 (clip_adam_ema): coef = min(1, thresh/(norm+1e-6)); Adam(lr, betas, eps, weight_decay), bias-corrected;
//                           shadow -= (1-decay)*(shadow - p)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grad_sqnorm_kernel(const float* __restrict__ g, int64_t n, double* __restrict__ acc) {
  // four 16-byte loads in flight per thread (one at a time ran at 1.3 TB/s: 31 us for the 42 MB arena of C2)
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * 1024;
  int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  for (; i + 3 * stride + 3 < n; i += 4 * stride) {
    const f32x4 a = *(const f32x4*)(g + i), b = *(const f32x4*)(g + i + stride), c = *(const f32x4*)(g + i + 2 * stride),
                d = *(const f32x4*)(g + i + 3 * stride);
    s += (double)a.x * a.x + (double)a.y * a.y + (double)a.z * a.z + (double)a.w * a.w;
    s += (double)b.x * b.x + (double)b.y * b.y + (double)b.z * b.z + (double)b.w * b.w;
    s += (double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z + (double)c.w * c.w;
    s += (double)d.x * d.x + (double)d.y * d.y + (double)d.z * d.z + (double)d.w * d.w;
  }
  for (; i < n; i += stride) {
    if (i + 3 < n) {
      const f32x4 v = *(const f32x4*)(g + i);
      s += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    } else {
      for (int64_t j = i; j < n; ++j) s += (double)g[j] * g[j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

__global__ void __launch_bounds__(256) clip_adam_ema_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                            float* __restrict__ m, float* __restrict__ v,
                                                            float* __restrict__ shadow, int64_t n,
                                                            const double* __restrict__ sqnorm, float* __restrict__ norm_out,
                                                            float step_size, float b1, float omb1, float b2, float omb2,
                                                            float eps, float wd, float bc2_sqrt, float clip,
                                                            float ema_keep) {
  const float norm = (float)sqrt(*sqnorm);
  float coef = 1.f;
  if (clip > 0.f) coef = fminf(1.f, clip / (norm + 1e-6f));
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    float gi = g[i] * coef;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + omb1 * gi;
    const float vi = b2 * v[i] + omb2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    p[i] = pi;
    if (shadow) {
      const float sh = shadow[i];
      shadow[i] = sh - ema_keep * (sh - pi);
    }
  }
}

extern "C" int wae_clip_adam_ema(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* shadow,
                                 int64_t n, double* scratch, float* grad_norm_out, int32_t step, double lr, double beta1,
                                 double beta2, double eps, double weight_decay, double clip_thresh, double ema_decay,
                                 void* stream) {
  WAE_REQUIRE(params && grads && exp_avg && exp_avg_sq && scratch && n > 0 && step >= 1, "clip_adam_ema: bad arguments");
  hipStream_t st = as_stream(stream);
  if (hipMemsetAsync(scratch, 0, sizeof(double), st) != hipSuccess) {
    wae_set_error("clip_adam_ema: memset failed");
    return WAE_EHIP;
  }
  const int grid = (int)((n + 1023) / 1024 > 1024 ? 1024 : (n + 1023) / 1024);
  hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(grid), dim3(256), 0, st, grads, n, scratch);
  // scalars are formed in double on the host exactly as torch.optim.Adam forms them, then rounded once to fp32
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(clip_adam_ema_kernel, dim3(2048), dim3(256), 0, st, params, grads, exp_avg, exp_avg_sq, shadow, n, scratch,
                     grad_norm_out, (float)(lr / bc1), (float)beta1, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                     (float)eps, (float)weight_decay, (float)sqrt(bc2), (float)clip_thresh, (float)(1.0 - ema_decay));
  return wae_check_launch("clip_adam_ema");
}

// ---------------------------------------------------------------------------------------------------
// Cross-entropy on explicit logits (the criterion object of vqwae_train.py:363-379 called on (B, C, T, 1) logits -- the
// drop-in MaskedCrossEntropyLoss; the training path proper uses the CE fused into the head kernel) and the masked mean with an
// arbitrary (B, T) weight mask.  logits (B, C, T) fp32: thread = one (b, t), classes strided by T (coalesced along t).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ce_logits_fwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                            float* __restrict__ nll, float* __restrict__ lse, int C, int T,
                                                            int32_t* __restrict__ err) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float* l = logits + (int64_t)b * C * T + t;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, l[(int64_t)c * T]);
  float den = 0.f;
  for (int c = 0; c < C; ++c) den += expf(l[(int64_t)c * T] - mx);
  const float ls = mx + logf(den);
  int64_t y = target[(int64_t)b * T + t];
  if (y < 0 || y >= C) {   // nn.CrossEntropyLoss raises: clamp and flag (WAE_ERR_TARGET_ID)
    if (err) atomicOr(err, 4);
    y = y < 0 ? 0 : C - 1;
  }
  nll[(int64_t)b * T + t] = ls - l[y * T];
  lse[(int64_t)b * T + t] = ls;
}
// dlogits[b,c,t] = (softmax - onehot(target)) * w[b,t]
__global__ void __launch_bounds__(256) ce_logits_bwd_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                            const float* __restrict__ lse, const float* __restrict__ w,
                                                            float* __restrict__ dlogits, int C, int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int64_t o = (int64_t)b * C * T + t;
  const float ls = lse[(int64_t)b * T + t], wt = w[(int64_t)b * T + t];
  int64_t y = target[(int64_t)b * T + t];
  y = y < 0 ? 0 : (y >= C ? C - 1 : y);
  for (int c = 0; c < C; ++c) dlogits[o + (int64_t)c * T] = (expf(logits[o + (int64_t)c * T] - ls) - (c == y ? 1.f : 0.f)) * wt;
}
extern "C" int wae_ce_logits_fwd(const float* logits, const int64_t* target, float* nll, float* lse, int32_t B, int32_t C, int32_t T,
                                 int32_t* err, void* stream) {
  WAE_REQUIRE(logits && target && nll && lse && B > 0 && C > 0 && T > 0, "ce_logits_fwd: bad arguments");
  hipLaunchKernelGGL(ce_logits_fwd_kernel, dim3((T + 255) / 256, B), dim3(256), 0, as_stream(stream), logits, target, nll, lse, C, T, err);
  return wae_check_launch("ce_logits_fwd");
}
extern "C" int wae_ce_logits_bwd(const float* logits, const int64_t* target, const float* lse, const float* w, float* dlogits,
                                 int32_t B, int32_t C, int32_t T, void* stream) {
  WAE_REQUIRE(logits && target && lse && w && dlogits && B > 0 && C > 0 && T > 0, "ce_logits_bwd: bad arguments");
  hipLaunchKernelGGL(ce_logits_bwd_kernel, dim3((T + 255) / 256, B), dim3(256), 0, as_stream(stream), logits, target, lse, w, dlogits, C, T);
  return wae_check_launch("ce_logits_bwd");
}

// out[0] = sum(v * m) / sum(m), out[1] = sum(m) over n elements (vqwae_train.py:379 with any mask); double accumulation
__global__ void __launch_bounds__(1024) weighted_mean_kernel(const float* __restrict__ v, const float* __restrict__ m, int64_t n,
                                                             float* __restrict__ out) {
  __shared__ double ps[16], pm[16];
  double s = 0.0, c = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float mi = m[i];
    s += (double)v[i] * (double)mi;
    c += (double)mi;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
  if ((threadIdx.x & 63) == 0) { ps[threadIdx.x >> 6] = s; pm[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tc = 0.0;
    for (int i = 0; i < 16; ++i) { ts += ps[i]; tc += pm[i]; }
    out[0] = (float)(ts / tc);
    out[1] = (float)tc;
  }
}
extern "C" int wae_weighted_mean(const float* v, const float* m, int64_t n, float* out, void* stream) {
  WAE_REQUIRE(v && m && out && n > 0, "weighted_mean: bad arguments");
  hipLaunchKernelGGL(weighted_mean_kernel, dim3(1), dim3(1024), 0, as_stream(stream), v, m, n, out);
  return wae_check_launch("weighted_mean");
}

// ---------------------------------------------------------------------------------------------------
// softmax over the channel dimension of (B, C, T) fp32 logits -- WaveNet.forward(softmax=True), wavenet.py:214 / vqvae_model.py:79-80
// (F.softmax(x, dim=1)) -- and its backward dx = p * (dp - sum_c p dp).  thread = one (b, t); channels are T apart: coalesced along t.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) softmax_bct_fwd_kernel(const float* __restrict__ x, float* __restrict__ p, int C, int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const float* xr = x + (int64_t)b * C * T + t;
  float* pr = p + (int64_t)b * C * T + t;
  float mx = -INFINITY;
  for (int c = 0; c < C; ++c) mx = fmaxf(mx, xr[(int64_t)c * T]);
  float den = 0.f;
  for (int c = 0; c < C; ++c) den += expf(xr[(int64_t)c * T] - mx);
  const float inv = 1.0f / den;
  for (int c = 0; c < C; ++c) pr[(int64_t)c * T] = expf(xr[(int64_t)c * T] - mx) * inv;
}
__global__ void __launch_bounds__(256) softmax_bct_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                              float* __restrict__ dx, int C, int T) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  const int64_t o = (int64_t)b * C * T + t;
  float dot = 0.f;
  for (int c = 0; c < C; ++c) dot = fmaf(p[o + (int64_t)c * T], dp[o + (int64_t)c * T], dot);
  for (int c = 0; c < C; ++c) dx[o + (int64_t)c * T] = p[o + (int64_t)c * T] * (dp[o + (int64_t)c * T] - dot);
}
extern "C" int wae_softmax_bct_fwd(const float* x, float* p, int32_t B, int32_t C, int32_t T, void* stream) {
  WAE_REQUIRE(x && p && B > 0 && C > 0 && T > 0, "softmax_bct_fwd: bad arguments");
  hipLaunchKernelGGL(softmax_bct_fwd_kernel, dim3((T + 255) / 256, B), dim3(256), 0, as_stream(stream), x, p, C, T);
  return wae_check_launch("softmax_bct_fwd");
}
extern "C" int wae_softmax_bct_bwd(const float* p, const float* dp, float* dx, int32_t B, int32_t C, int32_t T, void* stream) {
  WAE_REQUIRE(p && dp && dx && B > 0 && C > 0 && T > 0, "softmax_bct_bwd: bad arguments");
  hipLaunchKernelGGL(softmax_bct_bwd_kernel, dim3((T + 255) / 256, B), dim3(256), 0, as_stream(stream), p, dp, dx, C, T);
  return wae_check_launch("softmax_bct_bwd");
}

// ---------------------------------------------------------------------------------------------------
// C[n] = alpha * A[n] B[n], fp32 row-major, small batched products formed once per weight update (the per-layer matrices
// sqrt(.5) W1_cur W_out of wae_ar_generate_coop_fused).  16 x 16 output tile per workgroup, K in steps of 16 through LDS.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bmm_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c,
                                                      int M, int K, int N, int64_t lda, int64_t ldb, int64_t ldc, int64_t sa, int64_t sb,
                                                      int64_t sc, float alpha) {
  __shared__ float ta[16][17], tb[16][17];
  const int n = blockIdx.z;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int row = blockIdx.y * 16 + ty, col = blockIdx.x * 16 + tx;
  const float* an = a + n * sa;
  const float* bn = b + n * sb;
  float acc = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    ta[ty][tx] = (row < M && k0 + tx < K) ? an[(int64_t)row * lda + k0 + tx] : 0.f;
    tb[ty][tx] = (k0 + ty < K && col < N) ? bn[(int64_t)(k0 + ty) * ldb + col] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(ta[ty][k], tb[k][tx], acc);
    __syncthreads();
  }
  if (row < M && col < N) c[n * sc + (int64_t)row * ldc + col] = alpha * acc;
}
extern "C" int wae_bmm_f32(const float* a, const float* b, float* c, int32_t nbatch, int32_t M, int32_t K, int32_t N, int64_t lda,
                           int64_t ldb, int64_t ldc, int64_t stride_a, int64_t stride_b, int64_t stride_c, float alpha, void* stream) {
  WAE_REQUIRE(a && b && c && nbatch > 0 && M > 0 && K > 0 && N > 0 && lda >= K && ldb >= N && ldc >= N, "bmm_f32: bad arguments");
  hipLaunchKernelGGL(bmm_f32_kernel, dim3((N + 15) / 16, (M + 15) / 16, nbatch), dim3(256), 0, as_stream(stream), a, b, c, M, K, N, lda, ldb,
                     ldc, stride_a, stride_b, stride_c, alpha);
  return wae_check_launch("bmm_f32");
}
