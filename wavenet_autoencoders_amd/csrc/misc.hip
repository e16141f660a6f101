// Small HBM-bound kernels of the hot path: weight norm, fragment packing, encoder conv, VQ, upsample stages,
// hoisted global conditioning, first-conv gather, layout converters, masked mean.  All fp32 arithmetic.
#include <stdarg.h>
#include "wae_common.hpp"

static thread_local char g_err[512] = "";

void wae_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int wae_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    wae_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return WAE_EHIP;
  }
  return WAE_OK;
}
extern "C" const char* wae_version(void) { return "wae-hip 0.1 (gfx950)"; }
extern "C" const char* wae_last_error(void) { return g_err; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <typename E>
__device__ __forceinline__ void store_e(void* p, int64_t i, float v);
template <>
__device__ __forceinline__ void store_e<float>(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
template <>
__device__ __forceinline__ void store_e<__bf16>(void* p, int64_t i, float v) { ((__bf16*)p)[i] = (__bf16)v; }
template <>
__device__ __forceinline__ void store_e<f16>(void* p, int64_t i, float v) { ((f16*)p)[i] = (f16)v; }
template <typename E>
__device__ __forceinline__ float load_e(const void* p, int64_t i);
template <>
__device__ __forceinline__ float load_e<float>(const void* p, int64_t i) { return ((const float*)p)[i]; }
template <>
__device__ __forceinline__ float load_e<__bf16>(const void* p, int64_t i) { return (float)((const __bf16*)p)[i]; }
template <>
__device__ __forceinline__ float load_e<f16>(const void* p, int64_t i) { return (float)((const f16*)p)[i]; }

// ---------------------------------------------------------------------------------------------------
// weight norm: one wave per weight row (modules.py:18: w = g * v / ||v||, norm over all dims but 0).
// Rows are sorted by v_off and do not overlap.  Two launches, the arena touched once (the first form copied the whole arena with
// hipMemcpyAsync and then rewrote ~99 % of it): weight_norm_gaps_kernel passes everything BETWEEN the rows through (biases, the g
// scalars, embeddings, un-normed tensors), the row kernel does the rows.
// ---------------------------------------------------------------------------------------------------
// What these launches cost is latency, not bytes: 40 000 rows of 64..768 floats, each a table lookup, a read and a write that depend
// on one another, against ~2.5 us per trip to HBM.  One row per wave keeps 256 B..3 KiB in flight per wave -- with the chip's 8192
// resident waves that is ~0.8 TB/s (72 us for the arena of C2).  So a row belongs to a 16-lane GROUP (four rows per wave, 16 per
// workgroup), every lane issues all its 16-byte loads (up to 16: rows of up to 1024 floats) before the first use, and the norm is a
// 4-step butterfly inside the group.
#define WN_KMAX 16
__device__ __forceinline__ bool wn_vec_ok(const float* a, const float* b, const float* c, int n) {
  return n >= 4 && n <= 64 * WN_KMAX && (n & 3) == 0 && ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c)) & 15) == 0;
}
// the whole workgroup copies [a, b) (the largest gap of C2 -- the speaker embedding and the upsampling
// network, 14 000 floats -- copied by ONE 16-lane group, a load and a store per trip, was a 160-us tail on a 30-us launch)
__device__ __forceinline__ void wn_block_copy(const float* __restrict__ src, float* __restrict__ dst, int64_t a, int64_t b) {
  for (int64_t i0 = a + threadIdx.x; i0 < b; i0 += 256 * 16) {   // a 4096-float piece is ONE trip: sixteen loads in flight per thread
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = i0 + 256 * u < b ? src[i0 + 256 * u] : 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u)
      if (i0 + 256 * u < b) dst[i0 + 256 * u] = x[u];
  }
}
__device__ __forceinline__ float wn_dot(const f32x4& a, const f32x4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float wn_group_sum(float x) {   // over the 16 lanes of a group
  x += __shfl_xor(x, 1);
  x += __shfl_xor(x, 2);
  x += __shfl_xor(x, 4);
  x += __shfl_xor(x, 8);
  return x;
}

// dst = src on everything between the rows (biases, the g scalars, embeddings, whole un-normed tensors: the encoder's convolutions and the
// codebook are plain parameters -- gaps of up to a few hundred thousand floats).  blockIdx.x: 256 row boundaries, one per thread
// (boundary -1 = the head of the range); blockIdx.y: which 4096-float pieces of this block's gaps it copies (piece c, c + gridDim.y, ..).
#define WN_GAP_PIECE 4096
#define WN_GAP_Y 64
__global__ void __launch_bounds__(256) weight_norm_gaps_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               const int64_t* __restrict__ v_off, const int32_t* __restrict__ cols,
                                                               int nrows, int64_t lo, int64_t hi) {
  __shared__ int64_t s_a[256], s_b[256];
  __shared__ int s_n;
  const int row = blockIdx.x * 256 + threadIdx.x - 1;
  int64_t ga = 0, gb = 0;
  if (row < nrows) {
    if (nrows == 0) { ga = lo; gb = hi; }
    else {
      const int rc = max(row, 0), rn = min(row + 1, nrows - 1);
      ga = row < 0 ? lo : v_off[rc] + cols[rc];
      gb = row + 1 < nrows ? v_off[rn] : hi;
    }
  }
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  if (gb > ga) {
    const int k = atomicAdd(&s_n, 1);
    s_a[k] = ga;
    s_b[k] = gb;
  }
  __syncthreads();
  for (int k = 0; k < s_n; ++k) {
    const int64_t a = s_a[k], b = s_b[k];
    for (int64_t p0 = a + (int64_t)blockIdx.y * WN_GAP_PIECE; p0 < b; p0 += (int64_t)gridDim.y * WN_GAP_PIECE)
      wn_block_copy(src, dst, p0, min(b, p0 + WN_GAP_PIECE));
  }
}
static void wn_launch_gaps(const float* src, float* dst, const int64_t* v_off, const int32_t* cols, int nrows, int64_t lo, int64_t hi,
                           hipStream_t st) {
  hipLaunchKernelGGL(weight_norm_gaps_kernel, dim3((nrows + 1 + 255) / 256, WN_GAP_Y), dim3(256), 0, st, src, dst, v_off, cols, nrows, lo, hi);
}

__global__ void __launch_bounds__(256) weight_norm_fwd_kernel(const float* __restrict__ params, float* __restrict__ eff,
                                                              const int64_t* __restrict__ v_off,
                                                              const int64_t* __restrict__ g_off,
                                                              const int32_t* __restrict__ cols, int nrows) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  if (row >= nrows) return;
  const int64_t vo = v_off[row], go = g_off[row];
  const int n = cols[row];
  const float* v = params + vo;
  float* w = eff + vo;
  const float gsc = params[go];
  if (wn_vec_ok(v, w, v, n)) {
    f32x4 r[WN_KMAX];
#pragma unroll
    for (int k = 0; k < WN_KMAX; ++k)
      if (k * 64 < n) r[k] = *(const f32x4*)(v + min((l + 16 * k) * 4, n - 4));
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < WN_KMAX; ++k)
      if (k * 64 < n) ss += (l + 16 * k) * 4 < n ? wn_dot(r[k], r[k]) : 0.f;
    ss = wn_group_sum(ss);
    const float sc = gsc / sqrtf(ss);
#pragma unroll
    for (int k = 0; k < WN_KMAX; ++k)
      if ((l + 16 * k) * 4 < n) *(f32x4*)(w + (l + 16 * k) * 4) = r[k] * sc;
    return;
  }
  float ss = 0.f;
  for (int i = l; i < n; i += 16) ss += v[i] * v[i];
  ss = wn_group_sum(ss);
  const float sc = gsc / sqrtf(ss);
  for (int i = l; i < n; i += 16) w[i] = v[i] * sc;
}

extern "C" int wae_weight_norm_fwd(const float* params, float* eff, int64_t n_params, const int64_t* v_off,
                                   const int64_t* g_off, const int32_t* cols, int32_t nrows, void* stream) {
  WAE_REQUIRE(params && eff && n_params > 0 && nrows >= 0, "weight_norm_fwd: null arena");
  WAE_REQUIRE(nrows == 0 || (v_off && g_off && cols), "weight_norm_fwd: null tables");
  hipStream_t st = as_stream(stream);
  wn_launch_gaps(params, eff, v_off, cols, nrows, 0, n_params, st);
  if (nrows > 0)
    hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3((nrows + 15) / 16), dim3(256), 0, st, params, eff, v_off, g_off, cols, nrows);
  return wae_check_launch("weight_norm_fwd");
}

// The gap launch also passes over the g slots; the row launch -- later on the same stream -- overwrites them with the gradient of g.
__global__ void __launch_bounds__(256) weight_norm_bwd_kernel(const float* __restrict__ params,
                                                              const float* __restrict__ d_eff, float* __restrict__ grads,
                                                              const int64_t* __restrict__ v_off,
                                                              const int64_t* __restrict__ g_off,
                                                              const int32_t* __restrict__ cols, int nrows) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int l = threadIdx.x & 15;
  if (row >= nrows) return;
  const int64_t vo = v_off[row], go = g_off[row];
  const int n = cols[row];
  const float* v = params + vo;
  const float* dw = d_eff + vo;
  float* gv = grads + vo;
  const float g = params[go];
  if (wn_vec_ok(v, dw, gv, n)) {
    f32x4 rv[WN_KMAX], rd[WN_KMAX];
#pragma unroll
    for (int k = 0; k < WN_KMAX; ++k)
      if (k * 64 < n) {
        const int i = min((l + 16 * k) * 4, n - 4);
        rv[k] = *(const f32x4*)(v + i);
        rd[k] = *(const f32x4*)(dw + i);
      }
    float ss = 0.f, dv = 0.f;
#pragma unroll
    for (int k = 0; k < WN_KMAX; ++k)
      if (k * 64 < n) {
        const bool on = (l + 16 * k) * 4 < n;
        ss += on ? wn_dot(rv[k], rv[k]) : 0.f;
        dv += on ? wn_dot(rv[k], rd[k]) : 0.f;
      }
    ss = wn_group_sum(ss);
    dv = wn_group_sum(dv);
    const float inv = 1.0f / sqrtf(ss);
    const float gi = g * inv, c2 = dv * inv * inv;
#pragma unroll
    for (int k = 0; k < WN_KMAX; ++k)
      if ((l + 16 * k) * 4 < n) *(f32x4*)(gv + (l + 16 * k) * 4) = (rd[k] - rv[k] * c2) * gi;
    if (l == 0) grads[go] = dv * inv;
    return;
  }
  float ss = 0.f, dv = 0.f;
  for (int i = l; i < n; i += 16) {
    ss += v[i] * v[i];
    dv += v[i] * dw[i];
  }
  ss = wn_group_sum(ss);
  dv = wn_group_sum(dv);
  const float inv = 1.0f / sqrtf(ss);
  for (int i = l; i < n; i += 16) gv[i] = g * inv * (dw[i] - v[i] * dv * inv * inv);
  if (l == 0) grads[go] = dv * inv;
}

extern "C" int wae_weight_norm_bwd_range(const float* params, const float* d_eff, float* grads, int64_t lo, int64_t hi,
                                         const int64_t* v_off, const int64_t* g_off, const int32_t* cols, int32_t row_lo,
                                         int32_t row_hi, void* stream) {
  WAE_REQUIRE(params && d_eff && grads && lo >= 0 && hi > lo, "weight_norm_bwd: bad arena range");
  WAE_REQUIRE(row_lo >= 0 && row_hi >= row_lo, "weight_norm_bwd: bad row range");
  WAE_REQUIRE(row_hi == row_lo || (v_off && g_off && cols), "weight_norm_bwd: null tables");
  hipStream_t st = as_stream(stream);
  const int nrows = row_hi - row_lo;
  const int64_t* vo = nrows ? v_off + row_lo : v_off;
  const int64_t* go = nrows ? g_off + row_lo : g_off;
  const int32_t* co = nrows ? cols + row_lo : cols;
  wn_launch_gaps(d_eff, grads, vo, co, nrows, lo, hi, st);
  if (nrows > 0)
    hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3((nrows + 15) / 16), dim3(256), 0, st, params, d_eff, grads, vo, go, co, nrows);
  return wae_check_launch("weight_norm_bwd");
}

extern "C" int wae_weight_norm_bwd(const float* params, const float* d_eff, float* grads, int64_t n_params,
                                   const int64_t* v_off, const int64_t* g_off, const int32_t* cols, int32_t nrows,
                                   void* stream) {
  return wae_weight_norm_bwd_range(params, d_eff, grads, 0, n_params, v_off, g_off, cols, 0, nrows, stream);
}

// ---------------------------------------------------------------------------------------------------
// fragment packing: gather through a host-built index map
// ---------------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) pack_gather_kernel(const float* __restrict__ src, const int32_t* __restrict__ map,
                                                          void* __restrict__ dst, int64_t n, int64_t src_stride,
                                                          int64_t dst_stride) {
  const int b = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int32_t m = map[i];
    store_e<E>(dst, i + b * dst_stride, m < 0 ? 0.f : src[m + b * src_stride]);
  }
}

extern "C" int wae_pack_gather(const float* src, const int32_t* map, void* dst, int64_t n, int32_t nbatch,
                               int64_t src_stride, int64_t dst_stride, int32_t dtype, void* stream) {
  WAE_REQUIRE(src && map && dst && n > 0 && nbatch > 0, "pack_gather: bad arguments");
  WAE_REQUIRE(wae_dtype_ok(dtype), "pack_gather: bad dtype");
  const int gx = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (dtype == WAE_BF16)
    hipLaunchKernelGGL(pack_gather_kernel<__bf16>, dim3(gx, nbatch), dim3(256), 0, as_stream(stream), src, map, dst, n,
                       src_stride, dst_stride);
  else if (dtype == WAE_F16)
    hipLaunchKernelGGL(pack_gather_kernel<f16>, dim3(gx, nbatch), dim3(256), 0, as_stream(stream), src, map, dst, n,
                       src_stride, dst_stride);
  else
    hipLaunchKernelGGL(pack_gather_kernel<float>, dim3(gx, nbatch), dim3(256), 0, as_stream(stream), src, map, dst, n,
                       src_stride, dst_stride);
  return wae_check_launch("pack_gather");
}

__global__ void __launch_bounds__(256) unpack_scatter_add_kernel(const float* __restrict__ src,
                                                                 const int32_t* __restrict__ map, float* __restrict__ dst,
                                                                 int64_t n, int64_t src_stride, int64_t dst_stride, int src_cols,
                                                                 int64_t src_ld, int unique) {
  const int b = blockIdx.y;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int32_t m = map[i];
    const int64_t si = src_cols > 0 ? (i / src_cols) * src_ld + (i % src_cols) : i;
    if (m >= 0) {
      if (unique) dst[m + b * dst_stride] += src[si + b * src_stride];   // every slot has exactly one source in this launch
      else atomicAdd(dst + m + b * dst_stride, src[si + b * src_stride]);
    }
  }
}

// unique == 2: every row of the (rows x src_cols) block feeds ONE destination (map[row * src_cols]; the per-clip ones
// columns of a bias gradient): sum the row, add once.  Atomics from 128 columns x 24 layers into the same few thousand
// slots took 0.6 ms per step.
__global__ void __launch_bounds__(256) unpack_rowsum_add_kernel(const float* __restrict__ src, const int32_t* __restrict__ map,
                                                                float* __restrict__ dst, int rows, int src_cols, int64_t src_ld,
                                                                int64_t src_stride, int64_t dst_stride) {
  const int b = blockIdx.y;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* sr = src + b * src_stride + (int64_t)row * src_ld;
  float s = 0.f;
  for (int c = lane; c < src_cols; c += 64) s += sr[c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  const int32_t m = map[(int64_t)row * src_cols];
  if (lane == 0 && m >= 0) dst[m + b * dst_stride] += s;
}

extern "C" int wae_unpack_scatter_add(const float* src, const int32_t* map, float* dst, int64_t n, int32_t nbatch,
                                      int64_t src_stride, int64_t dst_stride, int32_t src_cols, int64_t src_ld, int32_t unique,
                                      void* stream) {
  WAE_REQUIRE(src && map && dst && n > 0 && nbatch > 0, "unpack_scatter_add: bad arguments");
  if (unique == 2) {
    WAE_REQUIRE(src_cols > 0 && n % src_cols == 0, "unpack_scatter_add: row-sum mode needs src_cols > 0 dividing n");
    const int rows = (int)(n / src_cols);
    hipLaunchKernelGGL(unpack_rowsum_add_kernel, dim3((rows + 3) / 4, nbatch), dim3(256), 0, as_stream(stream), src, map, dst,
                       rows, src_cols, src_ld, src_stride, dst_stride);
    return wae_check_launch("unpack_rowsum_add");
  }
  const int gx = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(unpack_scatter_add_kernel, dim3(gx, nbatch), dim3(256), 0, as_stream(stream), src, map, dst, n,
                     src_stride, dst_stride, src_cols, src_ld, unique);
  return wae_check_launch("unpack_scatter_add");
}

// ---------------------------------------------------------------------------------------------------
// The same two operations over a LIST of jobs in one launch.  A train step packs eleven weight families and scatters
// eleven gradient blocks; as separate launches of 5-10 us each (most of it dispatch) they cost ~0.2 ms of a 6.5 ms
// step.  The flat block index is mapped to (job, batch, block of the job) through a prefix table in the kernel arguments.
// ---------------------------------------------------------------------------------------------------
struct MultiGather {
  wae_gather_job j[WAE_MULTI_MAX];
  int first[WAE_MULTI_MAX + 1];   // first flat block of job k; first[njobs] = grid size
  int bx[WAE_MULTI_MAX];          // blocks per batch entry
  int njobs;
};
struct MultiScatter {
  wae_scatter_job j[WAE_MULTI_MAX];
  int first[WAE_MULTI_MAX + 1];
  int bx[WAE_MULTI_MAX];
  int njobs;
};
template <typename M>
__device__ __forceinline__ int multi_find(const M& p, int bid) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < WAE_MULTI_MAX; ++i)
    if (i < p.njobs && bid >= p.first[i]) k = i;
  return k;
}

// A thread owns two consecutive elements of a job's map and walks the job's batch entries (the layers: one map serves them all) with
// them, eight entries -- sixteen gathers -- in flight: the map is read once instead of once per layer, and a 16-bit pair is one 4-byte
// store.  (One element per thread and (job, layer) block: 50 us per pack of C2's 10 M weights, 10 bytes moved per element.)
#define MG_PAIR 2
#define MG_UNROLL 8
template <typename E>
__device__ __forceinline__ void gather_job_pairs(const wae_gather_job& jb, int64_t i) {
  const bool two = i + 1 < jb.n;
  const int32_t m0 = jb.map[i], m1 = two ? jb.map[i + 1] : -1;
  E* dst = (E*)jb.dst;
  const bool pair_store = two && sizeof(E) == 2 && ((jb.dst_stride & 1) == 0) && ((((uintptr_t)dst) & 3) == 0);
  for (int b0 = 0; b0 < jb.nbatch; b0 += MG_UNROLL) {
    float v0[MG_UNROLL], v1[MG_UNROLL];
#pragma unroll
    for (int u = 0; u < MG_UNROLL; ++u) {
      const float* src = jb.src + (int64_t)min(b0 + u, jb.nbatch - 1) * jb.src_stride;
      v0[u] = m0 < 0 ? 0.f : src[m0];
      v1[u] = m1 < 0 ? 0.f : src[m1];
    }
#pragma unroll
    for (int u = 0; u < MG_UNROLL; ++u) {
      if (b0 + u >= jb.nbatch) break;
      const int64_t o = i + (int64_t)(b0 + u) * jb.dst_stride;
      if (pair_store) {
        E pr[2] = {(E)v0[u], (E)v1[u]};
        *(uint32_t*)(dst + o) = *(const uint32_t*)pr;
      } else {
        dst[o] = (E)v0[u];
        if (two) dst[o + 1] = (E)v1[u];
      }
    }
  }
}
// a job without a map is a FILL: n fp32 zeros at dst (the gradient arenas of a step are cleared by the launch that packs the backward
// weights, beside its gathers -- round 5 cleared them with two torch fills of 43 + 50 MB between launches)
#define MG_FILL 8192        // floats per block of a fill job
__global__ void __launch_bounds__(256) gather_multi_kernel(MultiGather p) {
  const int k = multi_find(p, blockIdx.x);
  const wae_gather_job& jb = p.j[k];
  if (jb.map == nullptr) {
    float* dst = (float*)jb.dst;
    const int64_t base = (int64_t)(blockIdx.x - p.first[k]) * MG_FILL;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < MG_FILL / 1024; ++u) {
      const int64_t o = base + u * 1024 + threadIdx.x * 4;
      if (o + 3 < jb.n) *(f32x4*)(dst + o) = z;
      else for (int64_t q = o; q < jb.n; ++q) dst[q] = 0.f;
    }
    return;
  }
  const int64_t i = ((int64_t)(blockIdx.x - p.first[k]) * 256 + threadIdx.x) * MG_PAIR;
  if (i >= jb.n) return;
  if (jb.dtype == WAE_BF16) gather_job_pairs<__bf16>(jb, i);
  else if (jb.dtype == WAE_F16) gather_job_pairs<f16>(jb, i);
  else gather_job_pairs<float>(jb, i);
}

// (The same walk over the batch entries measured SLOWER for the scatter -- 51 us against 42 us per launch: its read-modify-write of
// 4-byte slots a weight row apart wants the slots of one layer touched close together in time -- so it keeps a block per (job, layer).)
__global__ void __launch_bounds__(256) scatter_multi_kernel(MultiScatter p) {
  const int k = multi_find(p, blockIdx.x);
  const wae_scatter_job jb = p.j[k];
  const int local = blockIdx.x - p.first[k], bx = p.bx[k];
  const int b = local / bx, x = local - b * bx;
  const float* src = jb.src + (int64_t)b * jb.src_stride;
  float* dst = jb.dst + (int64_t)b * jb.dst_stride;
  if (jb.unique == 2) {   // one wave per row of the tile: sum, add once (unpack_rowsum_add_kernel)
    const int rows = (int)(jb.n / jb.src_cols);
    const int row = x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* sr = src + (int64_t)row * jb.src_ld;
    float s = 0.f;
    for (int c = lane; c < jb.src_cols; c += 64) s += sr[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    const int32_t m = jb.map[(int64_t)row * jb.src_cols];
    if (lane == 0 && m >= 0) dst[m] += s;
    return;
  }
  for (int64_t i = (int64_t)x * 256 + threadIdx.x; i < jb.n; i += (int64_t)bx * 256) {
    const int32_t m = jb.map[i];
    const int64_t si = jb.src_cols > 0 ? (i / jb.src_cols) * jb.src_ld + (i % jb.src_cols) : i;
    if (m >= 0) {
      if (jb.unique) dst[m] += src[si];
      else atomicAdd(dst + m, src[si]);
    }
  }
}

extern "C" int wae_pack_gather_multi(const wae_gather_job* jobs_host, int32_t njobs, void* stream) {
  WAE_REQUIRE(jobs_host && njobs > 0 && njobs <= WAE_MULTI_MAX, "pack_gather_multi: 1..WAE_MULTI_MAX jobs");
  MultiGather a;
  int total = 0;
  for (int k = 0; k < njobs; ++k) {
    const wae_gather_job& j = jobs_host[k];
    WAE_REQUIRE(j.dst && j.n > 0 && (j.map ? (j.src && j.nbatch > 0 && wae_dtype_ok(j.dtype)) : ((((uintptr_t)j.dst) & 15) == 0)),
                "pack_gather_multi: bad job (a fill job -- map null -- needs a 16-byte aligned fp32 dst)");
    a.j[k] = j;
    a.bx[k] = j.map ? (int)((j.n + 256 * MG_PAIR - 1) / (256 * MG_PAIR))    // every batch entry of a block's elements is walked by that block
                    : (int)((j.n + MG_FILL - 1) / MG_FILL);
    a.first[k] = total;
    total += a.bx[k];
  }
  for (int k = njobs; k <= WAE_MULTI_MAX; ++k) a.first[k] = total;
  for (int k = njobs; k < WAE_MULTI_MAX; ++k) a.bx[k] = 1;
  a.njobs = njobs;
  hipLaunchKernelGGL(gather_multi_kernel, dim3(total), dim3(256), 0, as_stream(stream), a);
  return wae_check_launch("pack_gather_multi");
}

extern "C" int wae_unpack_scatter_add_multi(const wae_scatter_job* jobs_host, int32_t njobs, void* stream) {
  WAE_REQUIRE(jobs_host && njobs > 0 && njobs <= WAE_MULTI_MAX, "unpack_scatter_add_multi: 1..WAE_MULTI_MAX jobs");
  MultiScatter a;
  int total = 0;
  for (int k = 0; k < njobs; ++k) {
    const wae_scatter_job& j = jobs_host[k];
    WAE_REQUIRE(j.src && j.map && j.dst && j.n > 0 && j.nbatch > 0, "unpack_scatter_add_multi: bad job");
    WAE_REQUIRE(j.unique != 2 || (j.src_cols > 0 && j.n % j.src_cols == 0), "unpack_scatter_add_multi: row-sum mode needs src_cols > 0 dividing n");
    a.j[k] = j;
    int64_t nb = j.unique == 2 ? (j.n / j.src_cols + 3) / 4 : (j.n + 255) / 256;
    if (j.unique != 2 && nb > 2048) nb = 2048;
    a.bx[k] = (int)nb;
    a.first[k] = total;
    total += a.bx[k] * j.nbatch;
  }
  for (int k = njobs; k <= WAE_MULTI_MAX; ++k) a.first[k] = total;
  for (int k = njobs; k < WAE_MULTI_MAX; ++k) a.bx[k] = 1;
  a.njobs = njobs;
  hipLaunchKernelGGL(scatter_multi_kernel, dim3(total), dim3(256), 0, as_stream(stream), a);
  return wae_check_launch("unpack_scatter_add_multi");
}

// ---------------------------------------------------------------------------------------------------
// encoder block (vqvae_model.py:17-23).  (B,C,T) fp32; one wave per (b, cout, 64 output frames): lanes
// run along time so x reads coalesce; the (cin,k) reduction is sequential per lane.  The encoder runs at
// 1/160 .. 1/640 of the audio rate (<1 % of a step), so this stays a plain VALU kernel.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) enc_conv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int B,
                                                           int Cin, int Tin, int Cout, int Tout, int k, int stride, int pad,
                                                           int relu, int residual) {
  const int to = blockIdx.x * 64 + (threadIdx.x & 63);
  const int co = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.z;
  if (co >= Cout || to >= Tout) return;
  float acc = bias ? bias[co] : 0.f;
  const float* xb = x + (int64_t)b * Cin * Tin;
  const float* wr = w + (int64_t)co * Cin * k;
  for (int ci = 0; ci < Cin; ++ci) {
    const float* xr = xb + (int64_t)ci * Tin;
    for (int j = 0; j < k; ++j) {
      const int ti = to * stride + j - pad;
      if (ti >= 0 && ti < Tin) acc = fmaf(wr[ci * k + j], xr[ti], acc);
    }
  }
  if (relu) acc = fmaxf(acc, 0.f);
  if (residual) acc += xb[(int64_t)co * Tin + to];
  y[((int64_t)b * Cout + co) * Tout + to] = acc;
}

// Tiled form (round 2).  At the training shapes of hps/vqwae.json -- 8 clips x 32 frames, 256 channels -- the kernel above runs
// one thread per output with a serial 768-term reduction, half of its lanes idle (Tout = 32 of 64) and ~90 us per layer: twelve
// of them plus their backward were 3.5 ms of a 7.5 ms train step.  Here a block owns 32 output channels x 32 output steps of
// one clip: the input window of every input channel is staged in LDS (chunks of 256 channels), thread = (channel, one of 8
// slices of the reduction), 32 accumulators per thread, the window reads are broadcasts; the slices meet in LDS.
#define EC_T 32     // outputs per block along time and along channels
#define EC_NS 8     // reduction slices
#define EC_CH 256   // input channels staged per chunk
#define EC_WB 8     // channels whose weights a thread requests together
// ET: outputs per block along time (32, or 16 / 8 for the short sequences behind the encoder's strided blocks -- at the reference's
// 8 x 5120-sample shard the last eight blocks see 8 frames: a 32-frame tile there is 75 % padding and the launch is its dependent trips
// to memory, so the next batch of weights is requested while the current one is multiplied).
template <int K, int S, int ET>
__global__ void __launch_bounds__(256) enc_conv_fwd_tiled_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int Cin,
                                                                 int Tin, int Cout, int Tout, int pad, int relu, int residual) {
  extern __shared__ float sm[];
  constexpr int WIN = (ET - 1) * S + K;
  constexpr int WP = (WIN + 3) & ~3;   // LDS row pitch: whole 16-byte reads
  const int b = blockIdx.z, co0 = blockIdx.y * EC_T, to0 = blockIdx.x * ET;
  const int ti0 = to0 * S - pad;
  const int col = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int co = co0 + col;
  float acc[ET];
#pragma unroll
  for (int i = 0; i < ET; ++i) acc[i] = 0.f;
  const float* xb = x + (int64_t)b * Cin * Tin;
  for (int c0 = 0; c0 < Cin; c0 += EC_CH) {
    const int nc = min(EC_CH, Cin - c0);
    __syncthreads();
    // staged in batches of 16 independent loads per thread (a load -> LDS-store loop with a run-time trip count was one L2 /
    // HBM round trip per element: 32 dependent round trips, most of the launch)
    for (int i0 = threadIdx.x; i0 < nc * WP; i0 += 256 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int i = i0 + u * 256;
        const int ci = i / WP, wv = i - ci * WP, ti = ti0 + wv;
        v[u] = (i < nc * WP && wv < WIN && ti >= 0 && ti < Tin) ? xb[(int64_t)(c0 + ci) * Tin + ti] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (i0 + u * 256 < nc * WP) sm[i0 + u * 256] = v[u];
    }
    __syncthreads();
    const int per = (nc + EC_NS - 1) / EC_NS;
    const int ca = sl * per, cb = min(nc, ca + per);
    if (co < Cout) {
      // the weights of a thread (row co, channels [ca, cb): one contiguous run) come in batches of EC_WB channels requested
      // together, the NEXT batch while this one is used: one L2 round trip per thread and chunk is exposed instead of one per batch
      const float* wrow = w + ((int64_t)co * Cin + c0) * K;
      float wv[EC_WB][K], wn[EC_WB][K];
      auto fetch = [&](float (&dst)[EC_WB][K], int cc) {
#pragma unroll
        for (int u = 0; u < EC_WB; ++u)
#pragma unroll
          for (int j = 0; j < K; ++j) dst[u][j] = cc + u < cb ? wrow[(cc + u) * K + j] : 0.f;
      };
      if (ca < cb) fetch(wv, ca);
      for (int cc = ca; cc < cb; cc += EC_WB) {
        if (cc + EC_WB < cb) fetch(wn, cc + EC_WB);
#pragma unroll
        for (int u = 0; u < EC_WB; ++u) {
          // the channel's input window moves LDS -> registers once (16-byte broadcast reads) and serves every tap
          const f32x4* xr4 = (const f32x4*)(sm + min(cc + u, cb - 1) * WP);
          float xw[WP];
#pragma unroll
          for (int q = 0; q < WP / 4; ++q) {
            const f32x4 v = xr4[q];
            xw[4 * q] = v.x; xw[4 * q + 1] = v.y; xw[4 * q + 2] = v.z; xw[4 * q + 3] = v.w;
          }
#pragma unroll
          for (int j = 0; j < K; ++j)
#pragma unroll
            for (int t = 0; t < ET; ++t) acc[t] = fmaf(wv[u][j], xw[t * S + j], acc[t]);
        }
#pragma unroll
        for (int u = 0; u < EC_WB; ++u)
#pragma unroll
          for (int j = 0; j < K; ++j) wv[u][j] = wn[u][j];
      }
    }
  }
  __syncthreads();
  float* red = sm;   // [slice][t][33]
#pragma unroll
  for (int t = 0; t < ET; ++t) red[(sl * ET + t) * 33 + col] = acc[t];
  __syncthreads();
  for (int o = threadIdx.x; o < ET * EC_T; o += 256) {
    const int tl = o % ET, cl = o / ET;
    const int oc = co0 + cl, ot = to0 + tl;
    if (oc >= Cout || ot >= Tout) continue;
    float v = bias ? bias[oc] : 0.f;
#pragma unroll
    for (int q = 0; q < EC_NS; ++q) v += red[(q * ET + tl) * 33 + cl];
    if (relu) v = fmaxf(v, 0.f);
    if (residual) v += xb[(int64_t)oc * Tin + ot];
    y[((int64_t)b * Cout + oc) * Tout + ot] = v;
  }
}
template <int K, int S, int ET>
static int launch_enc_fwd_tiled_et(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Tin, int Cout,
                                   int Tout, int pad, int relu, int residual, hipStream_t st) {
  constexpr int WIN = (ET - 1) * S + K;
  constexpr int WP = (WIN + 3) & ~3;
  const size_t a = (size_t)(Cin < EC_CH ? Cin : EC_CH) * WP, r = (size_t)EC_NS * ET * 33;
  const size_t lds = (a > r ? a : r) * sizeof(float);
  static WaeLdsCache cache;
  if (int rc = wae_ensure_lds((const void*)enc_conv_fwd_tiled_kernel<K, S, ET>, cache, lds, "enc_conv_fwd"); rc != WAE_OK) return rc;   // y untouched: the caller must see the error
  hipLaunchKernelGGL((enc_conv_fwd_tiled_kernel<K, S, ET>), dim3((Tout + ET - 1) / ET, (Cout + EC_T - 1) / EC_T, B), dim3(256), lds, st,
                     x, w, bias, y, Cin, Tin, Cout, Tout, pad, relu, residual);
  return WAE_OK;
}
template <int K, int S>
static int launch_enc_fwd_tiled(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Tin, int Cout,
                                int Tout, int pad, int relu, int residual, hipStream_t st) {
  if (Tout <= 8) return launch_enc_fwd_tiled_et<K, S, 8>(x, w, bias, y, B, Cin, Tin, Cout, Tout, pad, relu, residual, st);
  if (Tout <= 16) return launch_enc_fwd_tiled_et<K, S, 16>(x, w, bias, y, B, Cin, Tin, Cout, Tout, pad, relu, residual, st);
  return launch_enc_fwd_tiled_et<K, S, 32>(x, w, bias, y, B, Cin, Tin, Cout, Tout, pad, relu, residual, st);
}

extern "C" int wae_enc_conv_fwd(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin,
                                int32_t Tin, int32_t Cout, int32_t k, int32_t stride, int32_t pad, int32_t relu,
                                int32_t residual, void* stream) {
  WAE_REQUIRE(x && w && y && B > 0 && Cin > 0 && Tin > 0 && Cout > 0 && k > 0 && stride > 0 && pad >= 0,
              "enc_conv: bad arguments");
  WAE_REQUIRE(!residual || (stride == 1 && Cin == Cout && 2 * pad == k - 1), "enc_conv: residual needs a same-shape conv");
  const int Tout = (Tin + 2 * pad - k) / stride + 1;
  WAE_REQUIRE(Tout > 0, "enc_conv: empty output");
  hipStream_t st = as_stream(stream);
  int rc = WAE_OK;
  if (k == 1 && stride == 1) rc = launch_enc_fwd_tiled<1, 1>(x, w, bias, y, B, Cin, Tin, Cout, Tout, pad, relu, residual, st);
  else if (k == 3 && stride == 1) rc = launch_enc_fwd_tiled<3, 1>(x, w, bias, y, B, Cin, Tin, Cout, Tout, pad, relu, residual, st);
  else if (k == 5 && stride == 2) rc = launch_enc_fwd_tiled<5, 2>(x, w, bias, y, B, Cin, Tin, Cout, Tout, pad, relu, residual, st);
  else if (k == 5 && stride == 1) rc = launch_enc_fwd_tiled<5, 1>(x, w, bias, y, B, Cin, Tin, Cout, Tout, pad, relu, residual, st);
  else   // any other shape (e.g. a wider conv_in, cin_pad > 2): the plain one-thread-per-output kernel
    hipLaunchKernelGGL(enc_conv_fwd_kernel, dim3((Tout + 63) / 64, (Cout + 3) / 4, B), dim3(256), 0, st, x, w, bias, y, B, Cin, Tin,
                       Cout, Tout, k, stride, pad, relu, residual);
  if (rc != WAE_OK) return rc;
  return wae_check_launch("enc_conv_fwd");
}

// ---------------------------------------------------------------------------------------------------
// VQ nearest (vector_quantization.py:21-49; the sliced/EMA classes of the same file score each slice the same way,
// :84-97, :163-176, :262-270).  One workgroup per latent vector; thread k scores code k with the reference's
// formulation ||e||^2 + ||x||^2 - 2 x.e, all fp32; first minimum wins (torch.argmin / argmax(-d) tie-break).
// The kernel works on channels [d0, d0+D) of a (B, Dtot, Tq) tensor.  mode 0: search + gather, 1: search only,
// 2: gather with the indices given (after an EMA update moved the codebook).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vq_nearest_kernel(const float* __restrict__ lat, const float* __restrict__ emb,
                                                         int64_t* __restrict__ idx, float* __restrict__ quant,
                                                         float* __restrict__ sqerr, int32_t* __restrict__ hist, int Dtot,
                                                         int d0, int D, int Tq, int K, int mode) {
  extern __shared__ float sh[];
  float* xs = sh;                      // D
  float* best_d = sh + D;              // 256
  int* best_i = (int*)(best_d + 256);  // 256
  const int row = blockIdx.x;          // b*Tq + t
  const int b = row / Tq, t = row % Tq;
  const int64_t base = ((int64_t)b * Dtot + d0) * Tq + t;
  for (int i = threadIdx.x; i < D; i += 256) xs[i] = lat[base + (int64_t)i * Tq];
  __syncthreads();
  int win;
  if (mode != 2) {
    float in_sqr = 0.f;
    for (int i = 0; i < D; ++i) in_sqr += xs[i] * xs[i];
    float bd = INFINITY;
    int bi = 0x7fffffff;
    for (int kk = threadIdx.x; kk < K; kk += 256) {
      const float* e = emb + (int64_t)kk * D;
      float es = 0.f, dot = 0.f;
      for (int i = 0; i < D; ++i) {
        es += e[i] * e[i];
        dot += xs[i] * e[i];
      }
      const float d = (es + in_sqr) - 2.0f * dot;
      if (d < bd) { bd = d; bi = kk; }
    }
    best_d[threadIdx.x] = bd;
    best_i[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        const float od = best_d[threadIdx.x + s];
        const int oi = best_i[threadIdx.x + s];
        if (od < best_d[threadIdx.x] || (od == best_d[threadIdx.x] && oi < best_i[threadIdx.x])) {
          best_d[threadIdx.x] = od;
          best_i[threadIdx.x] = oi;
        }
      }
      __syncthreads();
    }
    win = best_i[0];
    if (threadIdx.x == 0) {
      idx[row] = win;
      atomicAdd(hist + win, 1);
    }
  } else {
    win = (int)idx[row];
  }
  if (mode == 1) return;
  float se = 0.f;
  for (int i = threadIdx.x; i < D; i += 256) {
    const float q = emb[(int64_t)win * D + i];
    quant[base + (int64_t)i * Tq] = q;
    const float df = q - xs[i];
    se += df * df;
  }
  se = wave_sum(se);
  if ((threadIdx.x & 63) == 0 && se != 0.f) atomicAdd(sqerr, se);
}

// stats[0] = c_loss * mean((q-x)^2) over the slice (mode != 1); stats[1] = perplexity of the slice (mode != 2)
__global__ void __launch_bounds__(256) vq_stats_kernel(const float* __restrict__ sqerr, const int32_t* __restrict__ hist,
                                                       float* __restrict__ stats, int K, int N, int D, float c_loss, int mode) {
  __shared__ float part[4];
  float s = 0.f;
  if (mode != 2)
    for (int k = threadIdx.x; k < K; k += 256) {
      const float pk = (float)hist[k] / (float)N;
      s += pk * logf(pk + 1e-10f);
    }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float tot = part[0] + part[1] + part[2] + part[3];
    const float mse = sqerr[0] / ((float)N * (float)D);
    if (mode != 1) stats[0] = c_loss * mse;
    if (mode != 2) stats[1] = expf(-tot);
  }
}

extern "C" int wae_vq_slice(const float* lat, const float* emb, int64_t* idx, float* quant, float* stats, int32_t* hist,
                            int32_t B, int32_t Dtot, int32_t d0, int32_t D, int32_t Tq, int32_t K, float c_loss, int32_t mode,
                            void* stream) {
  WAE_REQUIRE(lat && emb && idx && stats && hist && (quant || mode == 1), "vq_slice: null pointer");
  WAE_REQUIRE(B > 0 && D > 0 && Tq > 0 && K > 0 && d0 >= 0 && d0 + D <= Dtot && mode >= 0 && mode <= 2, "vq_slice: bad sizes");
  hipStream_t st = as_stream(stream);
  // scratch: hist[K] counts, then one float (sum of squared errors) stored after it; mode 2 keeps the counts
  float* sqerr = (float*)(hist + K);
  const hipError_t e = mode == 2 ? hipMemsetAsync(sqerr, 0, sizeof(float), st)
                                 : hipMemsetAsync(hist, 0, (size_t)(K + 1) * sizeof(int32_t), st);
  if (e != hipSuccess) {
    wae_set_error("vq_slice: memset failed");
    return WAE_EHIP;
  }
  const size_t lds = (size_t)(D + 512) * sizeof(float);
  hipLaunchKernelGGL(vq_nearest_kernel, dim3(B * Tq), dim3(256), lds, st, lat, emb, idx, quant, sqerr, hist, Dtot, d0, D, Tq, K,
                     mode);
  hipLaunchKernelGGL(vq_stats_kernel, dim3(1), dim3(256), 0, st, sqerr, hist, stats, K, B * Tq, D, c_loss, mode);
  return wae_check_launch("vq_slice");
}

extern "C" int wae_vq_nearest(const float* lat, const float* emb, int64_t* idx, float* quant, float* stats, int32_t* hist,
                              int32_t B, int32_t D, int32_t Tq, int32_t K, float beta, void* stream) {
  // vector_quantization.py:41-43: the forward values of the two loss terms are equal, so vq_loss = (beta + 1) * mse
  return wae_vq_slice(lat, emb, idx, quant, stats, hist, B, D, 0, D, Tq, K, beta + 1.0f, 0, stream);
}

// ---------------------------------------------------------------------------------------------------
// EMA codebook update (vector_quantization.py:190-215 per slice, :275-287): cluster sizes with Laplace smoothing in one
// workgroup, then one workgroup per code sums its latents in row order (no atomics: the update is reproducible).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) vq_ema_sizes_kernel(const int32_t* __restrict__ hist, float* __restrict__ ema_n, int K,
                                                           float decay) {
  __shared__ float part[4];
  float s = 0.f;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float v = ema_n[k] * decay + (1.0f - decay) * (float)hist[k];
    ema_n[k] = v;
    s += v;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  const float n = part[0] + part[1] + part[2] + part[3];
  for (int k = threadIdx.x; k < K; k += 256) ema_n[k] = (ema_n[k] + 1e-5f) / (n + (float)K * 1e-5f) * n;
}
__global__ void __launch_bounds__(64) vq_ema_codes_kernel(const float* __restrict__ lat, const int64_t* __restrict__ idx,
                                                          const float* __restrict__ ema_n, float* __restrict__ ema_w,
                                                          float* __restrict__ emb, int Dtot, int d0, int D, int Tq, int N,
                                                          float decay) {
  const int k = blockIdx.x;
  for (int i = threadIdx.x; i < D; i += 64) {
    float dw = 0.f;
    for (int row = 0; row < N; ++row)
      if (idx[row] == k) dw += lat[((int64_t)(row / Tq) * Dtot + d0 + i) * Tq + row % Tq];
    const float w = ema_w[(int64_t)k * D + i] * decay + (1.0f - decay) * dw;
    ema_w[(int64_t)k * D + i] = w;
    emb[(int64_t)k * D + i] = w / ema_n[k];
  }
}
extern "C" int wae_vq_ema_update(const float* lat, const int64_t* idx, const int32_t* hist, float* ema_cluster_size, float* ema_w,
                                 float* emb, int32_t B, int32_t Dtot, int32_t d0, int32_t D, int32_t Tq, int32_t K, float decay,
                                 void* stream) {
  WAE_REQUIRE(lat && idx && hist && ema_cluster_size && ema_w && emb, "vq_ema_update: null pointer");
  WAE_REQUIRE(B > 0 && D > 0 && Tq > 0 && K > 0 && d0 >= 0 && d0 + D <= Dtot, "vq_ema_update: bad sizes");
  hipStream_t st = as_stream(stream);
  hipLaunchKernelGGL(vq_ema_sizes_kernel, dim3(1), dim3(256), 0, st, hist, ema_cluster_size, K, decay);
  hipLaunchKernelGGL(vq_ema_codes_kernel, dim3(K), dim3(64), 0, st, lat, idx, ema_cluster_size, ema_w, emb, Dtot, d0, D, Tq,
                     B * Tq, decay);
  return wae_check_launch("vq_ema_update");
}

// ---------------------------------------------------------------------------------------------------
// one upsample stage (upsample.py:19-21 nearest stretch + :39-46 shared FIR, zero padded at both ends)
// ---------------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) upsample_stage_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                             void* __restrict__ out, int C, int Tin, int s, int out_btc,
                                                             int Cp) {
  const int Tout = Tin * s;
  const int b = blockIdx.z;
  if (!out_btc) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y;
    if (t >= Tout) return;
    const float* r = in + ((int64_t)b * C + c) * Tin;
    float acc = 0.f;
    for (int j = 0; j <= 2 * s; ++j) {
      const int u = t + j - s;
      if (u >= 0 && u < Tout) acc = fmaf(w[j], r[u / s], acc);
    }
    ((float*)out)[((int64_t)b * C + c) * Tout + t] = acc;
  } else {
    // time-major output: thread -> (t, c) with c fastest so the stores coalesce
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int t = (int)(e / Cp), c = (int)(e % Cp);
    if (t >= Tout) return;
    float acc = 0.f;
    if (c < C) {
      const float* r = in + ((int64_t)b * C + c) * Tin;
      for (int j = 0; j <= 2 * s; ++j) {
        const int u = t + j - s;
        if (u >= 0 && u < Tout) acc = fmaf(w[j], r[u / s], acc);
      }
    }
    store_e<E>(out, ((int64_t)b * Tout + t) * Cp + c, acc);
  }
}

// Last stage (time-major output, audio rate): a block = UPT output steps of one clip.  The input frames those steps touch
// ((UPT + 2s)/s + 2 per channel) are staged in LDS with reads that run along time, then thread (c, t) walks its 2s+1 taps
// (the direct form's loads were one cache line per lane, 64 channel rows apart: 0.1 ms per step at C2).
#define UPT 64
template <typename E>
__global__ void __launch_bounds__(256) upsample_last_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                            void* __restrict__ out, int C, int Tin, int s, int Cp, int nfp) {
  extern __shared__ float sm[];              // [Cp][nfp] input frames, then [2s+1] taps
  const int b = blockIdx.y, t0 = blockIdx.x * UPT, Tout = Tin * s;
  const int fl = max(t0 - s, 0) / s, fh = min(t0 + UPT - 1 + s, Tout - 1) / s, nf = fh - fl + 1;
  float* taps = sm + Cp * nfp;
  for (int i = threadIdx.x; i < Cp * nf; i += 256) {
    const int c = i / nf, f = i - c * nf;
    sm[c * nfp + f] = c < C ? in[((int64_t)b * C + c) * Tin + fl + f] : 0.f;
  }
  if (threadIdx.x <= 2 * s) taps[threadIdx.x] = w[threadIdx.x];
  __syncthreads();
  // An output step whose 2s + 1 taps all fall inside the sequence touches three input frames f - 1, f, f + 1 (f = t / s) with the taps
  // summed per frame: A[r] = sum_{j < s - r} w[j], B[r] = the next s taps, C[r] = the rest (r = t % s) -- three FMAs instead of a walk
  // over 2s + 1 taps with an incremental (u / s, u % s): the walk was what this launch spent its 27 us on.  (Round 4; the sums reorder
  // the reference's FIR additions -- 1e-7 relative, far below a 16-bit output's rounding -- so fp32 launches and the s steps at either end of a
  // clip keep the tap-by-tap form.)
  float* co3 = taps + 2 * s + 1;             // [3][s]
  if (threadIdx.x < 3 * s) {
    const int which = threadIdx.x / s, rr = threadIdx.x - which * s;
    const int ja = which == 0 ? 0 : (which == 1 ? s - rr : 2 * s - rr), jb = which == 0 ? s - rr : (which == 1 ? 2 * s - rr : 2 * s + 1);
    float a = 0.f;
    for (int j = ja; j < jb; ++j) a += taps[j];
    co3[threadIdx.x] = a;
  }
  __syncthreads();
  const int c = threadIdx.x % Cp, tl = threadIdx.x / Cp, tstep = 256 / Cp;
  const float* r = sm + c * nfp - fl;
  for (int t = t0 + tl; t < min(t0 + UPT, Tout); t += tstep) {
    float acc = 0.f;
    if (c < C) {
      if (sizeof(E) == 2 && t >= s && t + s < Tout) {   // (fp32, the parity mode, keeps the reference's summation order everywhere)
        const int f = t / s, rr = t - f * s;
        acc = co3[rr] * r[f - 1];
        acc = fmaf(co3[s + rr], r[f], acc);
        acc = fmaf(co3[2 * s + rr], r[f + 1], acc);
      } else {
        int j0 = max(0, s - t), u = t + j0 - s;      // first tap with u >= 0
        int q = u / s, rem = u - q * s;
        for (int j = j0; j <= 2 * s && u < Tout; ++j, ++u) {
          acc = fmaf(taps[j], r[q], acc);
          if (++rem == s) { rem = 0; ++q; }
        }
      }
    }
    store_e<E>(out, ((int64_t)b * Tout + t) * Cp + c, acc);
  }
}

extern "C" int wae_upsample_stage_fwd(const float* in, const float* w, void* out, int32_t B, int32_t C, int32_t Tin,
                                      int32_t s, int32_t out_btc, int32_t Cp, int32_t dtype, void* stream) {
  WAE_REQUIRE(in && w && out && B > 0 && C > 0 && Tin > 0 && s > 0, "upsample_stage: bad arguments");
  const int Tout = Tin * s;
  hipStream_t st = as_stream(stream);
  if (!out_btc) {
    hipLaunchKernelGGL(upsample_stage_kernel<float>, dim3((Tout + 255) / 256, C, B), dim3(256), 0, st, in, w, out, C, Tin,
                       s, 0, C);
  } else {
    WAE_REQUIRE(Cp >= C, "upsample_stage: Cp < C");
    if (Cp <= 256 && 256 % Cp == 0) {
      const int nfp = UPT / s + 5;           // frames per channel row (+1: odd pitch against bank conflicts)
      const size_t lds = ((size_t)Cp * (nfp | 1) + 2 * s + 1 + 3 * s) * sizeof(float);
      WAE_REQUIRE(3 * s <= 256, "upsample_stage: scale %d too large for the last-stage kernel", s);
      dim3 g2((Tout + UPT - 1) / UPT, B);
      if (dtype == WAE_BF16)
        hipLaunchKernelGGL(upsample_last_kernel<__bf16>, g2, dim3(256), lds, st, in, w, out, C, Tin, s, Cp, nfp | 1);
      else if (dtype == WAE_F16)
        hipLaunchKernelGGL(upsample_last_kernel<f16>, g2, dim3(256), lds, st, in, w, out, C, Tin, s, Cp, nfp | 1);
      else
        hipLaunchKernelGGL(upsample_last_kernel<float>, g2, dim3(256), lds, st, in, w, out, C, Tin, s, Cp, nfp | 1);
      return wae_check_launch("upsample_stage_fwd");
    }
    const int64_t n = (int64_t)Tout * Cp;
    dim3 grid((unsigned)((n + 255) / 256), 1, B);
    if (dtype == WAE_BF16)
      hipLaunchKernelGGL(upsample_stage_kernel<__bf16>, grid, dim3(256), 0, st, in, w, out, C, Tin, s, 1, Cp);
    else if (dtype == WAE_F16)
      hipLaunchKernelGGL(upsample_stage_kernel<f16>, grid, dim3(256), 0, st, in, w, out, C, Tin, s, 1, Cp);
    else
      hipLaunchKernelGGL(upsample_stage_kernel<float>, grid, dim3(256), 0, st, in, w, out, C, Tin, s, 1, Cp);
  }
  return wae_check_launch("upsample_stage_fwd");
}

// ---------------------------------------------------------------------------------------------------
// hoisted global conditioning (modules.py:148-152 re-convolves the same g at every t; exact hoist)
// zb[b][l][row] , rows 0..Hp-1 = gate-a channels, Hp..2Hp-1 = gate-b channels (zero in the padding)
// Ids index tables (the first-conv table, the speaker embedding): the reference raises IndexError from nn.Embedding / the
// one-hot encoder for an id outside its table.  Here an id outside [0, n) is clamped -- no access ever leaves the table --
// and reported through a sticky device word that the host turns into that IndexError at its next checkpoint.
__device__ __forceinline__ int checked_id(int id, int n, int32_t* err, int code) {
  if ((unsigned)id < (unsigned)n) return id;
  if (err) atomicOr(err, code);
  return id < 0 ? 0 : n - 1;
}

__global__ void __launch_bounds__(256) check_ids_kernel(const int32_t* __restrict__ ids, int64_t n, int lo, int hi,
                                                        int32_t* __restrict__ err, int code) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad |= ids[i] < lo || ids[i] >= hi;
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(err, code);
}
extern "C" int wae_check_ids(const int32_t* ids, int64_t n, int32_t lo, int32_t hi, int32_t* err, int32_t code, void* stream) {
  WAE_REQUIRE(ids && err && n > 0 && hi > lo, "check_ids: bad arguments");
  const int grid = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(check_ids_kernel, dim3(grid), dim3(256), 0, as_stream(stream), ids, n, lo, hi, err, code);
  return wae_check_launch("check_ids");
}

// One-hot (B, C, T) fp32 inputs -> class ids (wavenet.py:203 applies first_conv to whatever (B, C, T) tensor it is handed; the
// teacher-forced kernels gather rows of its weight by class id, which is the same arithmetic for ONE-HOT columns only).  A column
// that is not exactly one 1.0 among zeros sets `code` in the sticky error word: the host refuses the call instead of silently
// taking an argmax of soft labels.  Strides in elements, so (B, T, C) views need no copy.
__global__ void __launch_bounds__(256) onehot_to_ids_kernel(const float* __restrict__ x, int64_t n, int C, int T, int64_t sb, int64_t sc,
                                                            int64_t st, int32_t* __restrict__ ids, int32_t* __restrict__ err, int code) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / T, t = i - b * T;
    const float* col = x + b * sb + t * st;
    int ones = 0, best = 0;
    bool other = false;
    for (int c = 0; c < C; ++c) {
      const float v = col[(int64_t)c * sc];
      if (v == 1.0f) { ++ones; best = c; }
      else other |= v != 0.0f;
    }
    ids[i] = best;
    bad |= ones != 1 || other;
  }
  if (__any(bad) && err && (threadIdx.x & 63) == 0) atomicOr(err, code);
}
extern "C" int wae_onehot_to_ids(const float* x, int32_t B, int32_t C, int32_t T, int64_t stride_b, int64_t stride_c, int64_t stride_t,
                                 int32_t* ids, int32_t* err, int32_t code, void* stream) {
  WAE_REQUIRE(x && ids && B > 0 && C > 0 && T > 0, "onehot_to_ids: bad arguments");
  const int64_t n = (int64_t)B * T;
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(onehot_to_ids_kernel, dim3(grid), dim3(256), 0, as_stream(stream), x, n, C, T, stride_b, stride_c, stride_t, ids, err, code);
  return wae_check_launch("onehot_to_ids");
}

// ---------------------------------------------------------------------------------------------------
// A stream that starts `us` microseconds late: the second of two half-batch chains of layer launches (engine.py: chain_plan) is
// started half a launch behind the first, so that the store bursts of one chain's launches meet the GEMM phases of the other's.
// One wave that sleeps on the constant-rate wall clock (s_memrealtime); no memory traffic.
__global__ void __launch_bounds__(64) stream_delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
}

extern "C" int wae_stream_delay(double us, void* stream) {
  WAE_REQUIRE(us >= 0 && us <= 1e6, "stream_delay: %g us is out of range [0, 1 s]", us);
  static int khz = 0;
  if (khz == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) {
      khz = 0;
      wae_set_error("stream_delay: cannot read the wall clock rate");
      return WAE_EHIP;
    }
  }
  const unsigned long long ticks = (unsigned long long)(us * khz / 1000.0);
  if (ticks == 0) return WAE_OK;
  hipLaunchKernelGGL(stream_delay_kernel, dim3(1), dim3(64), 0, as_stream(stream), ticks);
  return wae_check_launch("stream_delay");
}

// ---------------------------------------------------------------------------------------------------
// zb[b][l][:] = conv bias + conv1x1g(g_b) (modules.py:148-152 hoisted out of the time loop).  Block = (layer, clip, 32 gate rows);
// eight lanes share a row and walk its Cg weights 32 bytes at a time (one thread per row walked Cg dependent strided loads: 21 us).
__global__ void __launch_bounds__(256) gproj_fwd_kernel(const float* __restrict__ eff, int64_t wg_off, int64_t bias_off,
                                                        int64_t layer_stride, const int32_t* __restrict__ gid,
                                                        int64_t emb_off, const float* __restrict__ gvec,
                                                        float* __restrict__ zb, int L, int G, int Hp, int Cg, int n_speakers,
                                                        int32_t* __restrict__ err) {
  const int l = blockIdx.x, b = blockIdx.y;
  const int H = G / 2;
  const float* wg = wg_off >= 0 ? eff + wg_off + (int64_t)l * layer_stride : nullptr;
  const float* bs = eff + bias_off + (int64_t)l * layer_stride;
  const int sp = gid ? checked_id(gid[b], n_speakers, err, WAE_ERR_SPEAKER_ID) : 0;
  const float* gv = gid ? eff + emb_off + (int64_t)sp * Cg : (gvec ? gvec + (int64_t)b * Cg : nullptr);
  float* o = zb + ((int64_t)b * L + l) * 2 * Hp;
  const int r = blockIdx.z * 32 + (threadIdx.x >> 3), cs = threadIdx.x & 7;
  const int half = r >= Hp, i = r - half * Hp;
  const bool live = r < 2 * Hp && i < H;
  float acc = 0.f;
  if (live && wg && gv) {
    const float* wr = wg + (int64_t)(half * H + i) * Cg;
    for (int c = cs; c < Cg; c += 8) acc = fmaf(wr[c], gv[c], acc);
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  if (cs == 0 && r < 2 * Hp) o[r] = live ? acc + bs[half * H + i] : 0.f;
}

extern "C" int wae_gproj_fwd(const float* eff, int64_t wg_off, int64_t bias_off, int64_t layer_stride, const int32_t* gid,
                             int64_t emb_off, const float* gvec, float* zb, int32_t B, int32_t L, int32_t G, int32_t Hp,
                             int32_t Cg, int32_t n_speakers, int32_t* err, void* stream) {
  WAE_REQUIRE(eff && zb && B > 0 && L > 0 && G > 0 && G % 2 == 0 && Hp >= G / 2, "gproj: bad arguments");
  WAE_REQUIRE(!gid || n_speakers > 0, "gproj: speaker ids need the size of the embedding table");
  hipLaunchKernelGGL(gproj_fwd_kernel, dim3(L, B, (2 * Hp + 31) / 32), dim3(256), 0, as_stream(stream), eff, wg_off, bias_off,
                     layer_stride, gid, emb_off, gvec, zb, L, G, Hp, Cg, n_speakers, err);
  return wae_check_launch("gproj_fwd");
}

// backward of gproj: the per-clip column sums of dz (the "ones columns" of the dW1 tile of wae_gemm_tn) are the
// gradient of zb[b][l][:]; chain into the conv bias, conv1x1g and the speaker embedding (modules.py:148-152).
// Block = (layer, 32 gate rows).  The block's 32 x B sums and the B conditioning vectors are staged in LDS once (the first form
// re-read both -- and the speaker id -- from global memory inside the B x Cg loops: 45 us of dependent loads per step).
#define GPROJ_BMAX 32
__global__ void __launch_bounds__(256) gproj_bwd_kernel(const float* __restrict__ eff, float* __restrict__ d_eff, int64_t wg_off,
                                                        int64_t bias_off, int64_t layer_stride, const int32_t* __restrict__ gid,
                                                        int64_t emb_off, const float* __restrict__ gvec,
                                                        const float* __restrict__ c1, int64_t c_layer_stride, int64_t ld,
                                                        int ones_col, int B0, int B, int G, int Hp, int Cg, int n_speakers) {
  extern __shared__ float gp_lds[];
  float* s_cl = gp_lds;                      // [32][GPROJ_BMAX]
  float* s_e = gp_lds + 32 * GPROJ_BMAX;     // [nb][Cg]
  __shared__ int s_sp[GPROJ_BMAX];
  const int l = blockIdx.x;
  const int H = G / 2;
  const float* cl = c1 + (int64_t)l * c_layer_stride + ones_col;
  const bool proj = wg_off >= 0 && Cg > 0;
  const int nb = min(B - B0, GPROJ_BMAX);
  // thread = (row, feature slice) of phase 1; its slots of d_eff are fetched now, under the staging trips
  const int rr1 = threadIdx.x >> 3, r1 = blockIdx.y * 32 + rr1, cs = threadIdx.x & 7;
  const int half1 = r1 >= Hp, i1 = r1 - half1 * Hp;
  const bool live1 = r1 < 2 * Hp && i1 < H;
  const int ch1 = half1 * H + i1;
  float* dwg = d_eff + wg_off + (int64_t)l * layer_stride + (int64_t)ch1 * Cg;
  float* dbias = d_eff + bias_off + (int64_t)l * layer_stride + ch1;
  constexpr int PRE = 16;                    // Cg <= 128: the thread's Cg / 8 slots stay in registers
  float pre[PRE], pre_b = 0.f;
  const bool use_pre = Cg <= 8 * PRE;
  if (live1) {
    if (cs == 0) pre_b = *dbias;
    if (proj && use_pre) {
#pragma unroll
      for (int k = 0; k < PRE; ++k) pre[k] = cs + 8 * k < Cg ? dwg[cs + 8 * k] : 0.f;
    }
  }
  for (int e = threadIdx.x; e < 32 * nb; e += 256) {
    const int rr = e / nb, b = e - rr * nb, r = blockIdx.y * 32 + rr;
    s_cl[rr * GPROJ_BMAX + b] = r < 2 * Hp ? cl[(int64_t)r * ld + B0 + b] : 0.f;
  }
  if (gid && threadIdx.x < nb) s_sp[threadIdx.x] = checked_id(gid[B0 + threadIdx.x], n_speakers, nullptr, 0);
  __syncthreads();
  if (proj)
    for (int e = threadIdx.x; e < nb * Cg; e += 256) {
      const int b = e / Cg, c = e - b * Cg;
      s_e[e] = gid ? eff[emb_off + (int64_t)s_sp[b] * Cg + c] : gvec[(int64_t)(B0 + b) * Cg + c];
    }
  __syncthreads();
  // phase 1: bias gradient and the row of dWg
  if (live1) {
    if (cs == 0) {
      float sb = 0.f;
      for (int b = 0; b < nb; ++b) sb += s_cl[rr1 * GPROJ_BMAX + b];
      *dbias = pre_b + sb;                                         // this (layer,row) slot is touched by one thread only
    }
    if (proj) {
      if (use_pre) {
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
          const int c = cs + 8 * k;
          if (c < Cg) {
            float a = 0.f;
            for (int b = 0; b < nb; ++b) a = fmaf(s_cl[rr1 * GPROJ_BMAX + b], s_e[b * Cg + c], a);
            dwg[c] = pre[k] + a;
          }
        }
      } else {
        for (int c = cs; c < Cg; c += 8) {
          float a = 0.f;
          for (int b = 0; b < nb; ++b) a = fmaf(s_cl[rr1 * GPROJ_BMAX + b], s_e[b * Cg + c], a);
          dwg[c] += a;
        }
      }
    }
  }
  // phase 2: embedding rows: thread per (clip, feature) reduces over THIS block's 32 gate rows -> one atomic each
  if (gid && proj) {
    for (int e = threadIdx.x; e < nb * Cg; e += 256) {
      const int b = e / Cg, c = e - b * Cg;
      float wv[32];
#pragma unroll
      for (int rr = 0; rr < 32; ++rr) {
        const int r = blockIdx.y * 32 + rr;
        const int half = r >= Hp, i = r - half * Hp;
        wv[rr] = (r < 2 * Hp && i < H) ? eff[wg_off + (int64_t)l * layer_stride + (int64_t)(half * H + i) * Cg + c] : 0.f;
      }
      float a = 0.f;
#pragma unroll
      for (int rr = 0; rr < 32; ++rr) a = fmaf(s_cl[rr * GPROJ_BMAX + b], wv[rr], a);
      atomicAdd(d_eff + emb_off + (int64_t)s_sp[b] * Cg + c, a);
    }
  }
}
extern "C" int wae_gproj_bwd(const float* eff, float* d_eff, int64_t wg_off, int64_t bias_off, int64_t layer_stride,
                             const int32_t* gid, int64_t emb_off, const float* gvec, const float* c1, int64_t c_layer_stride,
                             int64_t ld, int32_t ones_col, int32_t B, int32_t L, int32_t G, int32_t Hp, int32_t Cg,
                             int32_t n_speakers, void* stream) {
  WAE_REQUIRE(eff && d_eff && c1 && B > 0 && L > 0 && G > 0 && G % 2 == 0, "gproj_bwd: bad arguments");
  WAE_REQUIRE(!gid || n_speakers > 0, "gproj_bwd: speaker ids need the size of the embedding table");
  const int64_t wg = (gid || gvec) ? wg_off : -1;
  const size_t lds = (32 * GPROJ_BMAX + (size_t)GPROJ_BMAX * (Cg > 0 ? Cg : 0)) * sizeof(float);
  WAE_REQUIRE(lds <= 64 * 1024, "gproj_bwd: Cg = %d too wide for the staged form", Cg);
  for (int b0 = 0; b0 < B; b0 += GPROJ_BMAX) {   // clips in groups of 32 (the sums of a group stay in LDS)
    hipLaunchKernelGGL(gproj_bwd_kernel, dim3(L, (2 * Hp + 31) / 32), dim3(256), lds, as_stream(stream), eff, d_eff, wg, bias_off,
                       layer_stride, gid, emb_off, gvec, c1, c_layer_stride, ld, ones_col, b0, B, G, Hp, Cg, n_speakers);
  }
  return wae_check_launch("gproj_bwd");
}

// sum of the per-layer skip biases (the head's GEMM 0 starts from it)
__global__ void __launch_bounds__(1024) sum_rows_kernel(const float* __restrict__ src, int64_t off, int64_t stride, int L, int n,
                                                       int n_pad, float* __restrict__ out) {
  // block = 64 columns x 16 row groups (the serial form walked up to 256 dependent loads per thread: 35-70 us per call)
  __shared__ float part[16][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
  float s = 0.f;
  if (col < n)
    for (int l = grp; l < L; l += 16) s += src[off + (int64_t)l * stride + col];
  part[grp][threadIdx.x & 63] = s;
  __syncthreads();
  if (grp == 0 && col < n_pad) {
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += part[g][threadIdx.x];
    out[col] = col < n ? t : 0.f;
  }
}
extern "C" int wae_sum_rows(const float* src, int64_t off, int64_t stride, int32_t L, int32_t n, int32_t n_pad, float* out,
                            void* stream) {
  WAE_REQUIRE(src && out && L > 0 && n > 0 && n_pad >= n, "sum_rows: bad arguments");
  hipLaunchKernelGGL(sum_rows_kernel, dim3((n_pad + 63) / 64), dim3(1024), 0, as_stream(stream), src, off, stride, L, n,
                     n_pad, out);
  return wae_check_launch("sum_rows");
}

// ---------------------------------------------------------------------------------------------------
// first_conv on a one-hot input == column gather + bias (wavenet.py:203); scalar input: w*x + b
// ---------------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) first_conv_kernel(const int32_t* __restrict__ idx, const float* __restrict__ xs,
                                                         const float* __restrict__ table, const float* __restrict__ bias,
                                                         void* __restrict__ x0, int64_t BT, int Rp, int O, int32_t* __restrict__ err) {
  // thread = 8 consecutive residual channels of one sample (Rp is a multiple of 128): 16-byte table / bias loads, one
  // 16-byte bf16 store; a block covers 16 samples
  const int tpr = Rp >> 3;
  for (int i = threadIdx.x; i < 16 * tpr; i += 256) {
    const int row = i / tpr, r = (i - row * tpr) * 8;
    const int64_t bt = (int64_t)blockIdx.x * 16 + row;
    if (bt >= BT) return;
    f32x4 v0, v1;
    const f32x4 b0 = *(const f32x4*)(bias + r), b1 = *(const f32x4*)(bias + r + 4);
    if (idx) {
      const float* tr = table + (int64_t)checked_id(idx[bt], O, err, WAE_ERR_CLASS_ID) * Rp + r;
      v0 = *(const f32x4*)tr + b0;
      v1 = *(const f32x4*)(tr + 4) + b1;
    } else {
      const float x = xs[bt];
      const f32x4 t0 = *(const f32x4*)(table + r), t1 = *(const f32x4*)(table + r + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v0[k] = fmaf(t0[k], x, b0[k]); v1[k] = fmaf(t1[k], x, b1[k]); }
    }
    if constexpr (sizeof(E) == 2) {
      typename ET<E>::frag o;
#pragma unroll
      for (int k = 0; k < 4; ++k) { o[k] = (E)v0[k]; o[4 + k] = (E)v1[k]; }
      *(typename ET<E>::frag*)((E*)x0 + bt * Rp + r) = o;
    } else {
      *(f32x4*)((float*)x0 + bt * Rp + r) = v0;
      *(f32x4*)((float*)x0 + bt * Rp + r + 4) = v1;
    }
  }
}

extern "C" int wae_first_conv_fwd(const int32_t* idx, const float* xs, const float* table, const float* bias, void* x0,
                                  int64_t BT, int32_t Rp, int32_t O, int32_t dtype, int32_t* err, void* stream) {
  WAE_REQUIRE((idx || xs) && table && bias && x0 && BT > 0 && Rp > 0 && Rp % 8 == 0, "first_conv: bad arguments");
  WAE_REQUIRE(!idx || O > 0, "first_conv: class ids need the number of classes");
  dim3 grid((unsigned)((BT + 15) / 16));
  if (dtype == WAE_BF16)
    hipLaunchKernelGGL(first_conv_kernel<__bf16>, grid, dim3(256), 0, as_stream(stream), idx, xs, table, bias, x0, BT, Rp, O, err);
  else if (dtype == WAE_F16)
    hipLaunchKernelGGL(first_conv_kernel<f16>, grid, dim3(256), 0, as_stream(stream), idx, xs, table, bias, x0, BT, Rp, O, err);
  else
    hipLaunchKernelGGL(first_conv_kernel<float>, grid, dim3(256), 0, as_stream(stream), idx, xs, table, bias, x0, BT, Rp, O, err);
  return wae_check_launch("first_conv_fwd");
}

// one-hot rows of the class ids, (n, width) in the storage dtype: the first conv's weight gradient sum_t onehot(id[t]) (x) dx0[t]
// (autograd of wavenet.py:203) then is one more P^T Q contraction of the weight-gradient launch (csrc/gemm_tn_static.hip).  An id
// outside [0, width) gives a zero row (the forward has flagged it).  thread = 8 columns of one row, one 16-byte store.
template <typename E>
__global__ void __launch_bounds__(256) onehot_rows_kernel(const int32_t* __restrict__ ids, E* __restrict__ out, int64_t n, int width) {
  const int tpr = width >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * tpr) return;
  const int64_t row = i / tpr;
  const int c0 = (int)(i - row * tpr) * 8;
  const int id = ids[row];
  E v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (E)(c0 + k == id ? 1.0f : 0.0f);
  E* o = out + row * width + c0;
#pragma unroll
  for (int k = 0; k < 8; ++k) o[k] = v[k];
}
extern "C" int wae_onehot_rows(const int32_t* ids, void* out, int64_t n, int32_t width, int32_t dtype, void* stream) {
  WAE_REQUIRE(ids && out && n > 0 && width > 0 && width % 8 == 0 && wae_dtype_ok(dtype), "onehot_rows: bad arguments");
  const int64_t threads = n * (width / 8);
  const dim3 grid((unsigned)((threads + 255) / 256));
  if (dtype == WAE_BF16) hipLaunchKernelGGL(onehot_rows_kernel<__bf16>, grid, dim3(256), 0, as_stream(stream), ids, (__bf16*)out, n, width);
  else if (dtype == WAE_F16) hipLaunchKernelGGL(onehot_rows_kernel<f16>, grid, dim3(256), 0, as_stream(stream), ids, (f16*)out, n, width);
  else hipLaunchKernelGGL(onehot_rows_kernel<float>, grid, dim3(256), 0, as_stream(stream), ids, (float*)out, n, width);
  return wae_check_launch("onehot_rows");
}

// ---------------------------------------------------------------------------------------------------
// layout converters (B,C,T) fp32 <-> (B,T,Cp) dtype through a 64x64 LDS tile (coalesced on both sides)
// ---------------------------------------------------------------------------------------------------
template <typename E>
__global__ void __launch_bounds__(256) to_btc_kernel(const float* __restrict__ in, void* __restrict__ out, int C, int T,
                                                     int Cp, const int32_t* __restrict__ lengths, float scale, int masked) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  for (int i = ly; i < 64; i += 4) {
    const int c = c0 + i, t = t0 + lx;
    tile[i][lx] = (c < C && t < T) ? in[((int64_t)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  // masked: row t carries the loss of target t + 1 (vqwae_train.py:764-766) -- weight `scale` while t + 1 < length, else 0
  const int len = masked ? (lengths ? min(lengths[b], T) : T) : 0;
  for (int i = ly; i < 64; i += 4) {
    const int t = t0 + i, c = c0 + lx;
    if (t < T && c < Cp) {
      float v = tile[lx][i];
      if (masked) v = t + 1 < len ? v * scale : 0.f;
      store_e<E>(out, ((int64_t)b * T + t) * Cp + c, v);
    }
  }
}
template <typename E>
__global__ void __launch_bounds__(256) from_btc_kernel(const void* __restrict__ in, float* __restrict__ out, int C, int T,
                                                       int Cp, float scale) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
  for (int i = ly; i < 64; i += 4) {
    const int t = t0 + i, c = c0 + lx;
    tile[i][lx] = (t < T && c < Cp) ? load_e<E>(in, ((int64_t)b * T + t) * Cp + c) : 0.f;
  }
  __syncthreads();
  for (int i = ly; i < 64; i += 4) {
    const int c = c0 + i, t = t0 + lx;
    if (c < C && t < T) out[((int64_t)b * C + c) * T + t] = tile[lx][i] * scale;
  }
}

extern "C" int wae_to_btc(const float* in, void* out, int32_t B, int32_t C, int32_t T, int32_t Cp, int32_t dtype,
                          void* stream) {
  WAE_REQUIRE(in && out && B > 0 && C > 0 && T > 0 && Cp >= C, "to_btc: bad arguments");
  dim3 grid((T + 63) / 64, (Cp + 63) / 64, B);
  if (dtype == WAE_BF16)
    hipLaunchKernelGGL(to_btc_kernel<__bf16>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, nullptr, 1.f, 0);
  else if (dtype == WAE_F16)
    hipLaunchKernelGGL(to_btc_kernel<f16>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, nullptr, 1.f, 0);
  else
    hipLaunchKernelGGL(to_btc_kernel<float>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, nullptr, 1.f, 0);
  return wae_check_launch("to_btc");
}
// ---- upsample_activation (upsample.py:44-46: getattr(nn, upsample_activation)(**params) behind every stage's smoothing FIR) ----------
// kind 1 ReLU, 2 LeakyReLU(slope), 3 Tanh, 4 Sigmoid.  Forward in place on the stage's fp32 output; backward from the activation's
// OUTPUT (all four derivatives are functions of it: sign, 1 - y^2, y (1 - y)), which is the next stage's saved input anyway.
__global__ void __launch_bounds__(256) act_fwd_kernel(float* __restrict__ x, int64_t n, int kind, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = x[i];
    float y;
    if (kind == 1) y = fmaxf(v, 0.f);
    else if (kind == 2) y = v > 0.f ? v : v * slope;
    else if (kind == 3) y = tanhf(v);
    else y = 1.0f / (1.0f + expf(-v));
    x[i] = y;
  }
}
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ y, float* __restrict__ d, int64_t n, int kind, float slope) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = y[i];
    float g;
    if (kind == 1) g = v > 0.f ? 1.f : 0.f;
    else if (kind == 2) g = v > 0.f ? 1.f : slope;
    else if (kind == 3) g = 1.f - v * v;
    else g = v * (1.f - v);
    d[i] *= g;
  }
}
extern "C" int wae_act_fwd(float* x, int64_t n, int32_t kind, float slope, void* stream) {
  WAE_REQUIRE(x && n > 0 && kind >= 1 && kind <= 4, "act_fwd: bad arguments (kind 1 ReLU, 2 LeakyReLU, 3 Tanh, 4 Sigmoid)");
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(act_fwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), x, n, kind, slope);
  return wae_check_launch("act_fwd");
}
extern "C" int wae_act_bwd(const float* y, float* d, int64_t n, int32_t kind, float slope, void* stream) {
  WAE_REQUIRE(y && d && n > 0 && kind >= 1 && kind <= 4, "act_bwd: bad arguments");
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), y, d, n, kind, slope);
  return wae_check_launch("act_bwd");
}

extern "C" int wae_to_btc_masked(const float* in, void* out, int32_t B, int32_t C, int32_t T, int32_t Cp, int32_t dtype,
                                 const int32_t* lengths, float scale, void* stream) {
  WAE_REQUIRE(in && out && B > 0 && C > 0 && T > 0 && Cp >= C, "to_btc_masked: bad arguments");
  WAE_REQUIRE(wae_dtype_ok(dtype), "to_btc_masked: bad dtype");
  dim3 grid((T + 63) / 64, (Cp + 63) / 64, B);
  if (dtype == WAE_BF16)
    hipLaunchKernelGGL(to_btc_kernel<__bf16>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, lengths, scale, 1);
  else if (dtype == WAE_F16)
    hipLaunchKernelGGL(to_btc_kernel<f16>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, lengths, scale, 1);
  else
    hipLaunchKernelGGL(to_btc_kernel<float>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, lengths, scale, 1);
  return wae_check_launch("to_btc_masked");
}
extern "C" int wae_from_btc_scaled(const void* in, float* out, int32_t B, int32_t C, int32_t T, int32_t Cp, int32_t dtype,
                                   float scale, void* stream) {
  WAE_REQUIRE(in && out && B > 0 && C > 0 && T > 0 && Cp >= C, "from_btc: bad arguments");
  dim3 grid((T + 63) / 64, (Cp + 63) / 64, B);
  if (dtype == WAE_BF16)
    hipLaunchKernelGGL(from_btc_kernel<__bf16>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, scale);
  else if (dtype == WAE_F16)
    hipLaunchKernelGGL(from_btc_kernel<f16>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, scale);
  else
    hipLaunchKernelGGL(from_btc_kernel<float>, grid, dim3(256), 0, as_stream(stream), in, out, C, T, Cp, scale);
  return wae_check_launch("from_btc");
}
extern "C" int wae_from_btc(const void* in, float* out, int32_t B, int32_t C, int32_t T, int32_t Cp, int32_t dtype,
                            void* stream) {
  return wae_from_btc_scaled(in, out, B, C, T, Cp, dtype, 1.0f, stream);
}

// ---------------------------------------------------------------------------------------------------
// masked mean (vqwae_train.py:379): sum_{b, t < len[b]-1} nll[b,t] / sum_b max(len[b]-1, 0)
// the mask is sequence_mask(lengths)[:,1:] applied to positions 0..T-2 (loss position t predicts y[t+1]).
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) masked_mean_kernel(const float* __restrict__ nll, const int32_t* __restrict__ lengths,
                                                           float* __restrict__ out, int B, int T) {
  __shared__ double part[16];
  double s = 0.0;
  for (int b = 0; b < B; ++b) {
    const int len = lengths ? min(lengths[b], T) : T;
    const float* r = nll + (int64_t)b * T;
    // four independent loads in flight per thread (one dependent fp64 chain per load took 26 us per step)
    int t = threadIdx.x;
    for (; t + 3072 < len - 1; t += 4096) {
      const float a0 = r[t], a1 = r[t + 1024], a2 = r[t + 2048], a3 = r[t + 3072];
      s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
    }
    for (; t < len - 1; t += 1024) s += (double)r[t];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot = 0.0, cnt = 0.0;
    for (int i = 0; i < 16; ++i) tot += part[i];
    for (int b = 0; b < B; ++b) {
      const int len = lengths ? min(lengths[b], T) : T;
      cnt += (double)max(len - 1, 0);
    }
    out[0] = (float)(tot / cnt);
    out[1] = (float)cnt;
  }
}

extern "C" int wae_masked_mean(const float* nll, const int32_t* lengths, float* out, int32_t B, int32_t T, void* stream) {
  WAE_REQUIRE(nll && out && B > 0 && T > 0, "masked_mean: bad arguments");
  hipLaunchKernelGGL(masked_mean_kernel, dim3(1), dim3(1024), 0, as_stream(stream), nll, lengths, out, B, T);
  return wae_check_launch("masked_mean");
}

// ---------------------------------------------------------------------------------------------------
// dropout in front of the dilated convolution (modules.py:127-128: x = F.dropout(x, p, training)), training with p > 0 only.
// The mask is a counter-based hash of (seed, element index) -- keep iff the top 24 bits of the mix are >= p * 2^24 -- so the
// forward kernel and the backward kernel regenerate the same mask and the CPU oracle can restate it (oracle: dropout_keep).
// forward:  xd = x * keep / (1 - p)          (xd is the conv operand of wae_glu_layer_fwd_drop and the Q operand of dW1)
// backward: out = alpha * (g_next + acc * keep / (1 - p))    (acc = sum_taps W1_tap^T dz, wae_gemm_tm mode 0)
// ---------------------------------------------------------------------------------------------------
// `key` = dropout_key(seed), formed on the host once per launch: the seed goes through a full 64-bit finaliser BEFORE it meets the
// element index, so the masks of consecutive seeds (layer l and l + 1, step n and n + 1, rank r and r + 1) are unrelated.  (Round 2
// hashed e + seed: the mask of seed + 1 was the mask of seed shifted by one element.)
static inline uint64_t dropout_key(uint64_t seed) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ bool dropout_keep(uint64_t key, uint64_t e, uint32_t thr) {
  uint64_t h = (e ^ key) * 0x9E3779B97F4A7C15ull;
  h ^= h >> 32;
  h *= 0xD6E8FEB86659FD93ull;
  h ^= h >> 32;
  return (uint32_t)(h >> 40) >= thr;
}
template <typename E>
__global__ void __launch_bounds__(256) dropout_fwd_kernel(const void* __restrict__ x, void* __restrict__ xd, int64_t n, uint64_t key,
                                                          uint32_t thr, float scale) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    store_e<E>(xd, i, dropout_keep(key, (uint64_t)i, thr) ? load_e<E>(x, i) * scale : 0.f);
}
template <typename E>
__global__ void __launch_bounds__(256) dropout_bwd_kernel(const void* __restrict__ acc, const void* __restrict__ g_next,
                                                          void* __restrict__ out, int64_t n, uint64_t key, uint32_t thr, float scale,
                                                          float alpha) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float a = dropout_keep(key, (uint64_t)i, thr) ? load_e<E>(acc, i) * scale : 0.f;
    store_e<E>(out, i, alpha * (load_e<E>(g_next, i) + a));
  }
}
static int dropout_args(float p, uint32_t* thr, float* scale) {
  WAE_REQUIRE(p >= 0.f && p < 1.f, "dropout: p must be in [0, 1)");
  *thr = (uint32_t)((double)p * 16777216.0 + 0.5);
  *scale = 1.0f / (1.0f - p);
  return WAE_OK;
}
extern "C" int wae_dropout_fwd(const void* x, void* xd, int64_t n, uint64_t seed, float p, int32_t dtype, void* stream) {
  WAE_REQUIRE(x && xd && n > 0 && wae_dtype_ok(dtype), "dropout_fwd: bad arguments");
  uint32_t thr; float scale;
  if (int rc = dropout_args(p, &thr, &scale); rc != WAE_OK) return rc;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  if (dtype == WAE_BF16) hipLaunchKernelGGL(dropout_fwd_kernel<__bf16>, dim3(grid), dim3(256), 0, as_stream(stream), x, xd, n, dropout_key(seed), thr, scale);
  else if (dtype == WAE_F16) hipLaunchKernelGGL(dropout_fwd_kernel<f16>, dim3(grid), dim3(256), 0, as_stream(stream), x, xd, n, dropout_key(seed), thr, scale);
  else hipLaunchKernelGGL(dropout_fwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), x, xd, n, dropout_key(seed), thr, scale);
  return wae_check_launch("dropout_fwd");
}
extern "C" int wae_dropout_bwd(const void* acc, const void* g_next, void* out, int64_t n, uint64_t seed, float p, float alpha,
                               int32_t dtype, void* stream) {
  WAE_REQUIRE(acc && g_next && out && n > 0 && wae_dtype_ok(dtype), "dropout_bwd: bad arguments");
  uint32_t thr; float scale;
  if (int rc = dropout_args(p, &thr, &scale); rc != WAE_OK) return rc;
  const int grid = (int)((n + 255) / 256 > 8192 ? 8192 : (n + 255) / 256);
  if (dtype == WAE_BF16) hipLaunchKernelGGL(dropout_bwd_kernel<__bf16>, dim3(grid), dim3(256), 0, as_stream(stream), acc, g_next, out, n, dropout_key(seed), thr, scale, alpha);
  else if (dtype == WAE_F16) hipLaunchKernelGGL(dropout_bwd_kernel<f16>, dim3(grid), dim3(256), 0, as_stream(stream), acc, g_next, out, n, dropout_key(seed), thr, scale, alpha);
  else hipLaunchKernelGGL(dropout_bwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), acc, g_next, out, n, dropout_key(seed), thr, scale, alpha);
  return wae_check_launch("dropout_bwd");
}
