// wae_gemm_tn: C[m][n] += alpha * sum_{clip b, t} P[b,t][m] * Q[b, t + shift][n]   (weight gradients)
//
// Every weight gradient of the path is a contraction over time of two time-major activations (autograd of the 1x1 and
// dilated convolutions: modules.py:134-160, wavenet.py:203,211-212):
//   dW1[g][(tap,r)] = sum dz[t][g] x[t-(k-1-tap)d][r]      dWc[g][c] = sum dz[t][g] c[t][c]
//   dW_out[r][h] = sum dx-hat[t][r] u[t][h]                 dW_skip_l[s][h] = sum dskip[t][s] u_l[t][h]  (all l at once)
//   dW_first[r][class] = sum dx0[t][r] onehot(id[t])[class]   (the one-hot operand is generated on the fly)
// plus the bias / per-clip conditioning-bias gradients as a virtual all-ones Q column ("ones_col"; clip b adds to
// column ones_col + b).
// MFMA view: A[m][k=t] = P^T and B[k=t][n] = Q both have the contraction index along the ROWS of the stored arrays, so
// a (32 rows x 128 channels) slab of each is staged in LDS with coalesced row loads and read back transposed
// (bf16: ds_read_b64_tr_b16, cdna_hip_programming.md T10; fp32: ds_read_b32 columns).  One workgroup = a 128x128 tile
// of C over one k-range of one clip; 4 waves, each a 64x64 sub-tile (2x2 MFMA tiles); results leave as fp32 atomics
// (rows of 32 consecutive n per half-wave = full-rate shape, MI355X_MICROARCH.md "Global float atomics").
#include "wae_common.hpp"

struct TnArgs {
  const char* P;
  const char* Q;
  const int32_t* onehot_idx;  // if set, P[t][m] = (idx[b*T+t] == m) and P is ignored
  float* C;
  int64_t p_stride, q_stride;  // elements per row
  int64_t ldc;
  int B, T, M, N;  // valid columns of P and Q
  int shift;       // Q row = t + shift
  int ones_col;    // < 0: off
  int tchunk;      // time steps per workgroup
  int ntiles;      // tiles along N
  float alpha;
};

#define TN_KT 32      // time rows per LDS slab
#define TN_PITCH_BF16 320   // bytes per slab row (128 bf16 + pad): conflict-free transposed reads
#define TN_PITCH_F32 528    // 128 fp32 + 16 B pad

template <typename E>
__device__ __forceinline__ void tn_load_frag(const char* slab, int k0, int cbase, int lane, typename ET<E>::frag& f);

// bf16: 8 consecutive k (time rows k0 + 8h + 0..7) of column cbase + (lane & 31)
template <>
__device__ __forceinline__ void tn_load_frag<__bf16>(const char* slab, int k0, int cbase, int lane, bf16x8& f) {
  const int grp = (lane >> 4) & 1, h = lane >> 5, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)slab +
                        (k0 + 8 * h + q) * TN_PITCH_BF16 + (cbase + 16 * grp + 4 * pp) * 2;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
  u32x2 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi)
               : "v"(addr), "n"(4 * TN_PITCH_BF16));
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  u32x4 r = {lo.x, lo.y, hi.x, hi.y};
  f = __builtin_bit_cast(bf16x8, r);
}
// fp32: fragment = 4 k-pairs; element j pairs time rows (k0 + 2j + h) -> one ds_read_b32 each
template <>
__device__ __forceinline__ void tn_load_frag<float>(const char* slab, int k0, int cbase, int lane, f32x4& f) {
  const int i = lane & 31, h = lane >> 5;
  const float* base = (const float*)(slab + (k0 + h) * TN_PITCH_F32) + cbase + i;
  f.x = base[0];
  f.y = base[2 * TN_PITCH_F32 / 4];
  f.z = base[4 * TN_PITCH_F32 / 4];
  f.w = base[6 * TN_PITCH_F32 / 4];
}

template <typename E>
__global__ void __launch_bounds__(256, 2) gemm_tn_kernel(TnArgs p) {
  using frag = typename ET<E>::frag;
  constexpr int ES = sizeof(E);
  constexpr int PITCH = ES == 2 ? TN_PITCH_BF16 : TN_PITCH_F32;
  constexpr int KSTEP = ES == 2 ? 16 : 8;  // time rows consumed by one fragment pair
  constexpr int SLAB = TN_KT * PITCH;
  constexpr int CPR = 128 * ES / 16;       // 16-byte pieces per slab row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slabP = smem;            // [2][SLAB]
  char* slabQ = smem + 2 * SLAB; // [2][SLAB]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int mt = blockIdx.x / p.ntiles, nt = blockIdx.x % p.ntiles;
  const int m0 = mt * 128, n0 = nt * 128;
  const int splits = (p.T + p.tchunk - 1) / p.tchunk;
  const int b = blockIdx.y / splits;
  const int tbeg = (blockIdx.y % splits) * p.tchunk;
  const int tend = min(p.T, tbeg + p.tchunk);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][c][r] = 0.f;

  // slab loaders: thread -> (row = i*rows_per_pass + tid / CPR, piece = tid % CPR)
  constexpr int RPP = 256 / CPR;      // rows per pass
  constexpr int NPASS = TN_KT / RPP;
  const int lrow = threadIdx.x / CPR, lpc = threadIdx.x % CPR;
  constexpr int EP = 16 / ES;         // elements per piece
  f32x4 rp[NPASS], rq[NPASS];
  auto fetch = [&](int t0) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int t = t0 + i * RPP + lrow;
      const int colp = m0 + lpc * EP, colq = n0 + lpc * EP;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      rp[i] = z;
      rq[i] = z;
      if (t < tend) {
        if (p.onehot_idx) {
          const int id = p.onehot_idx[(int64_t)b * p.T + t];
          if (id >= colp && id < colp + EP) {
            if constexpr (ES == 2) {
              bf16x8 oh = {};
              oh[id - colp] = (__bf16)1.0f;
              rp[i] = __builtin_bit_cast(f32x4, oh);
            } else {
              float o4[4] = {0.f, 0.f, 0.f, 0.f};
              o4[id - colp] = 1.0f;
              rp[i].x = o4[0]; rp[i].y = o4[1]; rp[i].z = o4[2]; rp[i].w = o4[3];
            }
          }
        } else if (colp < p.M) {
          rp[i] = *(const f32x4*)(p.P + (((int64_t)b * p.T + t) * p.p_stride + colp) * ES);
        }
        const int tq = t + p.shift;
        if (colq < p.N && tq >= 0 && tq < p.T) rq[i] = *(const f32x4*)(p.Q + (((int64_t)b * p.T + tq) * p.q_stride + colq) * ES);
        if (p.ones_col >= colq && p.ones_col < colq + EP) {   // virtual all-ones column
          if constexpr (ES == 2) {
            bf16x8 v = __builtin_bit_cast(bf16x8, rq[i]);
            v[p.ones_col - colq] = (__bf16)1.0f;
            rq[i] = __builtin_bit_cast(f32x4, v);
          } else {
            float* v = (float*)&rq[i];
            v[p.ones_col - colq] = 1.0f;
          }
        }
      }
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = i * RPP + lrow;
      *(f32x4*)(slabP + buf * SLAB + row * PITCH + lpc * 16) = rp[i];
      *(f32x4*)(slabQ + buf * SLAB + row * PITCH + lpc * 16) = rq[i];
    }
  };

  const int nslab = (tend - tbeg + TN_KT - 1) / TN_KT;
  if (nslab > 0) {
    fetch(tbeg);
    stash(0);
  }
  __syncthreads();
  for (int s = 0; s < nslab; ++s) {
    if (s + 1 < nslab) fetch(tbeg + (s + 1) * TN_KT);
    const char* sp = slabP + (s & 1) * SLAB;
    const char* sq = slabQ + (s & 1) * SLAB;
#pragma unroll
    for (int k0 = 0; k0 < TN_KT; k0 += KSTEP) {
      frag a[2], bq[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        tn_load_frag<E>(sp, k0, wm + 32 * i, lane, a[i]);
        tn_load_frag<E>(sq, k0, wn + 32 * i, lane, bq[i]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) mma32(acc[i][j], a[i], bq[j]);
    }
    if (s + 1 < nslab) stash((s + 1) & 1);
    __syncthreads();
  }

  // C += alpha * acc   (lane = column n, registers = rows m)
  const int nl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int col = n0 + wn + 32 * j + nl;
      const bool ones = p.ones_col >= 0 && col == p.ones_col;
      if (!(col < p.N || ones)) continue;
      if (ones) col += b;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        const int rows_ok = p.onehot_idx ? p.M : p.M;
        if (row < rows_ok) atomicAdd(p.C + (int64_t)row * p.ldc + col, p.alpha * acc[i][j][r]);
      }
    }
}

extern "C" int wae_gemm_tn(const wae_tn_desc* d, const void* P, int64_t p_stride, const int32_t* onehot_idx, const void* Q,
                           int64_t q_stride, float* C, int64_t ldc, void* stream) {
  WAE_REQUIRE(d && Q && C && (P || onehot_idx), "gemm_tn: null pointer argument");
  WAE_REQUIRE(d->dtype == WAE_F32 || d->dtype == WAE_BF16, "gemm_tn: bad dtype");
  WAE_REQUIRE(d->B > 0 && d->T > 0 && d->M > 0 && d->N > 0, "gemm_tn: bad sizes");
  const int ep = d->dtype == WAE_BF16 ? 8 : 4;
  WAE_REQUIRE(d->M % ep == 0 && d->N % ep == 0, "gemm_tn: M and N must be multiples of %d", ep);
  WAE_REQUIRE(d->ones_col < 0 || (d->ones_col >= d->N && d->ones_col + d->B <= ldc), "gemm_tn: ones_col must lie in [N, ldc-B]");
  TnArgs a;
  a.P = (const char*)P; a.Q = (const char*)Q; a.onehot_idx = onehot_idx; a.C = C; a.p_stride = p_stride; a.q_stride = q_stride;
  a.ldc = ldc; a.B = d->B; a.T = d->T; a.M = d->M; a.N = d->N; a.shift = d->shift; a.ones_col = d->ones_col; a.alpha = d->alpha;
  const int nmax = d->ones_col >= 0 ? d->ones_col + 1 : d->N;
  const int mtiles = (d->M + 127) / 128, ntiles = (nmax + 127) / 128;
  a.ntiles = ntiles;
  // k-split: enough workgroups to fill 256 CUs twice, at least 256 time steps each
  int splits = (2 * 256 + mtiles * ntiles * d->B - 1) / (mtiles * ntiles * d->B);
  if (splits < 1) splits = 1;
  int tchunk = (d->T + splits - 1) / splits;
  tchunk = (tchunk + TN_KT - 1) / TN_KT * TN_KT;
  if (tchunk < 256) tchunk = 256;
  a.tchunk = tchunk;
  splits = (d->T + tchunk - 1) / tchunk;
  const int pitch = d->dtype == WAE_BF16 ? TN_PITCH_BF16 : TN_PITCH_F32;
  const size_t lds = (size_t)4 * TN_KT * pitch;
  hipStream_t st = as_stream(stream);
  dim3 grid(mtiles * ntiles, d->B * splits);
  if (d->dtype == WAE_BF16) {
    hipLaunchKernelGGL(gemm_tn_kernel<__bf16>, grid, dim3(256), lds, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(256), lds, st, a);
  }
  return wae_check_launch("gemm_tn");
}
