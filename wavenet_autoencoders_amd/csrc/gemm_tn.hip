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

// One 128x128 output tile of one contraction; a launch processes an array of them (several weight gradients of a
// layer, or all the one-off ones, share one launch so that the chip is filled without splitting k finely: fp32
// atomics cap at ~1.3 TB/s chip-wide, so each workgroup must contract a long k-range per byte it adds to C).
struct TnTile {
  const char* P;          // time-major (B,T,p_stride), already offset to the tile's first column
  const char* Q;          // time-major (B,T,q_stride), already offset to the tile's first column
  const int32_t* onehot;  // if set: P[t][m] = (onehot[b*T+t] == m0 + m)
  float* C;               // top-left of the tile in the fp32 output
  int64_t p_stride, q_stride, ldc;
  int m_valid, n_valid;   // valid columns of P / Q inside this tile (<= 128)
  int m0;                 // first P column of the tile (one-hot compare only)
  int shift;              // Q row = t + shift
  int ones_col;           // tile-local index (< 128) of the virtual all-ones Q column, or -1
  float alpha;
};
struct TnArgs {
  const TnTile* tiles;
  int B, T;
  int tchunk;  // time steps per workgroup
  int dbg;     // timing-only ablation bits (tools/ablate_tn.py): 1 no global loads, 2 no LDS reads/MFMA, 4 no atomics
};
#ifdef WAE_DEBUG_KNOBS   // tools/ablate_tn.py only (make EXTRA=-DWAE_DEBUG_KNOBS)
static int g_tn_dbg = 0;
extern "C" void wae_debug_set_tn(int bits) { g_tn_dbg = bits; }
#else
static constexpr int g_tn_dbg = 0;
#endif

#define TN_KT 32      // time rows per LDS slab
#define TN_PITCH_BF16 320   // bytes per slab row (128 bf16 + pad): conflict-free transposed reads
#define TN_PITCH_F32 528    // 128 fp32 + 16 B pad

// Operand fragments of one k-step for a wave's 64x64 sub-tile: 2 A (P columns wm, wm+32) and 2 B (Q columns wn, wn+32).
// bf16: 8 consecutive k (time rows k0 + 8h + 0..7) of column cbase + (lane & 31): two ds_read_b64_tr_b16 each; all eight
// reads are issued back to back and retired by ONE wait that carries every destination (cdna_hip_programming.md 5.7 ii).
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
template <typename V8>   // bf16x8 or f16x8: the transposed read moves 16-bit elements whatever their format
__device__ __forceinline__ void tn_load_frags16(const char* sp, const char* sq, int k0, int wm, int wn, int lane, V8 (&a)[2],
                                                V8 (&b)[2]) {
  const int grp = (lane >> 4) & 1, h = lane >> 5, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned rowoff = (k0 + 8 * h + q) * TN_PITCH_BF16 + (16 * grp + 4 * pp) * 2;
  const unsigned ap = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)sp + rowoff + wm * 2;
  const unsigned bp = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)sq + rowoff + wn * 2;
  u32x2 r[8];
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[0]) : "v"(ap));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[1]) : "v"(ap), "n"(4 * TN_PITCH_BF16));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:64" : "=v"(r[2]) : "v"(ap));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[3]) : "v"(ap), "n"(4 * TN_PITCH_BF16 + 64));
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r[4]) : "v"(bp));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[5]) : "v"(bp), "n"(4 * TN_PITCH_BF16));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:64" : "=v"(r[6]) : "v"(bp));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r[7]) : "v"(bp), "n"(4 * TN_PITCH_BF16 + 64));
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]));
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    u32x4 va = {r[2 * i].x, r[2 * i].y, r[2 * i + 1].x, r[2 * i + 1].y};
    u32x4 vb = {r[4 + 2 * i].x, r[4 + 2 * i].y, r[4 + 2 * i + 1].x, r[4 + 2 * i + 1].y};
    a[i] = __builtin_bit_cast(V8, va);
    b[i] = __builtin_bit_cast(V8, vb);
  }
}
__device__ __forceinline__ void tn_load_frags(const char* sp, const char* sq, int k0, int wm, int wn, int lane, bf16x8 (&a)[2],
                                              bf16x8 (&b)[2]) {
  tn_load_frags16(sp, sq, k0, wm, wn, lane, a, b);
}
__device__ __forceinline__ void tn_load_frags(const char* sp, const char* sq, int k0, int wm, int wn, int lane, f16x8 (&a)[2],
                                              f16x8 (&b)[2]) {
  tn_load_frags16(sp, sq, k0, wm, wn, lane, a, b);
}
// fp32: fragment = 4 k-pairs; element j pairs time rows (k0 + 2j + h) -> one ds_read_b32 each (compiler scheduled)
__device__ __forceinline__ f32x4 tn_frag_f32(const char* slab, int k0, int cbase, int lane) {
  const int i = lane & 31, h = lane >> 5;
  const float* base = (const float*)(slab + (k0 + h) * TN_PITCH_F32) + cbase + i;
  f32x4 f;
  f.x = base[0];
  f.y = base[2 * TN_PITCH_F32 / 4];
  f.z = base[4 * TN_PITCH_F32 / 4];
  f.w = base[6 * TN_PITCH_F32 / 4];
  return f;
}
__device__ __forceinline__ void tn_load_frags(const char* sp, const char* sq, int k0, int wm, int wn, int lane, f32x4 (&a)[2],
                                              f32x4 (&b)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    a[i] = tn_frag_f32(sp, k0, wm + 32 * i, lane);
    b[i] = tn_frag_f32(sq, k0, wn + 32 * i, lane);
  }
}

template <typename E>
__global__ void __launch_bounds__(256, sizeof(E) == 2 ? 2 : 1) gemm_tn_kernel(TnArgs p) {
  using frag = typename ET<E>::frag;
  constexpr int ES = sizeof(E);
  constexpr int PITCH = ES == 2 ? TN_PITCH_BF16 : TN_PITCH_F32;
  constexpr int KSTEP = ES == 2 ? 16 : 8;  // time rows consumed by one fragment pair
  constexpr int SLAB = TN_KT * PITCH;
  constexpr int CPR = 128 * ES / 16;       // 16-byte pieces per slab row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* slabP = smem;            // [2][SLAB]
  char* slabQ = smem + 2 * SLAB; // [2][SLAB]
  // XCD-aware mapping (speed only): workgroups are dealt round-robin over the 8 XCDs in launch order, so the tiles that
  // contract the SAME k-range (they share their P and Q slabs) get flat ids that are congruent mod 8 and meet in one L2.
  int tile_i = blockIdx.x, kr = blockIdx.y;
  if ((gridDim.y & 7) == 0) {
    const int flat = blockIdx.y * gridDim.x + blockIdx.x;
    const int j = flat >> 3;
    tile_i = j % gridDim.x;
    kr = (flat & 7) + 8 * (j / gridDim.x);
  }
  const TnTile tl = p.tiles[tile_i];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int splits = (p.T + p.tchunk - 1) / p.tchunk;
  const int b = kr / splits;
  const int tbeg = (kr % splits) * p.tchunk;
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
  const int colp = lpc * EP, colq = lpc * EP;
  // Register prefetch ring: the loads of slab s + PF are in flight while slab s is contracted.  A slab is only 8 MFMAs
  // per wave (256 cycles), far less than one HBM/L2 round trip, so several slabs must be outstanding per workgroup.
  constexpr int PF = 4;
  f32x4 rp[PF][NPASS], rq[PF][NPASS];
  auto fetch = [&](int t0, f32x4 (&xp)[NPASS], f32x4 (&xq)[NPASS]) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int t = t0 + i * RPP + lrow;
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      xp[i] = z;
      xq[i] = z;
      if (t < tend) {
        if (tl.onehot) {
          const int id = tl.onehot[(int64_t)b * p.T + t] - tl.m0;
          if (id >= colp && id < colp + EP) {
            if constexpr (ES == 2) {
              typename ET<E>::frag oh = {};
              oh[id - colp] = (E)1.0f;
              xp[i] = __builtin_bit_cast(f32x4, oh);
            } else {
              float* o4 = (float*)&xp[i];
              o4[id - colp] = 1.0f;
            }
          }
        } else if (colp < tl.m_valid) {
          xp[i] = *(const f32x4*)(tl.P + (((int64_t)b * p.T + t) * tl.p_stride + colp) * ES);
        }
        const int tq = t + tl.shift;
        if (colq < tl.n_valid && tq >= 0 && tq < p.T) xq[i] = *(const f32x4*)(tl.Q + (((int64_t)b * p.T + tq) * tl.q_stride + colq) * ES);
        if (tl.ones_col >= colq && tl.ones_col < colq + EP) {   // virtual all-ones column
          if constexpr (ES == 2) {
            typename ET<E>::frag v = __builtin_bit_cast(typename ET<E>::frag, xq[i]);
            v[tl.ones_col - colq] = (E)1.0f;
            xq[i] = __builtin_bit_cast(f32x4, v);
          } else {
            float* v = (float*)&xq[i];
            v[tl.ones_col - colq] = 1.0f;
          }
        }
      }
    }
  };
  auto stash = [&](int buf, const f32x4 (&xp)[NPASS], const f32x4 (&xq)[NPASS]) {
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int row = i * RPP + lrow;
      *(f32x4*)(slabP + buf * SLAB + row * PITCH + lpc * 16) = xp[i];
      *(f32x4*)(slabQ + buf * SLAB + row * PITCH + lpc * 16) = xq[i];
    }
  };

  const int nslab = (tend - tbeg + TN_KT - 1) / TN_KT;
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < nslab && !(p.dbg & 1)) fetch(tbeg + j * TN_KT, rp[j], rq[j]);
  for (int s0 = 0; s0 < nslab; s0 += PF) {
#pragma unroll
    for (int j = 0; j < PF; ++j) {
      const int s = s0 + j;
      if (s < nslab) {                                   // workgroup-uniform
        stash(s & 1, rp[j], rq[j]);
        if (s + PF < nslab && !(p.dbg & 1)) fetch(tbeg + (s + PF) * TN_KT, rp[j], rq[j]);
        __syncthreads();                                 // slab s visible; every wave is past its reads of slab s-1
        if (p.dbg & 2) continue;
        const char* sp = slabP + (s & 1) * SLAB;
        const char* sq = slabQ + (s & 1) * SLAB;
#pragma unroll
        for (int k0 = 0; k0 < TN_KT; k0 += KSTEP) {
          frag a[2], bq[2];
          tn_load_frags(sp, sq, k0, wm, wn, lane, a, bq);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) mma32(acc[i][jj], a[i], bq[jj]);
        }
      }
    }
  }

  // C += alpha * acc   (lane = column n, registers = rows m)
  if ((p.dbg & 4) && acc[0][0][0] != 12345.f) return;
  const int nl = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      int col = wn + 32 * j + nl;
      const bool ones = tl.ones_col >= 0 && col == tl.ones_col;
      if (!(col < tl.n_valid || ones)) continue;
      if (ones) col += b;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (row < tl.m_valid) atomicAdd(tl.C + (int64_t)row * tl.ldc + col, tl.alpha * acc[i][j][r]);
      }
    }
}

extern "C" int wae_gemm_tn_tiles(int32_t dtype, const wae_tn_tile* tiles_dev, int32_t ntiles, int32_t B, int32_t T,
                                 int32_t splits, void* stream) {
  WAE_REQUIRE(tiles_dev && ntiles > 0 && B > 0 && T > 0 && splits >= 1, "gemm_tn_tiles: bad arguments");
  WAE_REQUIRE(wae_dtype_ok(dtype), "gemm_tn_tiles: bad dtype");
  static_assert(sizeof(wae_tn_tile) == sizeof(TnTile), "wae_tn_tile and TnTile must have the same layout");
  TnArgs a;
  a.tiles = (const TnTile*)tiles_dev;
  a.B = B; a.T = T; a.dbg = g_tn_dbg;
  int tchunk = (T + splits - 1) / splits;
  tchunk = (tchunk + TN_KT - 1) / TN_KT * TN_KT;
  a.tchunk = tchunk;
  const int nsp = (T + tchunk - 1) / tchunk;
  const int pitch = wae_is16(dtype) ? TN_PITCH_BF16 : TN_PITCH_F32;
  const size_t lds = (size_t)4 * TN_KT * pitch;
  hipStream_t st = as_stream(stream);
  dim3 grid(ntiles, B * nsp);
  if (dtype == WAE_BF16) {
    hipLaunchKernelGGL(gemm_tn_kernel<__bf16>, grid, dim3(256), lds, st, a);
  } else if (dtype == WAE_F16) {
    hipLaunchKernelGGL(gemm_tn_kernel<f16>, grid, dim3(256), lds, st, a);
  } else {
    (void)hipFuncSetAttribute((const void*)gemm_tn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(gemm_tn_kernel<float>, grid, dim3(256), lds, st, a);
  }
  return wae_check_launch("gemm_tn_tiles");
}
