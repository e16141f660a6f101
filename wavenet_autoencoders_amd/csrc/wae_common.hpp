// Shared device/host helpers for libwae_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/wae.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

void wae_set_error(const char* fmt, ...);
int wae_check_launch(const char* what);

// Dynamic-LDS opt-in (hipFuncAttributeMaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: the largest size
// already granted is remembered per device (one cache object per kernel instantiation), so a process that drives several
// GPUs -- or threads that launch concurrently -- never skips the call on a device that has not seen it.  Setting the
// attribute twice is harmless; the cache only saves the driver call.
#include <atomic>
#define WAE_MAX_DEVICES 64
struct WaeLdsCache {
  std::atomic<size_t> granted[WAE_MAX_DEVICES];
};
static inline int wae_ensure_lds(const void* kernel, WaeLdsCache& c, size_t lds, const char* what) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= WAE_MAX_DEVICES) dev = -1;
  if (dev >= 0 && c.granted[dev].load(std::memory_order_relaxed) >= lds) return WAE_OK;
  if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    wae_set_error("%s: cannot raise dynamic LDS to %zu bytes", what, lds);
    return WAE_EHIP;
  }
  if (dev >= 0) {
    size_t cur = c.granted[dev].load(std::memory_order_relaxed);
    while (cur < lds && !c.granted[dev].compare_exchange_weak(cur, lds, std::memory_order_relaxed)) {}
  }
  return WAE_OK;
}

#define WAE_REQUIRE(cond, ...)      \
  do {                              \
    if (!(cond)) {                  \
      wae_set_error(__VA_ARGS__);   \
      return WAE_EINVAL;            \
    }                               \
  } while (0)

// ---------------------------------------------------------------------------------------------------
// Element traits: one 16-byte MFMA operand fragment per lane = 8 bf16 (one 32x32x16 MFMA) or 4 f32
// (four exact-fp32 32x32x2 MFMAs).  A fragment block is 64 lanes x 16 B = 1 KiB in both cases, so the
// LDS images, chunk sizes and byte addressing of the two precisions are identical.
// ---------------------------------------------------------------------------------------------------
template <typename E>
struct ET;
template <>
struct ET<float> {
  using frag = f32x4;
  using vec4 = f32x4;  // 4 consecutive channels
  static constexpr int EPL = 4;   // elements per 16-B fragment
  static constexpr int CK = 32;   // channels per 128-B chunk row
  static constexpr int KBU = 4;   // 16-B k-blocks per 32-row accumulator tile used as next operand
  static constexpr int MT2 = 2;   // second-GEMM M-tiles per weight chunk (MT2*KBU == 8)
  static constexpr int DT = WAE_F32;
};
template <>
struct ET<__bf16> {
  using frag = bf16x8;
  using vec4 = bf16x4;
  static constexpr int EPL = 8;
  static constexpr int CK = 64;
  static constexpr int KBU = 2;
  static constexpr int MT2 = 4;
  static constexpr int DT = WAE_BF16;
};

// fp16 storage (WAE_F16, BASELINE config C5 "fp16 + MFMA"): same fragment geometry and MFMA rate as bf16
// (v_mfma_f32_32x32x16_f16), fp32 accumulate; 10 mantissa bits instead of 7, 5 exponent bits instead of 8 -- the backward
// pass therefore runs on loss-scaled gradients (engine.py: grad_scale).
template <>
struct ET<f16> {
  using frag = f16x8;
  using vec4 = f16x4;
  static constexpr int EPL = 8;
  static constexpr int CK = 64;
  static constexpr int KBU = 2;
  static constexpr int MT2 = 4;
  static constexpr int DT = WAE_F16;
};
static inline bool wae_dtype_ok(int dt) { return dt == WAE_F32 || dt == WAE_BF16 || dt == WAE_F16; }
static inline bool wae_is16(int dt) { return dt == WAE_BF16 || dt == WAE_F16; }

__device__ __forceinline__ void mma32(f32x16& acc, const f32x4& a, const f32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

__device__ __forceinline__ void mma32(f32x16& acc, const f16x8& a, const f16x8& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
}

__device__ __forceinline__ f32x4 to_f32x4(const f32x4& v) { return v; }
__device__ __forceinline__ f32x4 to_f32x4(const f16x4& v) {
  f32x4 r;
  r.x = (float)v.x; r.y = (float)v.y; r.z = (float)v.z; r.w = (float)v.w;
  return r;
}
__device__ __forceinline__ f32x4 to_f32x4(const bf16x4& v) {
  f32x4 r;
  r.x = (float)v.x; r.y = (float)v.y; r.z = (float)v.z; r.w = (float)v.w;
  return r;
}
template <typename E>
__device__ __forceinline__ typename ET<E>::vec4 from_f32x4(const f32x4& v);
template <>
__device__ __forceinline__ f32x4 from_f32x4<float>(const f32x4& v) { return v; }
template <>
__device__ __forceinline__ bf16x4 from_f32x4<__bf16>(const f32x4& v) {
  bf16x4 r;
  r.x = (__bf16)v.x; r.y = (__bf16)v.y; r.z = (__bf16)v.z; r.w = (__bf16)v.w;
  return r;
}

template <>
__device__ __forceinline__ f16x4 from_f32x4<f16>(const f32x4& v) {
  f16x4 r;
  r.x = (f16)v.x; r.y = (f16)v.y; r.z = (f16)v.z; r.w = (f16)v.w;
  return r;
}

// 16 accumulator registers of a 32x32 tile -> the KBU operand fragments of the NEXT MFMA that sums over
// the tile's row index.  bf16: k-step s takes registers 8s..8s+7 (row 16s+8(j>>2)+4h+(j&3)); f32: k-block
// g takes registers 4g..4g+3 (row 8g+4h+j).  The host packs the A operand in the matching k order.
__device__ __forceinline__ void acc_to_frags(const f32x16& u, bf16x8 (&f)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) f[s][j] = (__bf16)u[8 * s + j];
}
__device__ __forceinline__ void acc_to_frags(const f32x16& u, f16x8 (&f)[2]) {
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) f[s][j] = (f16)u[8 * s + j];
}
__device__ __forceinline__ void acc_to_frags(const f32x16& u, f32x4 (&f)[4]) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f[g].x = u[4 * g]; f[g].y = u[4 * g + 1]; f[g].z = u[4 * g + 2]; f[g].w = u[4 * g + 3];
  }
}

// Linear global -> LDS copy of `bytes` (a multiple of 4 KiB) by LDS-DMA: every wave instruction moves 1 KiB to a
// wave-uniform LDS base + lane*16 (cdna_hip_programming.md section 5, Caveat).  Each of the 4 waves copies one contiguous
// quarter, 1 KiB per instruction; the instruction offset field (which advances the global AND the LDS address) covers
// four pieces, so one M0 write and one address computation serve four DMAs.
template <int NWAVES = 4>
__device__ __forceinline__ void dma_chunk(const char* __restrict__ gsrc, char* lds_dst, int bytes, int wave, int lane) {
  const int per_wave = bytes / NWAVES;
  const char* g = gsrc + wave * per_wave + lane * 16;
  char* l = lds_dst + wave * per_wave;
  int off = 0;
  for (; off + 4096 <= per_wave; off += 4096) {
    const __attribute__((address_space(1))) void* gp = (const __attribute__((address_space(1))) void*)(g + off);
    __attribute__((address_space(3))) void* lp = (__attribute__((address_space(3))) void*)(l + off);
    __builtin_amdgcn_global_load_lds(gp, lp, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(gp, lp, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(gp, lp, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(gp, lp, 16, 3072, 0);
  }
  for (; off < per_wave; off += 1024)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off),
                                     (__attribute__((address_space(3))) void*)(l + off), 16, 0, 0);
}

// One weight chunk of N fragment blocks (LDS image [i][lane][16 B], i = blk*NM + m) against the B fragments
// of the chunk: acc[i % NM] += A_i . B[i / NM].  With one wave per SIMD nothing but the wave's own issue order
// hides LDS latency, and hipcc collapses a source-level prefetch back into read-wait-mfma.  So the A reads are
// inline-asm ds_read_b128 that run PD blocks ahead in a rotating register set, retired by counted
// s_waitcnt lgkmcnt(n) statements that carry the destination as "+v" (cdna_hip_programming.md 5.7 form ii):
// the MFMA consumes the wait's output, so it cannot be hoisted above it.  All reads are retired on exit.
template <int CNT>
__device__ __forceinline__ void lds_wait(f32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_wait(bf16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }
template <int CNT>
__device__ __forceinline__ void lds_wait(f16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(CNT)); }

template <int OFF, typename frag>
__device__ __forceinline__ void lds_read_async(frag& dst, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}

// KMAJOR = false: block I = blk*NM + m   -> acc[I % NM] += A_I . B[I / NM]   (first GEMM: k-blocks outer)
// KMAJOR = true : block I = mt*NB + kb   -> acc[I / NB] += A_I . B[I % NB]   (second GEMM: M-tiles outer)
// filler(std::integral_constant<int, I>) runs after MFMA I: callers spread their VMEM issue (LDS-DMA pieces, operand
// loads for later chunks) over the MFMAs of a chunk instead of issuing it as a burst that queues at the CU's one
// texture-address unit while the matrix pipe idles.
struct NoFiller {
  template <typename T>
  __device__ __forceinline__ void operator()(T) const {}
};
template <int V>
struct IntC { static constexpr int value = V; };

template <int I, int N, int NM, int PD, int NB, bool KMAJOR, typename frag, typename F>
struct GemmChunkStep {
  static __device__ __forceinline__ void run(unsigned addr, frag (&a)[PD], const frag (&B)[NB], f32x16 (&acc)[NM], F& filler) {
    constexpr int remaining = N - 1 - I;                      // reads issued after block I
    constexpr int cnt = remaining < PD - 1 ? remaining : PD - 1;
    lds_wait<cnt>(a[I % PD]);
    if constexpr (KMAJOR)
      mma32(acc[I / NB], a[I % PD], B[I % NB]);
    else
      mma32(acc[I % NM], a[I % PD], B[I / NM]);
    if constexpr (I + PD < N) lds_read_async<(I + PD) * 1024>(a[I % PD], addr);
    filler(IntC<I>{});
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < N) GemmChunkStep<I + 1, N, NM, PD, NB, KMAJOR, frag, F>::run(addr, a, B, acc, filler);
  }
};

template <int I, int PD, typename frag>
struct GemmChunkPrologue {
  static __device__ __forceinline__ void run(unsigned addr, frag (&a)[PD]) {
    lds_read_async<I * 1024>(a[I], addr);
    if constexpr (I + 1 < PD) GemmChunkPrologue<I + 1, PD, frag>::run(addr, a);
  }
};

template <int N, int NM, int NB, bool KMAJOR = false, int PDMAX = 8, typename frag, typename F>
__device__ __forceinline__ void gemm_chunk_fill(const char* buf, const frag (&B)[NB], f32x16 (&acc)[NM], F& filler) {
  constexpr int PD = N < PDMAX ? N : PDMAX;
  static_assert((N - 1) * 1024 < 65536, "ds_read offset field is 16 bits");
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)buf;
  frag a[PD];
  __builtin_amdgcn_sched_barrier(0);
  GemmChunkPrologue<0, PD, frag>::run(addr, a);
  GemmChunkStep<0, N, NM, PD, NB, KMAJOR, frag, F>::run(addr, a, B, acc, filler);
}
template <int N, int NM, int NB, bool KMAJOR = false, int PDMAX = 8, typename frag>
__device__ __forceinline__ void gemm_chunk(const char* buf, const frag (&B)[NB], f32x16 (&acc)[NM]) {
  NoFiller nf;
  gemm_chunk_fill<N, NM, NB, KMAJOR, PDMAX>(buf, B, acc, nf);
}

// The same chunk against CG column groups per wave (CG = 2: 64 time columns): every A fragment read from LDS feeds CG MFMAs,
// so the LDS read traffic per MFMA is 1/CG KiB.  (Round 1 priced the LDS port at 128 B/clk and expected CG = 2 to relieve it;
// tools/micro/lds_read.hip measures 256 B/clk, and one fragment per MFMA keeps the matrix pipe 98 % busy: CG = 2 buys nothing.)
// The filler runs after every MFMA (index I * CG + c).
template <int I, int N, int NM, int PD, int NB, int CG, bool KMAJOR, typename frag, typename F>
struct GemmChunkStepCG {
  static __device__ __forceinline__ void run(unsigned addr, frag (&a)[PD], const frag (&B)[CG][NB], f32x16 (&acc)[CG][NM], F& filler) {
    constexpr int remaining = N - 1 - I;
    constexpr int cnt = remaining < PD - 1 ? remaining : PD - 1;
    lds_wait<cnt>(a[I % PD]);
    if constexpr (KMAJOR)
      mma32(acc[0][I / NB], a[I % PD], B[0][I % NB]);
    else
      mma32(acc[0][I % NM], a[I % PD], B[0][I / NM]);
    filler(IntC<I * CG>{});
    if constexpr (CG == 2) {
      if constexpr (KMAJOR)
        mma32(acc[1][I / NB], a[I % PD], B[1][I % NB]);
      else
        mma32(acc[1][I % NM], a[I % PD], B[1][I / NM]);
      filler(IntC<I * CG + 1>{});
    }
    if constexpr (I + PD < N) lds_read_async<(I + PD) * 1024>(a[I % PD], addr);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (I + 1 < N) GemmChunkStepCG<I + 1, N, NM, PD, NB, CG, KMAJOR, frag, F>::run(addr, a, B, acc, filler);
  }
};
template <int N, int NM, int NB, int CG, bool KMAJOR = false, int PDMAX = 8, typename frag, typename F>
__device__ __forceinline__ void gemm_chunk_fill_cg(const char* buf, const frag (&B)[CG][NB], f32x16 (&acc)[CG][NM], F& filler) {
  constexpr int PD = N < PDMAX ? N : PDMAX;
  static_assert((N - 1) * 1024 < 65536, "ds_read offset field is 16 bits");
  static_assert(CG == 1 || CG == 2, "one or two column groups per wave");
  const unsigned addr = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)buf;
  frag a[PD];
  __builtin_amdgcn_sched_barrier(0);
  GemmChunkPrologue<0, PD, frag>::run(addr, a);
  GemmChunkStepCG<0, N, NM, PD, NB, CG, KMAJOR, frag, F>::run(addr, a, B, acc, filler);
}

// Operand fragments requested by inline asm and retired by a counted wait that carries the destinations ("+v": every use
// comes after it).  hipcc's own bookkeeping puts s_waitcnt vmcnt(0) in front of the first use of a loop-carried plain load,
// which drains the LDS-DMA requests of the NEXT chunks that were issued after it (vmcnt retires in order).
// The destination is a READ-WRITE operand ("+v"): the load lands in the register that held the fragment's previous value.
// With a write-only output hipcc may (a) give a conditionally executed request a scratch destination and copy it to the
// fragment's home register right after the join -- reading it before it has landed, after which the landing load overwrites
// whatever the scratch register holds by then (a wild pointer, in the first 64-column build of glu_fwd) -- or (b) treat a
// request whose result is overwritten before use as a dead definition and point it at registers that are in use.
// tools/check_asm_regs.py scans the built ISA for any instruction that touches a requested register before its counted wait.
template <int OFF, typename F>
__device__ __forceinline__ void gload_async(F& dst, const char* ptr) {
  static_assert(sizeof(F) == 16, "one 16-byte fragment");
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "+v"(dst) : "v"(ptr), "n"(OFF));
}
// N consecutive requests: fragment f of dst from ptr + 32 f
template <int I, int N, typename F>
__device__ __forceinline__ void gload_async_n(F (&dst)[N], const char* ptr) {
  gload_async<I * 32>(dst[I], ptr);
  if constexpr (I + 1 < N) gload_async_n<I + 1, N>(dst, ptr);
}
template <int CNT, typename F>
__device__ __forceinline__ void wait_vmcnt_frags(F (&a)[4]) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(CNT));
}

// one 1-KiB LDS-DMA piece: lane l copies 16 bytes from g (per-lane address) to lds_base (wave-uniform) + 16 l
__device__ __forceinline__ void dma_piece(const char* g, char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}

// 16 accumulator registers of a tile start from a per-row constant table: rows 8g + 4h + j, j < 4
__device__ __forceinline__ void init_rows(f32x16& acc, const float* tab /* 32 floats, 16-B aligned */, int h) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = *(const f32x4*)(tab + 8 * g + 4 * h);
    acc[4 * g + 0] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
  }
}

// ---------------------------------------------------------------------------------------------------
// stage_store_tiles: one wave stores NT accumulator tiles (32 time rows x 32 channels each; lane = time column
// n, register group g = channels 8g+4h..+3) as contiguous channel runs of its 32 time rows, through a
// wave-private 8 KiB LDS tile (row pitch 256 B, 16-byte chunks XOR-swizzled by the row: conflict-free row
// reads, 2-way column writes) so that every global store instruction writes full 16-byte-per-lane row segments.
// Passes of up to 256 bytes per row: 4/2/1 tiles (bf16) or 2/1 tiles (f32).
// ---------------------------------------------------------------------------------------------------
#define STG_BYTES 8192
// PITCH = bytes per staged row (256: 8 KiB tile; 128: 4 KiB tile for kernels that run two workgroups per CU)
template <typename EO, int NTP, int PITCH = 256>
__device__ __forceinline__ void stage_store_pass(char* stg, const f32x16* y, char* gout, int64_t row_stride, int rows_valid,
                                                 int lane) {
  using vec4 = typename ET<EO>::vec4;
  constexpr int SEG = NTP * 32 * sizeof(EO);
  static_assert(SEG <= PITCH && SEG >= 64, "a staging pass covers 64..PITCH bytes per row");
  constexpr int LPR = SEG / 16, RPI = 64 / LPR, NI = 32 / RPI;
  constexpr int KEY = PITCH / 16 - 1;
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < NTP; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int c16, sub;
      if constexpr (sizeof(EO) == 2) { c16 = 4 * mt + g; sub = 8 * h; } else { c16 = 8 * mt + 2 * g + h; sub = 0; }
      const f32x4 v = {y[mt][4 * g], y[mt][4 * g + 1], y[mt][4 * g + 2], y[mt][4 * g + 3]};
      *(vec4*)(stg + n * PITCH + ((c16 ^ (n & KEY)) << 4) + sub) = from_f32x4<EO>(v);
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int rr = lane / LPR, ck = lane % LPR;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = i * RPI + rr;
    const f32x4 v = *(const f32x4*)(stg + row * PITCH + ((ck ^ (row & KEY)) << 4));
    if (row < rows_valid) *(f32x4*)(gout + row * row_stride + ck * 16) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <typename EO, int NT, int PITCH = 256>
__device__ __forceinline__ void stage_store_tiles(char* stg, const f32x16* y, char* gout, int64_t row_stride, int rows_valid,
                                                  int lane) {
  constexpr int PT = PITCH / (32 * (int)sizeof(EO));  // tiles per full pass: 4 (bf16) / 2 (f32) at PITCH 256
  if constexpr (NT >= PT) {
    stage_store_pass<EO, PT, PITCH>(stg, y, gout, row_stride, rows_valid, lane);
    if constexpr (NT > PT) stage_store_tiles<EO, NT - PT, PITCH>(stg, y + PT, gout + PITCH, row_stride, rows_valid, lane);
  } else if constexpr (NT >= 2) {
    stage_store_pass<EO, 2, PITCH>(stg, y, gout, row_stride, rows_valid, lane);
    if constexpr (NT > 2) stage_store_tiles<EO, NT - 2, PITCH>(stg, y + 2, gout + 64 * sizeof(EO), row_stride, rows_valid, lane);
  } else {
    stage_store_pass<EO, 1, PITCH>(stg, y, gout, row_stride, rows_valid, lane);
  }
}

// stage_load_tiles: the inverse -- NT tiles (32 rows x 32 channels) of a time-major (rows, channels) array into the
// accumulator layout (lane = time column, register group g = channels 8g+4h..+3): coalesced 16-B row loads -> LDS ->
// per-lane column pieces.  Rows at or beyond rows_valid read row rows_valid-1 (callers never use them).
template <typename EO, int NTP, int PITCH = 256>
__device__ __forceinline__ void stage_load_pass(char* stg, f32x16* y, const char* gin, int64_t row_stride, int rows_valid,
                                                int lane) {
  using vec4 = typename ET<EO>::vec4;
  constexpr int SEG = NTP * 32 * sizeof(EO);
  static_assert(SEG <= PITCH && SEG >= 64, "a staging pass covers 64..PITCH bytes per row");
  constexpr int LPR = SEG / 16, RPI = 64 / LPR, NI = 32 / RPI;
  constexpr int KEY = PITCH / 16 - 1;
  const int n = lane & 31, h = lane >> 5;
  const int rr = lane / LPR, ck = lane % LPR;
  f32x4 tmp[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = min(i * RPI + rr, rows_valid - 1);
    tmp[i] = *(const f32x4*)(gin + row * row_stride + ck * 16);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = i * RPI + rr;
    *(f32x4*)(stg + row * PITCH + ((ck ^ (row & KEY)) << 4)) = tmp[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int mt = 0; mt < NTP; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int c16, sub;
      if constexpr (sizeof(EO) == 2) { c16 = 4 * mt + g; sub = 8 * h; } else { c16 = 8 * mt + 2 * g + h; sub = 0; }
      const f32x4 v = to_f32x4(*(const vec4*)(stg + n * PITCH + ((c16 ^ (n & KEY)) << 4) + sub));
      y[mt][4 * g] = v.x; y[mt][4 * g + 1] = v.y; y[mt][4 * g + 2] = v.z; y[mt][4 * g + 3] = v.w;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <typename EO, int NT, int PITCH = 256>
__device__ __forceinline__ void stage_load_tiles(char* stg, f32x16* y, const char* gin, int64_t row_stride, int rows_valid,
                                                 int lane) {
  constexpr int PT = PITCH / (32 * (int)sizeof(EO));
  if constexpr (NT >= PT) {
    stage_load_pass<EO, PT, PITCH>(stg, y, gin, row_stride, rows_valid, lane);
    if constexpr (NT > PT) stage_load_tiles<EO, NT - PT, PITCH>(stg, y + PT, gin + PITCH, row_stride, rows_valid, lane);
  } else if constexpr (NT >= 2) {
    stage_load_pass<EO, 2, PITCH>(stg, y, gin, row_stride, rows_valid, lane);
    if constexpr (NT > 2) stage_load_tiles<EO, NT - 2, PITCH>(stg, y + 2, gin + 64 * sizeof(EO), row_stride, rows_valid, lane);
  } else {
    stage_load_pass<EO, 1, PITCH>(stg, y, gin, row_stride, rows_valid, lane);
  }
}

// Split form of stage_load_tiles for kernels that know early which rows they will need at the end: stage_fetch_tiles
// issues the coalesced row loads into registers (call before the GEMM; the data arrives under the MFMAs) and
// stage_unpack_tiles turns them into accumulator-layout tiles through the LDS tile afterwards.  One f32x4[8] per pass.
template <int NT, typename EO>
struct StagePasses {
  static constexpr int PT = 256 / (32 * (int)sizeof(EO));
  static constexpr int N = (NT + PT - 1) / PT + (NT % PT == 3 ? 1 : 0);   // e.g. bf16: 6 -> {4,2}; 3 -> {2,1}
};
template <typename EO, int NTP>
__device__ __forceinline__ void stage_fetch_pass(f32x4 (&r)[8], const char* gin, int64_t row_stride, int rows_valid, int lane) {
  constexpr int SEG = NTP * 32 * sizeof(EO);
  constexpr int LPR = SEG / 16, RPI = 64 / LPR, NI = 32 / RPI;
  const int rr = lane / LPR, ck = lane % LPR;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = min(i * RPI + rr, rows_valid - 1);
    r[i] = *(const f32x4*)(gin + row * row_stride + ck * 16);
  }
}
template <typename EO, int NTP, int PITCH = 256>
__device__ __forceinline__ void stage_unpack_pass(char* stg, f32x16* y, const f32x4 (&r)[8], int lane) {
  using vec4 = typename ET<EO>::vec4;
  constexpr int SEG = NTP * 32 * sizeof(EO);
  static_assert(SEG <= PITCH, "a staging pass covers at most PITCH bytes per row");
  constexpr int LPR = SEG / 16, RPI = 64 / LPR, NI = 32 / RPI;
  constexpr int KEY = PITCH / 16 - 1;
  const int n = lane & 31, h = lane >> 5;
  const int rr = lane / LPR, ck = lane % LPR;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int row = i * RPI + rr;
    *(f32x4*)(stg + row * PITCH + ((ck ^ (row & KEY)) << 4)) = r[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int mt = 0; mt < NTP; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      int c16, sub;
      if constexpr (sizeof(EO) == 2) { c16 = 4 * mt + g; sub = 8 * h; } else { c16 = 8 * mt + 2 * g + h; sub = 0; }
      const f32x4 v = to_f32x4(*(const vec4*)(stg + n * PITCH + ((c16 ^ (n & KEY)) << 4) + sub));
      y[mt][4 * g] = v.x; y[mt][4 * g + 1] = v.y; y[mt][4 * g + 2] = v.z; y[mt][4 * g + 3] = v.w;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
template <typename EO, int NT, int PI = 0, int NP>
__device__ __forceinline__ void stage_fetch_tiles(f32x4 (&r)[NP][8], const char* gin, int64_t row_stride, int rows_valid, int lane) {
  constexpr int PT = 256 / (32 * (int)sizeof(EO));
  constexpr int NTP = NT >= PT ? PT : (NT >= 2 ? 2 : 1);
  static_assert(PI < NP, "pass array too small");
  stage_fetch_pass<EO, NTP>(r[PI], gin, row_stride, rows_valid, lane);
  if constexpr (NT > NTP) stage_fetch_tiles<EO, NT - NTP, PI + 1>(r, gin + NTP * 32 * sizeof(EO), row_stride, rows_valid, lane);
}
template <typename EO, int NT, int PI = 0, int NP>
__device__ __forceinline__ void stage_unpack_tiles(char* stg, f32x16* y, const f32x4 (&r)[NP][8], int lane) {
  constexpr int PT = 256 / (32 * (int)sizeof(EO));
  constexpr int NTP = NT >= PT ? PT : (NT >= 2 ? 2 : 1);
  stage_unpack_pass<EO, NTP>(stg, y, r[PI], lane);
  if constexpr (NT > NTP) stage_unpack_tiles<EO, NT - NTP, PI + 1>(stg, y + NTP, r, lane);
}

// Residual x[t] arrives as operand-shaped 16-byte fragments: fragment f of lane (n, h) = row bytes
// [32 f + 16 h, +16) of the chunk.  f32: that IS the accumulator layout (tile f/4, group f%4, channels 8g+4h+j).
// bf16: 8 channels 16 f + 8 h + j; the accumulator layout wants channels 8g + 4h' + j' -> exchange register pairs
// between the lane halves once (v_permlane32_swap), after which registers {0,1} hold group 2(f%2) and {2,3}
// group 2(f%2)+1 of tile f/2 for this lane's half.
template <int N>
__device__ __forceinline__ void residual_to_acc_layout(f32x4 (&)[N]) {}
template <typename V8, int N>
__device__ __forceinline__ void residual_swap16(V8 (&res)[N]) {
#pragma unroll
  for (int f = 0; f < N; ++f) {
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 r = __builtin_bit_cast(u32x4, res[f]);
    auto s0 = __builtin_amdgcn_permlane32_swap(r.x, r.z, false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(r.y, r.w, false, false);
    r.x = s0[0]; r.z = s0[1]; r.y = s1[0]; r.w = s1[1];
    res[f] = __builtin_bit_cast(V8, r);
  }
}
template <int N>
__device__ __forceinline__ void residual_to_acc_layout(bf16x8 (&res)[N]) { residual_swap16(res); }
template <int N>
__device__ __forceinline__ void residual_to_acc_layout(f16x8 (&res)[N]) { residual_swap16(res); }
template <typename E, int N>
__device__ __forceinline__ f32x4 residual_piece(const f32x4 (&res)[N], int mt, int g) { return res[4 * mt + g]; }
template <typename E, int N>
__device__ __forceinline__ f32x4 residual_piece(const bf16x8 (&res)[N], int mt, int g) {
  const bf16x8 v = res[2 * mt + (g >> 1)];
  const int o = 4 * (g & 1);
  f32x4 r = {(float)v[o], (float)v[o + 1], (float)v[o + 2], (float)v[o + 3]};
  return r;
}
template <typename E, int N>
__device__ __forceinline__ f32x4 residual_piece(const f16x8 (&res)[N], int mt, int g) {
  const f16x8 v = res[2 * mt + (g >> 1)];
  const int o = 4 * (g & 1);
  f32x4 r = {(float)v[o], (float)v[o + 1], (float)v[o + 2], (float)v[o + 3]};
  return r;
}

// XCD-aware tile order.  The dispatcher deals workgroups round-robin over the 8 XCDs (flat id mod 8) and every XCD has its
// own 4 MiB L2.  Time tiles that are neighbours -- they share the dilated taps x[t - d], x[t - 2d] and the halo rows --
// must therefore NOT get consecutive flat ids: XCD k takes the k-th contiguous eighth of the tile list instead, so the
// rows a tile's taps reach were (or are being) fetched into the same L2 by the tiles running next to it.  Bijective for any n.
__device__ __forceinline__ int xcd_contiguous_tile(int bid, int n) {
  const int k = bid & 7, q = n >> 3, r = n & 7;
  return k * q + min(k, r) + (bid >> 3);
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
