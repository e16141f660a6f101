// Microbenchmark (diagnostic only): what the LDS read port sustains for the A-fragment reads of the hot-path kernels, alone
// and interleaved with MFMAs.  Every wave reads 1-KiB fragments (ds_read_b128, lane l at base + 16 l: 16 consecutive lanes
// cover the 64 banks once) out of a 32-KiB chunk image, R reads per M MFMAs (v_mfma_f32_32x32x16_bf16), in a PD-deep
// rotation with counted lgkmcnt waits like csrc/wae_common.hpp: gemm_chunk.
//   usage: lds_read <waves per workgroup 4|8> <workgroups per CU 1|2> <reads> <mfmas> [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int R, int M>
__global__ void __launch_bounds__(512) k(float* out, int iters, long long* cyc, int chunked) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) ((float*)smem)[i] = (float)i * 1e-6f;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 a[8];
  bf16x8 b = {};
  for (int i = 0; i < 8; ++i) a[i] = b;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    // one "chunk": 32 steps; step s issues read s (if s % (32 / R') ...) -- R reads and M MFMAs per group of max(R, M) steps
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      if constexpr (R > 0) {
        if (s % (R >= M ? 1 : M / R) == 0 || R >= M) {
#pragma unroll
          for (int rr = 0; rr < (R >= M ? R / (M > 0 ? M : 1) : 1); ++rr)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[(s + rr) & 7]) : "v"(base), "n"(((s * 2 + rr) & 31) * 1024));
        }
      }
      if constexpr (M > 0) {
        if (R > 0) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s & 3]) : "v"(a[(s + 4) & 7]), "v"(b));
      }
    }
    if (chunked) {   // a chunk boundary as the hot-path kernels have it: reads drained, workgroup barrier, cold restart
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int m = 0; m < 4; ++m) sum += acc[m][0] + acc[m][5];
  for (int i = 0; i < 8; ++i) sum += (float)a[i][0];
  if (sum == 123.456f) out[0] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int R, int M>
static void run(int nw, int occ, int iters, int chunked) {
  int ncu = 0;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; long long* cyc;
  hipMalloc(&out, 4); hipMalloc(&cyc, 8 * ncu * occ);
  const size_t lds = occ == 1 ? 65536 : 65536;     // 64 KiB per workgroup: at most two per CU
  hipFuncSetAttribute((const void*)k<R, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<R, M>), dim3(ncu * occ), dim3(nw * 64), lds, 0, out, iters, cyc, chunked);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const int reads_per_it = R == 0 ? 0 : (R >= M ? 32 * (R / (M > 0 ? M : 1)) : 32 / (M / R));
  const int mfma_per_it = M > 0 ? 32 : 0;
  const double waves = (double)ncu * occ * nw;
  const double rbytes = waves * iters * reads_per_it * 1024.0, mf = waves * iters * mfma_per_it;
  // shader clock from the stamps of workgroup 0
  long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
  const double clk_ghz = c0 / (ms * 1e6);
  printf("%s waves/wg %d wg/CU %d reads:mfma %d:%d  %.3f ms  clock(s_memtime) %.2f GHz  LDS read %.1f B/clk/CU  MFMA util %.1f %% (of 1 per 32 clk per SIMD)\n", chunked ? "32-step chunks + barrier:" : "free-running:", nw, occ, R, M,
         ms, clk_ghz, rbytes / ncu / (double)c0, 100.0 * (mf / (ncu * 4.0)) * 32.0 / (double)c0);
}

int main(int argc, char** argv) {
  const int nw = argc > 1 ? atoi(argv[1]) : 4, occ = argc > 2 ? atoi(argv[2]) : 1;
  const int R = argc > 3 ? atoi(argv[3]) : 1, M = argc > 4 ? atoi(argv[4]) : 1, iters = argc > 5 ? atoi(argv[5]) : 2000, chunked = argc > 6 ? atoi(argv[6]) : 0;
  if (R == 1 && M == 0) run<1, 0>(nw, occ, iters, chunked);
  else if (R == 0 && M == 1) run<0, 1>(nw, occ, iters, chunked);
  else if (R == 1 && M == 1) run<1, 1>(nw, occ, iters, chunked);
  else if (R == 1 && M == 2) run<1, 2>(nw, occ, iters, chunked);
  else if (R == 1 && M == 4) run<1, 4>(nw, occ, iters, chunked);
  else if (R == 2 && M == 1) run<2, 1>(nw, occ, iters, chunked);
  else printf("unsupported ratio\n");
  return 0;
}
