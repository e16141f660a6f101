// Microbenchmark (diagnostic only): the chunk loop of the hot-path kernels in isolation -- csrc/wae_common.hpp: gemm_chunk.
// Per chunk of N steps: prologue of D ds_read_b128 A-fragment reads; step i = {counted lgkmcnt wait for read i, one
// v_mfma_f32_32x32x16_bf16 on accumulator i % NM, read i + D into the register just consumed}; then lgkmcnt(0) and a workgroup
// barrier (cold restart, as after every weight chunk).  Reports the matrix-pipe utilisation against one MFMA per 32 clocks and SIMD.
//   usage: lds_loop <waves per workgroup 4|8> <depth 2|4|8> <steps per chunk 24|48> [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int I, int N, int D>
struct Step {
  static __device__ __forceinline__ void run(unsigned base, bf16x8 (&a)[D], const bf16x8& b, f32x16 (&acc)[6]) {
    constexpr int remaining = N - 1 - I, cnt = remaining < D - 1 ? remaining : D - 1;
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[I % D]) : "n"(cnt));
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[I % 6]) : "v"(a[I % D]), "v"(b));
    if constexpr (I + D < N) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[I % D]) : "v"(base), "n"(((I + D) % 48) * 1024));
    if constexpr (I + 1 < N) Step<I + 1, N, D>::run(base, a, b, acc);
  }
};
template <int I, int D>
struct Pro {
  static __device__ __forceinline__ void run(unsigned base, bf16x8 (&a)[D]) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[I]) : "v"(base), "n"(I * 1024));
    if constexpr (I + 1 < D) Pro<I + 1, D>::run(base, a);
  }
};

template <int N, int D>
__global__ void __launch_bounds__(512) k(float* out, int iters, long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 49152 / 4; i += blockDim.x) ((float*)smem)[i] = (float)i * 1e-6f;
  __syncthreads();
  const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem + lane * 16;
  f32x16 acc[6];
  for (int m = 0; m < 6; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 a[D];
  bf16x8 b = {};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    Pro<0, D>::run(base, a);
    Step<0, N, D>::run(base, a, b, acc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float sum = 0.f;
  for (int m = 0; m < 6; ++m) sum += acc[m][0] + acc[m][5];
  if (sum == 123.456f) out[0] = sum;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int N, int D>
static void run(int nw, int iters) {
  int ncu = 0;
  (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  float* out; long long* cyc;
  (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 8 * ncu);
  (void)hipFuncSetAttribute((const void*)k<N, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, D>), dim3(ncu), dim3(nw * 64), 98304, 0, out, iters, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  }
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  long long c0; (void)hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
  const double mf = (double)nw * iters * N / 4.0;    // MFMAs per SIMD
  printf("waves/wg %d  depth %d  steps/chunk %d: %.3f ms, clock %.2f GHz, matrix pipe %.1f %% busy, %.0f clocks per chunk (MFMAs alone: %d)\n", nw, D, N, ms,
         c0 / (ms * 1e6), 100.0 * mf * 32.0 / (double)c0, (double)c0 / iters, N * 32 * nw / 4);
}

int main(int argc, char** argv) {
  const int nw = argc > 1 ? atoi(argv[1]) : 4, d = argc > 2 ? atoi(argv[2]) : 4, n = argc > 3 ? atoi(argv[3]) : 48;
  const int iters = argc > 4 ? atoi(argv[4]) : 2000;
  if (n == 48 && d == 2) run<48, 2>(nw, iters);
  else if (n == 48 && d == 4) run<48, 4>(nw, iters);
  else if (n == 48 && d == 8) run<48, 8>(nw, iters);
  else if (n == 24 && d == 2) run<24, 2>(nw, iters);
  else if (n == 24 && d == 4) run<24, 4>(nw, iters);
  else if (n == 24 && d == 8) run<24, 8>(nw, iters);
  else printf("unsupported\n");
  return 0;
}
