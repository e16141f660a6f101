// Microbenchmark: what limits the one-wave-per-SIMD MFMA + LDS-read loop of glu_fwd (diagnostic only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../wavenet_autoencoders_amd/csrc/wae_common.hpp"
void wae_set_error(const char*, ...) {}
int wae_check_launch(const char*) { return 0; }

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(const char* w, float* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NM = 12;
  constexpr int CHB = NM * 4 * 1024;
  for (int i = threadIdx.x; i < 2 * CHB / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  f32x16 acc[NM];
  for (int m = 0; m < NM; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 B[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) B[i][j] = (__bf16)(0.01f * (lane + i + j));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int q = 0; q < iters; ++q) {
    if (MODE >= 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (MODE >= 3) dma_chunk(w + (size_t)((q + 1) % 8) * CHB, smem + ((q + 1) & 1) * CHB, CHB, wave, lane);
    const char* buf = smem + (q & 1) * CHB + lane * 16;
    if (MODE == 0) {
      bf16x8 a = B[1];
#pragma unroll
      for (int i = 0; i < 4 * NM; ++i) {
        mma32(acc[i % NM], a, B[i / NM]);
      }
    } else {
      gemm_chunk<4 * NM, NM, 4>(buf, B, acc);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int m = 0; m < NM; ++m)
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const char* w, float* out, unsigned long long* cyc, int nwg) {
  const int iters = 26;
  const size_t lds = 2 * 12 * 4 * 1024;
  hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), lds, 0, w, out, cyc, iters);
  hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(256), lds, 0, w, out, cyc, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[1024];
  hipMemcpy(h, cyc, nwg * 8, hipMemcpyDeviceToHost);
  unsigned long long mn = ~0ull, mx = 0;
  for (int i = 0; i < nwg; ++i) { if (h[i] < mn) mn = h[i]; if (h[i] > mx) mx = h[i]; }
  const double nm = iters * 48.0;
  printf("%-28s nwg=%4d  %.1f us/launch  cycles/MFMA min %.1f max %.1f\n", name, nwg, ms * 100.0, mn / nm, mx / nm);
}

int main() {
  char* w; float* out; unsigned long long* cyc;
  hipMalloc(&w, 16 << 20);
  hipMemset(w, 0x11, 16 << 20);
  hipMalloc(&out, 1024 * 256 * 4);
  hipMalloc(&cyc, 1024 * 8);
  for (int nwg : {1, 256, 512}) {
    run<0>("mfma only (regs)", w, out, cyc, nwg);
    run<1>("mfma + pipelined ds_read", w, out, cyc, nwg);
    run<2>("  + barrier per chunk", w, out, cyc, nwg);
    run<3>("  + LDS-DMA 48KB per chunk", w, out, cyc, nwg);
  }
  return 0;
}
