// Microbenchmark: LDS-DMA weight streaming as the hot-path kernels do it (diagnostic only).
// Every workgroup streams the SAME `wbytes` of weights (L2-hot after the first pass), CH bytes per chunk through an
// NS-slot ring with a barrier per chunk, optionally consumes each chunk with ds_read_b128 + MFMA, optionally loads
// private operand fragments (4 x 16 B per lane and chunk) from a per-workgroup region.
//   usage: dma_stream <mode> <waves 4|8> <NS> <chunk KB> <MFMAs per wave and chunk> <wg per CU>
//   mode bits: 1 = DMA, 2 = ds_read + MFMA, 4 = operand loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../wavenet_autoencoders_amd/csrc/wae_common.hpp"
void wae_set_error(const char*, ...) {}
int wae_check_launch(const char*) { return 0; }

__device__ __forceinline__ void wait_vm(int w) {
  switch (w) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 20: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    case 24: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    case 30: asm volatile("s_waitcnt vmcnt(30)" ::: "memory"); break;
    case 40: asm volatile("s_waitcnt vmcnt(40)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

template <int NW>
__global__ void __launch_bounds__(NW * 64) k(const char* w, const char* x, float* out, int mode, int ns, int ch, int nchunks,
                                             int wchunks, int nmfma) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int per_wave = ch / NW, ppw = per_wave / 1024;
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 B[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) B[i][j] = (__bf16)(0.01f * (lane + i + j));
  const char* xw = x + ((size_t)blockIdx.x * NW + wave) * 32 * 512 + (lane & 31) * 512 + (lane >> 5) * 16;
  auto issue = [&](int q) {
    if (mode & 1) {
      const char* src = w + (size_t)(q % wchunks) * ch + wave * per_wave + lane * 16;
      char* dst = smem + (q % ns) * ch + wave * per_wave;
      for (int i = 0; i < ppw; ++i) dma_piece(src + i * 1024, dst + i * 1024);
    }
  };
  const int D = ns - 1;
  for (int q = 0; q < D && q < nchunks; ++q) issue(q);
  bf16x8 Bn[4] = {B[0], B[1], B[2], B[3]};
  for (int q = 0; q < nchunks; ++q) {
    const int younger = min(D - 1, nchunks - 1 - q);
    wait_vm((mode & 1) ? younger * ppw + ((mode & 4) ? 0 : 0) : 0);
    __builtin_amdgcn_s_barrier();
    if (q + D < nchunks) issue(q + D);
    if (mode & 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i) B[i] = Bn[i];
#pragma unroll
      for (int i = 0; i < 4; ++i) Bn[i] = *(const bf16x8*)(xw + ((q + 1) & 3) * 128 + i * 32);
    }
    if (mode & 2) {
      const char* buf = smem + (q % ns) * ch + lane * 16;
      for (int i = 0; i < nmfma; ++i) {
        const bf16x8 a = *(const bf16x8*)(buf + (i * 1024) % ch);
        acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, B[i & 3], acc[i & 3], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 1, nw = argc > 2 ? atoi(argv[2]) : 4, ns = argc > 3 ? atoi(argv[3]) : 2;
  const int chkb = argc > 4 ? atoi(argv[4]) : 24, nmfma = argc > 5 ? atoi(argv[5]) : 24, wgpc = argc > 6 ? atoi(argv[6]) : 2;
  const int ch = chkb * 1024, wchunks = 30, nchunks = 30 * 8;
  const int nwg = 256 * wgpc;
  char *w, *x;
  float* out;
  hipMalloc(&w, (size_t)wchunks * ch);
  hipMalloc(&x, (size_t)nwg * nw * 32 * 512 + 4096);
  hipMalloc(&out, 4096);
  hipMemset(w, 0, (size_t)wchunks * ch);
  hipMemset(x, 0, (size_t)nwg * nw * 32 * 512 + 4096);
  const size_t lds = (size_t)ns * ch;
  if (nw == 4) hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  else hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    if (nw == 4) hipLaunchKernelGGL(k<4>, dim3(nwg), dim3(256), lds, 0, w, x, out, mode, ns, ch, nchunks, wchunks, nmfma);
    else hipLaunchKernelGGL(k<8>, dim3(nwg), dim3(512), lds, 0, w, x, out, mode, ns, ch, nchunks, wchunks, nmfma);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double us_per_chunk = best * 1e3 / nchunks;
  printf("mode %d waves %d slots %d chunk %d KB mfma/chunk %d wg/CU %d: %.3f ms, %.3f us per chunk step, DMA %.1f GB/s per CU, %.2f TB/s chip\n",
         mode, nw, ns, chkb, nmfma, wgpc, best, us_per_chunk, (mode & 1) ? ch * wgpc / us_per_chunk * 1e-3 : 0.0,
         (mode & 1) ? ch * (double)nwg / us_per_chunk * 1e-6 : 0.0);
  return 0;
}
