// tools/clk_probe.hip: what one wave on one SIMD issues per clock (s_memtime ticks vs s_memrealtime, dependent / independent VALU chains,
// LDS round trip, ds_bpermute, L2-hit load round trip).  Build + run: hipcc --offload-arch=gfx950 -O2 tools/clk_probe.hip -o /tmp/clk && /tmp/clk
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned long long* out, float* buf, int nblk_active) {
  __shared__ float sm[1024];
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  float a = buf[threadIdx.x], b = 1.0001f, c = 0.5f;
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int i = 0; i < 1024; ++i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float x0 = a, x1 = b, x2 = c, x3 = a + 1;
#pragma unroll
  for (int i = 0; i < 256; ++i) {
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x0) : "v"(b), "v"(c));
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x1) : "v"(b), "v"(c));
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x2) : "v"(b), "v"(c));
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x3) : "v"(b), "v"(c));
  }
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  int idx = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 64; ++i) { idx = ((volatile int*)sm)[idx & 1023] > 1e9f ? 1 : (idx + 1) & 63; }
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  float s = a;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += __shfl_xor(s, 8, 64);
  unsigned long long t4 = __builtin_amdgcn_s_memtime();
  // dependent L2-hit loads (buf is small and was read before)
  int j = threadIdx.x;
  for (int i = 0; i < 64; ++i) j = (int)__hip_atomic_load(buf + (j & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 255;
  unsigned long long t5 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = t4 - t3; out[4] = t5 - t4; out[5] = t5 - t0; out[6] = r1 - r0;
  }
  buf[256 + threadIdx.x + blockIdx.x * 256] = x0 + x1 + x2 + x3 + idx + s + j;
}
int main() {
  unsigned long long* out; float* buf;
  hipMalloc(&out, 64); hipMalloc(&buf, 4 * (256 + 256 * 512)); hipMemset(buf, 0, 4 * (256 + 256 * 512));
  for (int nb : {1, 32, 256, 512}) for (int nt : {64, 256}) {
    probe<<<nb, nt>>>(out, buf, nb); hipDeviceSynchronize();
    probe<<<nb, nt>>>(out, buf, nb); hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, out, 56, hipMemcpyDeviceToHost);
    printf("blocks %3d x %3d thr: dep fmac %.2f ticks each, 4-way indep %.2f each, LDS dependent read %.1f, ds_bpermute+add %.1f, L2 load round trip %.1f; s_memtime/s_memrealtime = %.2f (realtime = 100 MHz -> s_memtime at %.0f MHz)\n",
           nb, nt, h[0] / 1024.0, h[1] / 1024.0, h[2] / 64.0, h[3] / 64.0, h[4] / 64.0, (double)h[5] / h[6], 100.0 * h[5] / h[6]);
  }
  return 0;
}
