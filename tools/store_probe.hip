// tools/store_probe.hip: how long a wave waits for the acknowledgement of a global store / atomic (s_waitcnt vmcnt(0) after it)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void sp(unsigned long long* buf, unsigned long long* out) {
  unsigned long long* q = buf + blockIdx.x * 64 + threadIdx.x;
  unsigned long long t[8];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  t[0] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) { *(volatile unsigned long long*)q = i; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  t[1] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) { __hip_atomic_store(q, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  t[2] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) { __hip_atomic_fetch_add(q, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
  t[3] = __builtin_amdgcn_s_memtime();
  unsigned long long acc = 0;
  for (int i = 0; i < 64; ++i) { acc += __hip_atomic_fetch_add(q, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1; }
  t[4] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) { acc += __hip_atomic_load(q + (acc & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1; }
  t[5] = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < 64; ++i) { *(volatile unsigned long long*)q = i; acc += __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1; }
  t[6] = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { for (int i = 0; i < 6; ++i) out[i] = t[i + 1] - t[i]; out[7] = acc; }
}
int main() {
  unsigned long long *buf, *out; hipMalloc(&buf, 1 << 20); hipMalloc(&out, 64); hipMemset(buf, 0, 1 << 20);
  for (int nb : {1, 32}) {
    sp<<<nb, 64>>>(buf, out); hipDeviceSynchronize(); sp<<<nb, 64>>>(buf, out); hipDeviceSynchronize();
    unsigned long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("%2d blocks: plain store + wait %.0f | agent atomic store + wait %.0f | no-return atomic add + wait %.0f | returning atomic add %.0f | sc1 load %.0f | store then sc1 load of the same address %.0f clocks\n",
           nb, h[0] / 64.0, h[1] / 64.0, h[2] / 64.0, h[3] / 64.0, h[4] / 64.0, h[5] / 64.0);
  }
  return 0;
}
