// tools/handover_probe.hip: latency of one store -> visible-to-a-polling-CU hand-over inside an XCD, measured with the shared clock: the
// writer puts s_memtime into the payload, the reader subtracts it from its own s_memtime when the value appears.  Reader pacing varies.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SLEEP, int NPOLL>
__global__ void hp(unsigned long long* slot, unsigned long long* out, int iters) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == 8 ? 1 : -1);
  if (me < 0) return;
  // side 0 writes slot[0], waits for the echo in slot[64]; side 1 polls slot[0] with NPOLL waves x 64 lanes (all read the same line), echoes
  unsigned long long sum = 0, mx = 0, mn = ~0ull;
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) {
      if (threadIdx.x == 0) {
        const unsigned long long v = (__builtin_amdgcn_s_memtime() << 32) | (unsigned)i;
        *(volatile unsigned long long*)slot = v;
      }
      if (threadIdx.x < 64) {
        int spins = 0;
        while ((unsigned)__hip_atomic_load(slot + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)i && ++spins < 100000) __builtin_amdgcn_s_sleep(2);
        if (spins >= 100000) return;
      }
      __syncthreads();
      // a pause, so that the reader is already polling when the next store goes out
      for (int k = 0; k < 8; ++k) __builtin_amdgcn_s_sleep(32);
    } else {
      unsigned long long v;
      int spins = 0;
      do {
        v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
      } while ((unsigned)v != (unsigned)i && ++spins < 100000);
      if (spins >= 100000) return;
      const unsigned long long now = __builtin_amdgcn_s_memtime();
      const unsigned long long lat = (unsigned)now - (unsigned)(v >> 32);
      sum += lat; mx = lat > mx ? lat : mx; mn = lat < mn ? lat : mn;
      __syncthreads();
      if (threadIdx.x == 0) *(volatile unsigned long long*)(slot + 64) = (unsigned long long)i;
    }
  }
  if (me == 1 && threadIdx.x == 0) { out[0] = sum; out[1] = mn; out[2] = mx; }
}
template <int SLEEP, int NPOLL>
void run(unsigned long long* slot, unsigned long long* out) {
  hipMemset(slot, 0, 4096); hipMemset(out, 0, 64);
  const int iters = 1000;
  hp<SLEEP, NPOLL><<<16, 64 * NPOLL>>>(slot, out, iters);
  hipDeviceSynchronize();
  unsigned long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  printf("reader: %d wave(s), s_sleep %2d between polls: store -> seen %.0f clocks on average (min %llu, max %llu)\n", NPOLL, SLEEP, h[0] / (double)iters, h[1], h[2]);
  fflush(stdout);
}
int main() {
  unsigned long long *slot, *out; hipMalloc(&slot, 4096); hipMalloc(&out, 64);
  run<0, 1>(slot, out); run<1, 1>(slot, out); run<4, 1>(slot, out); run<0, 4>(slot, out); run<1, 4>(slot, out);
  return 0;
}
