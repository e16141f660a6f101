// tools/pingpong_probe.hip: one-way latency of a flag/value hand-over between two workgroups on the same XCD (blocks 0 and 8 of a
// 16-block launch), by ping-pong: store variant x poll variant.  hipcc --offload-arch=gfx950 -O2 tools/pingpong_probe.hip -o /tmp/pp && /tmp/pp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ST, int LD>
__global__ void pp(unsigned long long* slot, unsigned long long* out, int iters) {
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == 8 ? 1 : -1);
  if (me < 0 || threadIdx.x >= 64) return;
  unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  unsigned long long* mine = slot + me * 64 + threadIdx.x;          // 512 B per side: 4 lines
  unsigned long long* theirs = slot + (1 - me) * 64 + threadIdx.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) {
      const unsigned long long v = ((unsigned long long)i << 32) | (unsigned)i;
      if (ST == 0) *(volatile unsigned long long*)mine = v;
      else if (ST == 1) __hip_atomic_store(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (ST == 2) __hip_atomic_exchange(mine, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (ST == 3) { *(volatile unsigned long long*)mine = v; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      else if (ST == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n" :: "v"(mine), "v"(v) : "memory");
      else if (ST == 5) asm volatile("global_store_dwordx2 %0, %1, off nt\n" :: "v"(mine), "v"(v) : "memory");
      else if (ST == 6) asm volatile("global_store_dwordx2 %0, %1, off\n" :: "v"(mine), "v"(v) : "memory");   // plain, no wait
    }
    unsigned long long v;
    int spins = 0;
    do {
      if (LD == 0) v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (LD == 1) v = __hip_atomic_fetch_add(theirs, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (LD == 2) v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      else if (LD == 3) { asm volatile("buffer_inv sc1" ::: "memory"); v = *(volatile unsigned long long*)theirs; }
      else if (LD == 4) { v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_sleep(4); }
      else if (LD == 6) { v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_sleep(16); }
      else if (LD == 7) { v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_sleep(64); }
      else if (LD == 9) {     // scalar load, glc: past the scalar cache, from L2 (uniform address: every lane polls lane 0's granule)
        const unsigned long long* q = slot + (1 - me) * 64;
        asm volatile("s_load_dwordx2 %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(q) : "memory");
      }
      else if (LD == 10) { asm volatile("global_load_dwordx2 %0, %1, off nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs) : "memory"); }
      else if (LD == 11) { asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs) : "memory"); }
      else if (LD == 12) { asm volatile("buffer_inv sc0\n global_load_dwordx2 %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs) : "memory"); }
      else if (LD == 13) { asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(theirs) : "memory"); }
      else if (LD == 8) { __builtin_amdgcn_s_sleep(8); v = __hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
      else { v = __hip_atomic_load(theirs + (spins & 1) * 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    } while ((unsigned)v != (unsigned)i && ++spins < 20000);      // bounded: a variant that never becomes visible must not hang the box
    if (spins >= 20000) { if (threadIdx.x == 0) out[me * 2] = ~0ull; return; }
    if (me == 1) {
      const unsigned long long w = ((unsigned long long)i << 32) | (unsigned)i;
      if (ST == 0) *(volatile unsigned long long*)mine = w;
      else if (ST == 1) __hip_atomic_store(mine, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (ST == 2) __hip_atomic_exchange(mine, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else if (ST == 3) { *(volatile unsigned long long*)mine = w; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      else if (ST == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n" :: "v"(mine), "v"(w) : "memory");
      else if (ST == 5) asm volatile("global_store_dwordx2 %0, %1, off nt\n" :: "v"(mine), "v"(w) : "memory");
      else if (ST == 6) asm volatile("global_store_dwordx2 %0, %1, off\n" :: "v"(mine), "v"(w) : "memory");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[me * 2] = t1 - t0; out[me * 2 + 1] = xcc & 15; }
}
template <int ST, int LD>
void run(const char* name, unsigned long long* slot, unsigned long long* out, int nthr = 64) {
  hipMemset(slot, 0, 4096); hipMemset(out, 0, 64);
  const int iters = 2000;
  pp<ST, LD><<<16, nthr>>>(slot, out, iters);
  if (hipDeviceSynchronize() != hipSuccess) { printf("%s: failed\n", name); return; }
  unsigned long long h[4]; hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
  if (h[0] == ~0ull || h[2] == ~0ull) printf("%-58s never became visible within 20000 polls\n", name);
  else printf("%-58s one way %.0f clocks (XCC %llu / %llu)\n", name, h[0] / (2.0 * iters), h[1], h[3]);
  fflush(stdout);
}
int main() {
  unsigned long long *slot, *out; hipMalloc(&slot, 4096); hipMalloc(&out, 64);
  run<0, 0>("plain volatile store, sc1 load", slot, out);
  run<1, 0>("agent-scope atomic store, sc1 load", slot, out);
  run<2, 0>("workgroup-scope atomic exchange (RMW in L2), sc1 load", slot, out);
  run<3, 0>("plain store + s_waitcnt vmcnt(0), sc1 load", slot, out);
  run<4, 0>("store sc0 sc1, sc1 load", slot, out);
  run<5, 0>("store nt, sc1 load", slot, out);
  run<6, 0>("asm store without the compiler's wait, sc1 load", slot, out);
  run<6, 9>("asm store without the compiler's wait, scalar load glc", slot, out);
  run<6, 4>("asm store without the compiler's wait, sc1 load + s_sleep 4", slot, out);
  run<0, 1>("plain store, poll by agent-scope atomic add 0 (RMW in L2)", slot, out);
  run<2, 1>("atomic exchange, poll by agent-scope atomic add 0", slot, out);
  run<0, 2>("plain store, system-scope load (sc0 sc1)", slot, out);
  run<0, 3>("plain store, buffer_inv sc1 + plain load", slot, out);
  run<0, 4>("plain store, sc1 load + s_sleep 4 between polls", slot, out);
  run<0, 9>("plain store, scalar load glc", slot, out);
  run<0, 10>("plain store, vector load nt", slot, out);
  run<0, 11>("plain store, vector load sc0 (asm)", slot, out);
  run<0, 12>("plain store, buffer_inv sc0 + plain load", slot, out);
  run<0, 13>("plain store, vector load sc0 sc1 nt", slot, out);
  run<0, 6>("plain store, sc1 load + s_sleep 16", slot, out);
  run<0, 7>("plain store, sc1 load + s_sleep 64", slot, out);
  run<0, 8>("plain store, s_sleep 8 + sc1 load (sleep first)", slot, out);
  run<0, 0>("plain store, sc1 load, ONE lane per side", slot, out, 1);
  run<0, 4>("plain store, sc1 load + s_sleep 4, ONE lane per side", slot, out, 1);
  run<0, 0>("plain store, sc1 load, 8 lanes per side (64 B)", slot, out, 8);
  return 0;
}
