#include <hip/hip_runtime.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(char* src, unsigned bytes, unsigned* out, int shift) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) ((unsigned*)smem)[i] = 0xdeadbeefu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t srd = __builtin_amdgcn_make_buffer_rsrc(src, 0, bytes, 0x00020000);
  unsigned voff = (unsigned)((lane + shift) * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)smem, 16, voff, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024 / 4; i += blockDim.x) out[i] = ((unsigned*)smem)[i];
}
int main() {
  char* d; unsigned* o;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  unsigned h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  hipMemcpy(d, h, 4096, hipMemcpyHostToDevice);
  for (int shift : {0, -2, 60}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, 1024u, o, shift);
    unsigned r[256]; hipMemcpy(r, o, 1024, hipMemcpyDeviceToHost);
    printf("shift %d:", shift);
    for (int l = 0; l < 64; ++l) printf(" %x", r[l * 4]);
    printf("\n");
  }
  return 0;
}
