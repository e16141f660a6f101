// Shared between the backward pair kernels (glu_bwd.hip: 4 waves x 128 columns, two workgroups per CU; glu_bwd8.hip: 8 waves x 256 columns,
// both operands through LDS).
#pragma once
#include "wae_common.hpp"

struct GbArgs {
  const char* dz;       // (B,T,dz_stride): dz_l columns (2Hp) of layer l
  const char* g_next;   // (B,T,Rp): dx_{l+1}-hat
  char* g_out;          // (B,T,Rp): dx_l-hat
  const char* dskip;    // (B,T,Sp)
  const char* z_prev;   // (B,T,2Hp): pre-activations of layer l-1
  char* dz_prev;        // (B,T,dz_stride): dz_{l-1} columns
  const char* w_x;      // first_gemm_map(Rp, k*2Hp): chunks [q][blk][m]
  const char* w_uo;     // second_gemm_map(Hp, Rp): [mt][kb] in accumulator-row k order
  const char* w_us;     // first_gemm_map(Hp, Sp)
  int64_t dz_stride;
  float alpha;
  int B, T, Sp, ktaps, dilation;
  unsigned long long* stamps;   // diagnostic builds (-DWAE_GBP_STAMPS) only, else null
  // 16-bit pair kernel, dc folded in (Ccp = 64, 3 taps): dc += Wc_l^T dz_l rides on the chunks of the shift-0 tap, whose operand
  // fragments ARE dz_l[t]; the running sum over the layers lives in an fp32 (B,T,64) array, the last launch writes the 16-bit dc
  const char* w_c;      // this layer's chunks of the dc weight stream (first_gemm_map(Ccp, 2Hp): 8 KiB per column block), or null
  float* dc_acc;        // (B,T,64) fp32
  char* dc_out;         // (B,T,64) in the storage dtype: written instead of dc_acc when dc_mode & 2
  int dc_mode;          // bit 0: add the previous sum (dc_acc); bit 1: write dc_out (the last layer of the sweep)
  int last;             // 1: layer 0 -- phase A + epilogue A only (there is no layer below to gate)
};


// one tile (32 rows x 32 channels, accumulator layout) into a row-major fp32 / 16-bit array through the wave's 4-KiB staging tile:
// v = acc (+ old[row]) ; old is the fp32 running sum, the result goes to out32 (fp32) or out16 (storage dtype)
// (the old values are fetched by the caller ahead of time: rmw_fetch; rows at or beyond rows_valid fetch the last valid row)
__device__ __forceinline__ void rmw_fetch(f32x4 (&w)[4], const float* old32, int tile, int rows_valid, int lane) {
  const int rr = lane >> 3, ck = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) w[i] = *(const f32x4*)(old32 + (int64_t)min(i * 8 + rr, rows_valid - 1) * 64 + tile * 32 + ck * 4);
}
template <typename E>
__device__ __forceinline__ void stage_rmw_tile(char* stg, const f32x16& y, const f32x4 (&w)[4], bool add, float* out32, char* out16,
                                               int tile, int rows_valid, int lane) {
  const int n = lane & 31, h = lane >> 5;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 v = {y[4 * g], y[4 * g + 1], y[4 * g + 2], y[4 * g + 3]};
    *(f32x4*)(stg + n * 128 + (((2 * g + h) ^ (n & 7)) << 4)) = v;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const int rr = lane >> 3, ck = lane & 7;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + rr;
    f32x4 v = *(const f32x4*)(stg + row * 128 + ((ck ^ (row & 7)) << 4));
    if (add) v = v + w[i];
    if (row < rows_valid) {
      const int64_t o = (int64_t)row * 64 + tile * 32 + ck * 4;
      if (out16) *(typename ET<E>::vec4*)(out16 + o * (int64_t)sizeof(E)) = from_f32x4<E>(v);
      else *(f32x4*)(out32 + o) = v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}


// glu_bwd8.hip: 16-bit storage, Rp = 256, three taps, the conditioning gradient folded in (w_c != null): 8 waves x 256 columns, one workgroup
// per CU.  Sets *handled when it launched; leaves it false for every shape it has no instantiation of.
int wae_glu_bwd8_launch(const GbArgs& a, int dtype, int ntx, int ntu, hipStream_t st, bool* handled);
