// Shared between the generic time-major GEMM (gemm_tm.hip) and its static-schedule 8-wave form (gemm_tm8.hip).
#pragma once
#include "wae_common.hpp"

#define TM_MAX_SRC 4
#define TM_PLAIN 0
#define TM_RESIDUAL 1  // out = alpha * (acc + res[t])
#define TM_GATE_BWD 2  // acc = du (NT = Hp/32 tiles); out (t, 2Hp) = [da | db] from z (t, 2Hp)
// The wide decoder head (skip / head widths above 256: the fused csrc/head_fwd.hip / head_bwd.hip keep every tile of a time
// column in one wave) runs as separate launches of this kernel, intermediate activations through HBM:
#define TM_BIAS_RELU 3  // out = relu(alpha * (bias[m] + acc)), aux = fp32 bias (M)             (wavenet.py:208-213)
#define TM_RELU_BWD 4   // out = aux[t][m] > 0 ? alpha * acc : 0, aux = the saved activation      (autograd of the ReLUs)
#define TM_CE 5         // acc = bias + logits (M = Op): optional (B,O,T) store, nll / lse of the shifted targets
#define TM_CE_BWD 6     // out = (softmax(bias + acc) - onehot(target[t+1])) * w[t]  from the saved lse

struct TmCe {           // modes 5 / 6 (vqwae_train.py:363-379 with the shift of :764)
  float* logits;
  const int32_t* target;
  float* nll;
  float* lse;
  const int32_t* lengths;
  float inv_count;
  int O;
};

struct TmArgs {
  unsigned long long* stamps;   // diagnostic builds only, else null
  const char* src[TM_MAX_SRC];
  int64_t src_stride[TM_MAX_SRC];  // elements per row
  int src_cols[TM_MAX_SRC];        // multiple of CK
  int src_shift[TM_MAX_SRC];       // operand row = t + shift (zero outside [0,T))
  int nsrc;
  const char* w;
  char* out;
  int64_t out_stride;
  const char* aux;  // RESIDUAL: res (t, M) ; GATE_BWD: z (t, 2Hp)
  int64_t aux_stride;
  float alpha;
  int B, T, mode;
  int interleave;  // chunk q -> source q % nsrc, column block q / nsrc (all sources equally wide)
  int flags;       // wae_tm_desc.flags
  TmCe ce;
  int nslices;     // gemm_tm8s_kernel: output slices of 256 rows, adjacent in the launch order (else 0)
};


// gemm_tm8.hip: 16-bit, one source, M = 256: 8 waves x 32 columns, one workgroup per CU, weights through an 8-slot ring of K = 32
// half-chunks.  Sets *handled when it launched; leaves it false for every shape it has no instantiation of.
int wae_gemm_tm8_launch(const TmArgs& a, int dtype, int M, hipStream_t st, bool* handled);
