// Shared between the generic fused-layer kernel (glu_fwd.hip) and its static-schedule instantiations (glu_fwd_static.hip).
#pragma once
#include "wae_common.hpp"

struct GluArgs {
  const char* x_in;
  const char* x_conv;  // operand of the dilated convolution: x_in, or dropout(x_in) in training with p > 0 (modules.py:127-128)
  char* x_out;
  const char* c_up;
  char* u_out;
  const float* zb;
  char* z_save;
  const char* w;
  const float* bias_out;
  int64_t zb_stride;
  int64_t u_stride;  // elements per time row of u_out
  int B, T, Rp, Ccp, Hp, ktaps, dilation, flags;
  int nslot;  // LDS ring slots (>= 2); weight chunk q lives in slot q % nslot and is requested nslot-1 chunks ahead
  unsigned long long* stamps;  // diagnostic only (wae_debug_set_stamps): 16 x u64 per workgroup, else null
};

// glu_fwd_static.hip: 16-bit instantiations whose whole chunk schedule (ring slots, request counts, tap offsets) is a compile-time
// constant.  Returns WAE_OK and sets *handled when it launched; leaves *handled false for every geometry it has no instantiation of.
int wae_glu_static_launch(const GluArgs& a, int dtype, hipStream_t st, bool* handled);
