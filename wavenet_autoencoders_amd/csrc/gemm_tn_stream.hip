// wae_gemm_tn_stream: ALL weight-gradient contractions of a backward pass in ONE launch (bf16 operands, fp32 result)
//
//   C_j[m][n] += alpha_j * sum_{clip b, t} P_j[b,t][m] * Q_j[b, t + shift_j][n]        for every job j
//
// Same arithmetic as csrc/gemm_tn.hip (autograd of the 1x1 and dilated convolutions, modules.py:134-160,
// wavenet.py:203,211-212), different decomposition.  gemm_tn.hip cuts every contraction into 128x128 output tiles and
// splits the time axis ~50 ways per layer launch: arithmetic intensity 64 FLOP per byte moved from L2, ~70 MB of fp32
// atomics per launch, one launch per layer.  Here a *job* is a 384 x 256 output region contracted over the whole batch;
// the (job, time-slab) work list of every layer is cut into one contiguous share per CU ("stream-K"), so that
//   * a workgroup keeps 96 accumulator tiles (12 waves x 2x4) resident while it streams its share of the time axis
//     (157 FLOP per byte from L2, 2.4x fewer bytes than 128x128 tiles),
//   * the fp32 atomics shrink to one flush per (workgroup, job) boundary -- (#CUs + #jobs) regions per step instead of
//     ~1200 per layer,
//   * all CUs finish together whatever the mix of region shapes.
// Operand slabs (32 time rows) go global -> LDS by LDS-DMA (no staging registers) into a 3-slot ring; the MFMA operands
// are read transposed from LDS (ds_read_b64_tr_b16: both operands have the contraction index along their rows).
// The host (backward.py: StreamTable) builds the job and segment tables once per (B, T).
#include "wae_common.hpp"

struct TsJob {
  const char* P;   // (B,T,p_stride) bf16, already offset to the region's first column
  const char* Q;   // (B,T,q_stride) bf16, already offset to the region's first column
  float* C;        // top-left of the region in the fp32 output
  int64_t p_stride, q_stride, ldc;   // elements
  int m_valid, n_valid;              // valid columns of P (<= 384) / Q (<= 256) inside the region, multiples of 8
  int shift;                         // Q row = t + shift
  int ones_col;                      // region-local index (n_valid <= ones_col < 256) of the virtual all-ones Q column, or -1;
                                     // clip b's sums go to C column ones_col + b
  float alpha;
  int pad_;
};
struct TsSeg {
  int job, slab_begin, slab_end;     // slabs are numbered b * slabs_per_clip + t / 32
};
struct TsArgs {
  const TsJob* jobs;
  const TsSeg* segs;
  const int* wg_seg;   // [nwg + 1] prefix offsets into segs
  int B, T, spc;
};

#define TS_KT 32
#define TS_PP 832      // P slab row pitch: 384 bf16 + 64 B (pitch = 64 mod 256: conflict-free transposed reads)
#define TS_QP 576      // Q slab row pitch: 256 bf16 + 64 B
#define TS_SP (TS_KT * TS_PP)
#define TS_SQ (TS_KT * TS_QP)
#define TS_SLOT (TS_SP + TS_SQ)
#define TS_NS 3
#define TS_PU (TS_PP / 16)   // 16-byte units per P row (52)
#define TS_QU (TS_QP / 16)   // 36
#define TS_NPP (TS_SP / 1024)  // 1-KiB DMA pieces per P slab (26)
#define TS_NPQ (TS_SQ / 1024)  // 18
#define TS_NW 12
#define TS_MAXPC 4             // ceil(44 / 12)

typedef __attribute__((ext_vector_type(2))) unsigned ts_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned ts_u32x4;

__device__ __forceinline__ void ts_wait_vmcnt(int w) {
#define TS_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (w < 8 ? w : 8) {
    TS_VMC(0) TS_VMC(1) TS_VMC(2) TS_VMC(3) TS_VMC(4) TS_VMC(5) TS_VMC(6) TS_VMC(7)
    default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
  }
#undef TS_VMC
}

__global__ void __launch_bounds__(TS_NW * 64, 1) gemm_tn_stream_kernel(TsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (wave % 6) * 64, wn = (wave / 6) * 128;
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;

  const int seg_b = p.wg_seg[blockIdx.x], seg_e = p.wg_seg[blockIdx.x + 1];
  for (int si = seg_b; si < seg_e; ++si) {
    const TsSeg sg = p.segs[si];
    const TsJob jb = p.jobs[sg.job];
    const int up_valid = (jb.m_valid + 7) >> 3, uq_valid = (jb.n_valid + 7) >> 3;
    // this wave's DMA pieces: piece pc = wave + 12 j of the slot image [P slab | Q slab]; lane -> 16-byte unit.
    // (row, unit) of a lane are recomputed from the lane id at every use: values kept live across the MFMA blocks
    // would be spilled (168 registers, 128 of them accumulators), and a scratch reload drains the DMA queue.
    auto piece_geom = [&](int ln, int j, int& row, int& col) -> bool {
      const int pc = wave + TS_NW * j;
      if (pc < TS_NPP) {
        const int u = pc * 64 + ln;
        row = u / TS_PU; col = u - row * TS_PU;
        return col < up_valid;
      } else if (pc < TS_NPP + TS_NPQ) {
        const int u = (pc - TS_NPP) * 64 + ln;
        row = u / TS_QU; col = u - row * TS_QU;
        return col < uq_valid;
      }
      row = 0; col = 0;
      return false;
    };
    int np_issued = 0;
#pragma unroll
    for (int j = 0; j < TS_MAXPC; ++j) {
      int r_, c_;
      if (__any(piece_geom(lane, j, r_, c_))) ++np_issued;
    }
    // active MFMA tiles of this wave (wave-uniform)
    const int n_end = jb.ones_col >= 0 ? jb.ones_col + p.B : jb.n_valid;
    const int nmt = min(max((jb.m_valid - wm + 31) >> 5, 0), 2);
    const int nnt = min(max((n_end - wn + 31) >> 5, 0), 4);
    const bool active = nmt > 0 && nnt > 0;

    __syncthreads();   // every wave is done with the previous segment's slabs
    // The virtual all-ones column: one Q column per clip (ones_col + b), outside the DMA'd columns; only the column of
    // the clip being contracted holds ones, so C[m][ones_col + b] collects sum_t P[b,t][m].  All start at zero.
    if (jb.ones_col >= 0) {
      for (int i = threadIdx.x; i < TS_NS * TS_KT * p.B; i += TS_NW * 64) {
        const int bb = i % p.B, rr = i / p.B;
        const int slot = rr / TS_KT, r = rr - slot * TS_KT;
        *(__bf16*)(smem + slot * TS_SLOT + TS_SP + r * TS_QP + (jb.ones_col + bb) * 2) = (__bf16)0.0f;
      }
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // slabs whose every row pairs with a Q row outside the clip contribute nothing: both cursors skip them
    auto useful = [&](int s) {
      const int t0 = (s % p.spc) * TS_KT;
      return t0 + TS_KT - 1 + jb.shift >= 0 && t0 + jb.shift < p.T;
    };
    auto next_useful = [&](int s) {
      while (s < sg.slab_end && !useful(s)) ++s;
      return s;
    };
    auto issue = [&](int s, int slot) {
      const int b = s / p.spc, t0 = (s - b * p.spc) * TS_KT;
      char* dst = smem + slot * TS_SLOT;
      int ln = threadIdx.x & 63;
      asm volatile("" : "+v"(ln));   // keeps the geometry below from being hoisted out of the slab loop
#pragma unroll
      for (int j = 0; j < TS_MAXPC; ++j) {
        const int pc = wave + TS_NW * j;
        int row, col;
        if (piece_geom(ln, j, row, col)) {
          const char* src;
          if (pc < TS_NPP) {
            const int t = min(t0 + row, p.T - 1);
            src = jb.P + (((int64_t)b * p.T + t) * jb.p_stride) * 2 + col * 16;
          } else {
            const int t = min(max(t0 + row + jb.shift, 0), p.T - 1);
            src = jb.Q + (((int64_t)b * p.T + t) * jb.q_stride) * 2 + col * 16;
          }
          dma_piece(src, dst + pc * 1024);
        }
      }
    };
    // rows of P outside the clip, or paired with a Q row outside it, must not contribute: zero them once landed
    auto zero_invalid_rows = [&](int s, int slot) {
      const int t0 = (s % p.spc) * TS_KT;
      if (t0 + TS_KT <= p.T && t0 + jb.shift >= 0 && t0 + TS_KT - 1 + jb.shift < p.T) return;
      int ln = threadIdx.x & 63;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int j = 0; j < TS_MAXPC; ++j) {
        const int pc = wave + TS_NW * j;
        int row, col;
        if (pc < TS_NPP && piece_geom(ln, j, row, col)) {
          const int t = t0 + row;
          if (t >= p.T || t + jb.shift < 0 || t + jb.shift >= p.T) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            *(f32x4*)(smem + slot * TS_SLOT + pc * 1024 + ln * 16) = z;
          }
        }
      }
    };

    // ---- pipeline: slab k of the sequence lives in slot k % NS; requests run NS-1 slabs ahead -------------------
    int s_issue = next_useful(sg.slab_begin), s_comp = s_issue;
    int k_issue = 0, k_comp = 0, cur_b = -1;
    for (; k_issue < TS_NS - 1 && s_issue < sg.slab_end; ++k_issue) {
      issue(s_issue, k_issue % TS_NS);
      s_issue = next_useful(s_issue + 1);
    }
    while (s_comp < sg.slab_end) {
      const int slot = k_comp % TS_NS;
      ts_wait_vmcnt(np_issued * (k_issue - k_comp - 1));   // slab k_comp landed; younger requests stay in flight
      zero_invalid_rows(s_comp, slot);
      // bare barrier (not __syncthreads(): its fence is lowered to s_waitcnt vmcnt(0) and would drain the prefetch);
      // the zero-fill stores, if any, are retired first
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s_issue < sg.slab_end) {
        issue(s_issue, k_issue % TS_NS);
        ++k_issue;
        s_issue = next_useful(s_issue + 1);
      }
      const int b = s_comp / p.spc;
      if (jb.ones_col >= 0 && b != cur_b) {   // clip change (workgroup-uniform): move the ones to the new clip's column
        if (threadIdx.x < TS_NS * TS_KT) {
          const int slot2 = threadIdx.x / TS_KT, r = threadIdx.x - slot2 * TS_KT;
          __bf16* qrow = (__bf16*)(smem + slot2 * TS_SLOT + TS_SP + r * TS_QP) + jb.ones_col;
          if (cur_b >= 0) qrow[cur_b] = (__bf16)0.0f;
          qrow[b] = (__bf16)1.0f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      cur_b = b;
      if (active) {
        const unsigned sp = lds0 + slot * TS_SLOT, sq = sp + TS_SP;
        int ln = threadIdx.x & 63;
        asm volatile("" : "+v"(ln));
        // transposed-read lane geometry (see csrc/gemm_tn.hip tn_load_frags)
        const int hh2 = ln >> 5, grp = (ln >> 4) & 1, q4 = (ln & 15) >> 2, pp = ln & 3;
#pragma unroll
        for (int k0 = 0; k0 < TS_KT; k0 += 16) {
          const unsigned ap = sp + (k0 + 8 * hh2 + q4) * TS_PP + (16 * grp + 4 * pp) * 2 + wm * 2;
          const unsigned bp = sq + (k0 + 8 * hh2 + q4) * TS_QP + (16 * grp + 4 * pp) * 2 + wn * 2;
          // One k-step (16 time rows) of the wave's 2 x 4 tiles.  Hand-allocated operand registers: a transposed read
          // returns half an MFMA operand, and letting the compiler pair the halves costs a copy of every fragment
          // (24 VGPRs this 168-register kernel does not have).  v[144:151] = A fragments of M-tiles 0,1;
          // v[152:167] = B fragments of N-tiles 0..3.  Tiles beyond the job's valid region are computed too (their
          // results are never written out).
          asm volatile(
              "ds_read_b64_tr_b16 v[144:145], %8\n\t"
              "ds_read_b64_tr_b16 v[146:147], %8 offset:%10\n\t"
              "ds_read_b64_tr_b16 v[152:153], %9\n\t"
              "ds_read_b64_tr_b16 v[154:155], %9 offset:%11\n\t"
              "ds_read_b64_tr_b16 v[156:157], %9 offset:64\n\t"
              "ds_read_b64_tr_b16 v[158:159], %9 offset:%12\n\t"
              "ds_read_b64_tr_b16 v[148:149], %8 offset:64\n\t"
              "ds_read_b64_tr_b16 v[150:151], %8 offset:%13\n\t"
              "ds_read_b64_tr_b16 v[160:161], %9 offset:128\n\t"
              "ds_read_b64_tr_b16 v[162:163], %9 offset:%14\n\t"
              "ds_read_b64_tr_b16 v[164:165], %9 offset:192\n\t"
              "ds_read_b64_tr_b16 v[166:167], %9 offset:%15\n\t"
              "s_waitcnt lgkmcnt(8)\n\t"
              "v_mfma_f32_32x32x16_bf16 %0, v[144:147], v[152:155], %0\n\t"
              "s_waitcnt lgkmcnt(6)\n\t"
              "v_mfma_f32_32x32x16_bf16 %1, v[144:147], v[156:159], %1\n\t"
              "s_waitcnt lgkmcnt(4)\n\t"
              "v_mfma_f32_32x32x16_bf16 %4, v[148:151], v[152:155], %4\n\t"
              "v_mfma_f32_32x32x16_bf16 %5, v[148:151], v[156:159], %5\n\t"
              "s_waitcnt lgkmcnt(2)\n\t"
              "v_mfma_f32_32x32x16_bf16 %2, v[144:147], v[160:163], %2\n\t"
              "v_mfma_f32_32x32x16_bf16 %6, v[148:151], v[160:163], %6\n\t"
              "s_waitcnt lgkmcnt(0)\n\t"
              "v_mfma_f32_32x32x16_bf16 %3, v[144:147], v[164:167], %3\n\t"
              "v_mfma_f32_32x32x16_bf16 %7, v[148:151], v[164:167], %7\n\t"
              : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),
                "+v"(acc[1][2]), "+v"(acc[1][3])
              : "v"(ap), "v"(bp), "n"(4 * TS_PP), "n"(4 * TS_QP), "n"(4 * TS_QP + 64), "n"(4 * TS_PP + 64), "n"(4 * TS_QP + 128),
                "n"(4 * TS_QP + 192)
              : "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157",
                "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167");
        }
      }
      ++k_comp;
      s_comp = next_useful(s_comp + 1);
    }

    // the MFMAs above are opaque to the compiler's hazard recogniser: cover the MFMA-result -> VALU-read wait states here
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int lne = threadIdx.x & 63;
    asm volatile("" : "+v"(lne));   // the 128 output addresses are formed here, not hoisted above the slab loop
    const int nl = lne & 31, hh = lne >> 5;
    // ---- C += alpha * acc   (lane = column n, registers = rows m); fp32 atomics: other workgroups own other
    //      time ranges of the same region ------------------------------------------------------------------------
    if (active) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (!(i < nmt && j < nnt)) continue;
          const int col = wn + 32 * j + nl;
          if (!(col < jb.n_valid || (jb.ones_col >= 0 && col >= jb.ones_col && col < jb.ones_col + p.B))) continue;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (row < jb.m_valid) atomicAdd(jb.C + (int64_t)row * jb.ldc + col, jb.alpha * acc[i][j][r]);
          }
        }
    }
  }
}

extern "C" int wae_gemm_tn_stream(const wae_ts_job* jobs_dev, const wae_ts_seg* segs_dev, const int32_t* wg_seg_dev,
                                  int32_t nwg, int32_t B, int32_t T, void* stream) {
  WAE_REQUIRE(jobs_dev && segs_dev && wg_seg_dev && nwg > 0 && B > 0 && T > 0, "gemm_tn_stream: bad arguments");
  static_assert(sizeof(wae_ts_job) == sizeof(TsJob), "wae_ts_job and TsJob must have the same layout");
  static_assert(sizeof(wae_ts_seg) == sizeof(TsSeg), "wae_ts_seg and TsSeg must have the same layout");
  static_assert(TS_NPP * 1024 == TS_SP && TS_NPQ * 1024 == TS_SQ, "slab images must be whole DMA pieces");
  TsArgs a;
  a.jobs = (const TsJob*)jobs_dev;
  a.segs = (const TsSeg*)segs_dev;
  a.wg_seg = wg_seg_dev;
  a.B = B; a.T = T; a.spc = (T + TS_KT - 1) / TS_KT;
  const size_t lds = (size_t)TS_NS * TS_SLOT;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)gemm_tn_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      wae_set_error("gemm_tn_stream: cannot raise dynamic LDS to %zu", lds);
      return WAE_EHIP;
    }
    attr_done = true;
  }
  hipLaunchKernelGGL(gemm_tn_stream_kernel, dim3(nwg), dim3(TS_NW * 64), lds, as_stream(stream), a);
  return wae_check_launch("gemm_tn_stream");
}
