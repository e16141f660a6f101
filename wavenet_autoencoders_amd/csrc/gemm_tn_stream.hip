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

// timing-only ablation (tools/ablate_ts.sh): -DWAE_TS_ABLATE=bits; 1 no operand DMA, 2 no MFMA block (neither LDS reads nor MFMAs),
// 4 MFMAs without their LDS reads, 8 no flush of the accumulators, 16 LDS reads without the MFMAs
#ifndef WAE_TS_ABLATE
#define WAE_TS_ABLATE 0
#endif
#define TS_ABL(bit) ((WAE_TS_ABLATE & (bit)) != 0)

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
  int job, slab_begin, slab_end;     // job of team member 0; slabs are numbered b * slabs_per_clip + t / 32
};
struct TsArgs {
  const TsJob* jobs;
  const TsSeg* segs;
  const int* team_seg;   // [nteams + 1] prefix offsets into segs
  int nteams, team_size;
  int B, T, spc;
  int* pace;     // [nteams][8] slab positions consumed by each member (zeroed by the caller before the launch), or null
  int window;    // a paced member requests at most `window` slabs beyond the slowest of the members it follows
  int pace_from; // members [pace_from, team_size) are paced against members [0, pace_from), which only publish
};

#ifndef TS_KT
#define TS_KT 32     // time rows per slab (16 or 32; the host numbers slabs in units of 32 rows)
#endif
#define TS_RATIO (32 / TS_KT)
#define TS_PP 832      // P slab row pitch: 384 bf16 + 64 B (pitch = 64 mod 256: conflict-free transposed reads)
#define TS_QP 576      // Q slab row pitch: 256 bf16 + 64 B
#define TS_SP (TS_KT * TS_PP)
#define TS_SQ (TS_KT * TS_QP)
#define TS_SLOT (TS_SP + TS_SQ)
#ifndef TS_NS
#define TS_NS 3
#endif
#define TS_PU (TS_PP / 16)   // 16-byte units per P row (52)
#define TS_QU (TS_QP / 16)   // 36
#define TS_NPP (TS_SP / 1024)  // 1-KiB DMA pieces per P slab (26)
#define TS_NPQ (TS_SQ / 1024)  // 18
#define TS_NW 12
#define TS_MAXPC ((TS_NPP + TS_NPQ + TS_NW - 1) / TS_NW)   // 4 = ceil(44 / 12)

typedef __attribute__((ext_vector_type(2))) unsigned ts_u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned ts_u32x4;

__device__ __forceinline__ void ts_wait_vmcnt(int w) {
#define TS_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (w < 16 ? w : 16) {
    TS_VMC(0) TS_VMC(1) TS_VMC(2) TS_VMC(3) TS_VMC(4) TS_VMC(5) TS_VMC(6) TS_VMC(7)
    TS_VMC(8) TS_VMC(9) TS_VMC(10) TS_VMC(11) TS_VMC(12) TS_VMC(13) TS_VMC(14) TS_VMC(15)
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
#undef TS_VMC
}

// Team pacing.  The members of a team (the jobs of one layer: three taps, the conditioning 1x1, conv1x1_out) stream the same
// time slabs, and four of them read the same dz slab as their P operand -- but the XCD's 4 MiB L2 is shared by ~6 teams, a team's
// share holds about five slabs of everything it streams, and unpaced members drift apart by more than that (the light jobs run
// ahead): round 1 measured an L2 hit rate of 28 % and 6.07 GB fetched for 3.47 GB of unique operands -- no sharing at all.
// Each member publishes the position it has consumed ((segment << 20) | slab in segment); wave 0 keeps a copy of the team's row,
// refreshed by a scalar load (SMEM: counted by lgkmcnt, so the vmcnt bookkeeping of the LDS-DMA ring is untouched) that was
// issued one slab earlier, and holds the whole workgroup in front of its barrier while its next REQUEST would run more than
// `window` slabs ahead of the slowest member.  Purely a timing device: a wait that does not end within ~0.3 ms switches pacing
// off for the rest of the launch (no member can hang the device), results never depend on it.
typedef __attribute__((ext_vector_type(8))) int ts_i32x8;
__device__ __forceinline__ void ts_row_request(ts_i32x8& row, const int* p) {
  asm volatile("s_load_dwordx8 %0, %1, 0x0 glc" : "=s"(row) : "s"(p));
}
__device__ __forceinline__ void ts_row_wait(ts_i32x8& row) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(row)); }
__device__ __forceinline__ int ts_row_min_first(const ts_i32x8& row, int n) {
  int m = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < 8; ++i)
    if (i < n) m = min(m, row[i]);
  return m;
}

#ifdef WAE_TS_STAMPS
// diagnostic build (tools/stamps_ts.py): TsArgs::pace is an int64 [workgroup][8] array; wave 0 adds up the shader-clock ticks it
// spends in each phase of the slab loop.  Never quote run times from this build.
__device__ __forceinline__ int64_t ts_clock() {
  int64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
__device__ __forceinline__ int64_t ts_wallclock() {
  int64_t t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
  return t;
}
#define TS_STAMP(var) const int64_t var = ts_clock()
#define TS_ACC(dst, a, b) dst += (b) - (a)
#else
#define TS_STAMP(var)
#define TS_ACC(dst, a, b)
#endif

template <typename E, bool PACED>   // E: __bf16 or f16 (the DMA and the transposed LDS reads move bits), fp32 result
__global__ void __launch_bounds__(TS_NW * 64, 1) gemm_tn_stream_kernel(TsArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = (wave % 6) * 64, wn = (wave / 6) * 128;
  const unsigned lds0 = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)smem;

  // Teams: team_size consecutive *logical* workgroups walk the same segment list, member m on job (segment job + m):
  // the jobs of one layer share operands (dz is read by every tap), and members that sweep the same time range at
  // the same pace find them in L2.  Workgroups are dealt round-robin over the 8 XCDs in launch order, so logical ids
  // are chosen to make a team's members land in ONE XCD (speed only).
  int logical = blockIdx.x;
  if ((gridDim.x & 7) == 0) logical = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int team = logical / p.team_size, member = logical - team * p.team_size;
  if (team >= p.nteams) return;
  const int seg_b = p.team_seg[team], seg_e = p.team_seg[team + 1];
  int* pace_row = PACED ? p.pace + team * 8 : nullptr;
  bool pacing = PACED && member >= p.pace_from;
  const bool leads = PACED && member < p.pace_from;
  ts_i32x8 row = {};
#ifdef WAE_TS_STAMPS
  int64_t k_wait = 0, k_bar = 0, k_issue = 0, k_mma = 0, k_iter = 0, k_seg = 0;
  const int64_t k_t0 = ts_clock(), k_w0 = ts_wallclock();
#endif
  // a member publishes the position of its next REQUEST: requests are what fetch from HBM / L2, and the member with the
  // smallest request position never waits (no cycle of waits can form, whatever slabs the members skip)
  auto publish = [&](int pos) {   // one lane, fire and forget (an agent-scope store: visible to the other CUs' scalar loads)
    if constexpr (PACED) {
      if (leads && threadIdx.x == 0) __hip_atomic_store(pace_row + member, pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  for (int si = seg_b; si < seg_e; ++si) {
    TsSeg sg = p.segs[si];
    sg.slab_begin *= TS_RATIO; sg.slab_end *= TS_RATIO;
    const TsJob jb = p.jobs[sg.job + member];
    const int seg_pos = (si - seg_b) << 20;
    if (jb.m_valid <= 0) {   // null job (e.g. the last layer has no conv1x1_out gradient): done with this segment at once
      publish(seg_pos + (1 << 20));
      continue;
    }
    const int up_valid = (jb.m_valid + 7) >> 3, uq_valid = (jb.n_valid + 7) >> 3;
    // This wave's DMA pieces: piece pc = wave + 12 j of the slot image [P slab | Q slab]; lane -> one 16-byte unit.
    // Per lane and piece: validity bit and byte offset from the slab's first row (kept in 5 registers across the MFMA
    // blocks; the slab loop itself must stay lean: at 3 waves per SIMD every 100 scalar/vector instructions per
    // slab and wave cost as much issue time as a third of the slab's MFMAs).
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
    unsigned vbits = 0;
    unsigned poff[TS_MAXPC];
#pragma unroll
    for (int j = 0; j < TS_MAXPC; ++j) {
      int r_, c_;
      const bool v = piece_geom(lane, j, r_, c_);
      if (v) vbits |= 1u << j;
      const int64_t stride_b = (wave + TS_NW * j < TS_NPP ? jb.p_stride : jb.q_stride) * 2;
      poff[j] = (unsigned)(r_ * stride_b + c_ * 16);
      if (__any(v)) ++np_issued;
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
        *(E*)(smem + slot * TS_SLOT + TS_SP + r * TS_QP + (jb.ones_col + bb) * 2) = (E)0.0f;
      }
    }

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Slab cursors (clip b, first row t0, linear slab number s), advanced without divisions.  Slabs whose every row pairs
    // with a Q row outside the clip contribute nothing: the useful rows of a clip are [t_lo, t_hi), both cursors skip
    // the rest.
    const int t_lo = max(0, (-jb.shift - (TS_KT - 1) + TS_KT - 1) & ~(TS_KT - 1));
    const int t_hi = jb.shift > 0 ? min(p.spc * TS_KT, (p.T - jb.shift + TS_KT - 1) & ~(TS_KT - 1)) : p.spc * TS_KT;
    struct Cur { int b, t0, s; };
    auto cur_fix = [&](Cur& c) {   // move to the first useful slab at or after the current position
      if (c.t0 < t_lo) c.t0 = t_lo;
      if (c.t0 >= t_hi) { ++c.b; c.t0 = t_lo; }
      c.s = c.b * p.spc + c.t0 / TS_KT;
    };
    auto cur_adv = [&](Cur& c) {
      c.t0 += TS_KT; ++c.s;
      if (c.t0 >= t_hi) { ++c.b; c.t0 = t_lo; c.s = c.b * p.spc + t_lo / TS_KT; }
    };
    Cur ci, cc;
    ci.b = sg.slab_begin / p.spc; ci.t0 = (sg.slab_begin - ci.b * p.spc) * TS_KT; ci.s = sg.slab_begin;
    cur_fix(ci);
    cc = ci;

    auto issue = [&](const Cur& c, int slot) {
      char* dst = smem + slot * TS_SLOT;
      const bool whole = c.t0 + TS_KT <= p.T && c.t0 + jb.shift >= 0 && c.t0 + TS_KT - 1 + jb.shift < p.T;
      if (whole) {   // every row of the slab inside the clip (all but the first/last slabs of a clip): uniform bases
        const char* pb = jb.P + (((int64_t)c.b * p.T + c.t0) * jb.p_stride) * 2;
        const char* qb = jb.Q + (((int64_t)c.b * p.T + c.t0 + jb.shift) * jb.q_stride) * 2;
#pragma unroll
        for (int j = 0; j < TS_MAXPC; ++j) {
          const int pc = wave + TS_NW * j;
          if ((vbits & (1u << j)) && !TS_ABL(1)) dma_piece((pc < TS_NPP ? pb : qb) + poff[j], dst + pc * 1024);
        }
      } else {       // rows clamped into the clip one by one
        int ln = threadIdx.x & 63;
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int j = 0; j < TS_MAXPC; ++j) {
          const int pc = wave + TS_NW * j;
          int row, col;
          if (piece_geom(ln, j, row, col)) {
            const char* src;
            if (pc < TS_NPP) {
              const int t = min(c.t0 + row, p.T - 1);
              src = jb.P + (((int64_t)c.b * p.T + t) * jb.p_stride) * 2 + col * 16;
            } else {
              const int t = min(max(c.t0 + row + jb.shift, 0), p.T - 1);
              src = jb.Q + (((int64_t)c.b * p.T + t) * jb.q_stride) * 2 + col * 16;
            }
            if (!TS_ABL(1)) dma_piece(src, dst + pc * 1024);
          }
        }
      }
    };
    // rows of P outside the clip, or paired with a Q row outside it, must not contribute: zero them once landed
    auto zero_invalid_rows = [&](const Cur& c, int slot) {
      if (c.t0 + TS_KT <= p.T && c.t0 + jb.shift >= 0 && c.t0 + TS_KT - 1 + jb.shift < p.T) return;
      int ln = threadIdx.x & 63;
      asm volatile("" : "+v"(ln));
#pragma unroll
      for (int j = 0; j < TS_MAXPC; ++j) {
        const int pc = wave + TS_NW * j;
        int row, col;
        if (pc < TS_NPP && piece_geom(ln, j, row, col)) {
          const int t = c.t0 + row;
          if (t >= p.T || t + jb.shift < 0 || t + jb.shift >= p.T) {
            float zf = 0.f;
            asm volatile("" : "+v"(zf));   // materialised here: a zero quad kept live across the slab loop ends up in scratch
            const f32x4 z = {zf, zf, zf, zf};
            *(f32x4*)(smem + slot * TS_SLOT + pc * 1024 + ln * 16) = z;
          }
        }
      }
    };

    // ---- pipeline: slab k of the sequence lives in slot k % NS; requests run NS-1 slabs ahead -------------------
    int slot_i = 0, slot_c = 0, ahead = 0, cur_b = -1;   // ahead = slabs requested and not yet consumed
    for (; ahead < TS_NS - 1 && ci.s < sg.slab_end; ++ahead) {
      issue(ci, slot_i);
      slot_i = slot_i + 1 == TS_NS ? 0 : slot_i + 1;
      cur_adv(ci);
    }
    if constexpr (PACED) {
      publish(seg_pos + min(ci.s, sg.slab_end) - sg.slab_begin);
      if (pacing && wave == 0) ts_row_request(row, pace_row);
    }
    while (cc.s < sg.slab_end) {
      TS_STAMP(s0);
      if (!TS_ABL(1)) ts_wait_vmcnt(np_issued * (ahead - 1));   // the oldest requested slab has landed; younger requests stay in flight
      if constexpr (PACED) {
        if (pacing && wave == 0 && ci.s < sg.slab_end) {
          // the slab about to be requested against the slowest other member's request position (one slab old, refreshed below)
          ts_row_wait(row);
          const int mine = seg_pos + (ci.s - sg.slab_begin);
          int spins = 0;
          while (mine - ts_row_min_first(row, p.pace_from) > p.window) {
            __builtin_amdgcn_s_sleep(4);
            ts_row_request(row, pace_row);
            ts_row_wait(row);
            if (++spins > 512) { pacing = false; break; }   // ~0.5 ms: give up pacing, never hang
          }
          if (pacing) ts_row_request(row, pace_row);
        }
      }
      TS_STAMP(s1);
      zero_invalid_rows(cc, slot_c);
      // bare barrier (not __syncthreads(): its fence is lowered to s_waitcnt vmcnt(0) and would drain the prefetch);
      // the zero-fill stores, if any, are retired first
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      TS_STAMP(s2);
      if (ci.s < sg.slab_end) {
        issue(ci, slot_i);
        slot_i = slot_i + 1 == TS_NS ? 0 : slot_i + 1;
        cur_adv(ci);
        ++ahead;
        publish(seg_pos + min(ci.s, sg.slab_end) - sg.slab_begin);
      }
      if (jb.ones_col >= 0 && cc.b != cur_b) {   // clip change (workgroup-uniform): move the ones to the new clip's column
        if (threadIdx.x < TS_NS * TS_KT) {
          const int slot2 = threadIdx.x / TS_KT, r = threadIdx.x - slot2 * TS_KT;
          E* qrow = (E*)(smem + slot2 * TS_SLOT + TS_SP + r * TS_QP) + jb.ones_col;
          if (cur_b >= 0) qrow[cur_b] = (E)0.0f;
          qrow[cc.b] = (E)1.0f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
      cur_b = cc.b;
      TS_STAMP(s3);
      if (active && !TS_ABL(2)) {
        // transposed-read lane geometry (see csrc/gemm_tn.hip tn_load_frags), formed here from an opaque copy of the lane id:
        // two registers that stay live across the slab loop are two more than the 168-register budget has -- hipcc then parks
        // one in scratch, and its reload in front of the MFMA block carries s_waitcnt vmcnt(0): the whole LDS-DMA ring drained
        // on every slab (the paced instantiation of round 2 did exactly that; tools/check_asm_regs.py now looks for it)
        int lx = threadIdx.x & 63;
        asm volatile("" : "+v"(lx));
        const int hh2 = lx >> 5, grp = (lx >> 4) & 1, q4 = (lx & 15) >> 2, pp = lx & 3;
        const unsigned lane_row = 8 * hh2 + q4, lane_col = (16 * grp + 4 * pp) * 2;
        const unsigned ap = lds0 + slot_c * TS_SLOT + lane_row * TS_PP + lane_col + wm * 2;
        const unsigned bp = lds0 + slot_c * TS_SLOT + TS_SP + lane_row * TS_QP + lane_col + wn * 2;
        // Two k-steps (16 time rows each) of the wave's 2 x 4 tiles.  Hand-allocated operand registers: a transposed
        // read returns half an MFMA operand, and letting the compiler pair the halves costs a copy of every fragment
        // (24 VGPRs this 168-register kernel does not have).  v[144:151] = A fragments of M-tiles 0,1;
        // v[152:167] = B fragments of N-tiles 0..3.  Tiles beyond the job's valid region are computed too (their
        // results are never written out).
#if WAE_TS_ABLATE & 4
#define TS_RD(x) ""
#else
#define TS_RD(x) x
#endif
#if WAE_TS_ABLATE & 16
#define TS_MM(x) ""
#else
#define TS_MM(x) x
#endif
#define TS_KSTEP(KO, MF)                                                                                                     \
        asm volatile(                                                                                                      \
            TS_RD("ds_read_b64_tr_b16 v[144:145], %8 offset:%10\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[146:147], %8 offset:%11\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[152:153], %9 offset:%14\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[154:155], %9 offset:%15\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[156:157], %9 offset:%16\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[158:159], %9 offset:%17\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[148:149], %8 offset:%12\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[150:151], %8 offset:%13\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[160:161], %9 offset:%18\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[162:163], %9 offset:%19\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[164:165], %9 offset:%20\n\t")                                                             \
            TS_RD("ds_read_b64_tr_b16 v[166:167], %9 offset:%21\n\t")                                                             \
            "s_waitcnt lgkmcnt(8)\n\t"                                                                                     \
            TS_MM("v_mfma_f32_32x32x16_" MF " %0, v[144:147], v[152:155], %0\n\t")                                                  \
            "s_waitcnt lgkmcnt(6)\n\t"                                                                                     \
            TS_MM("v_mfma_f32_32x32x16_" MF " %1, v[144:147], v[156:159], %1\n\t")                                                  \
            "s_waitcnt lgkmcnt(4)\n\t"                                                                                     \
            TS_MM("v_mfma_f32_32x32x16_" MF " %4, v[148:151], v[152:155], %4\n\t")                                                  \
            TS_MM("v_mfma_f32_32x32x16_" MF " %5, v[148:151], v[156:159], %5\n\t")                                                  \
            "s_waitcnt lgkmcnt(2)\n\t"                                                                                     \
            TS_MM("v_mfma_f32_32x32x16_" MF " %2, v[144:147], v[160:163], %2\n\t")                                                  \
            TS_MM("v_mfma_f32_32x32x16_" MF " %6, v[148:151], v[160:163], %6\n\t")                                                  \
            "s_waitcnt lgkmcnt(0)\n\t"                                                                                     \
            TS_MM("v_mfma_f32_32x32x16_" MF " %3, v[144:147], v[164:167], %3\n\t")                                                  \
            TS_MM("v_mfma_f32_32x32x16_" MF " %7, v[148:151], v[164:167], %7\n\t")                                                  \
            : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]),        \
              "+v"(acc[1][2]), "+v"(acc[1][3])                                                                             \
            : "v"(ap), "v"(bp), "n"((KO) * TS_PP), "n"((KO) * TS_PP + 4 * TS_PP), "n"((KO) * TS_PP + 64),                   \
              "n"((KO) * TS_PP + 4 * TS_PP + 64), "n"((KO) * TS_QP), "n"((KO) * TS_QP + 4 * TS_QP), "n"((KO) * TS_QP + 64), \
              "n"((KO) * TS_QP + 4 * TS_QP + 64), "n"((KO) * TS_QP + 128), "n"((KO) * TS_QP + 4 * TS_QP + 128),             \
              "n"((KO) * TS_QP + 192), "n"((KO) * TS_QP + 4 * TS_QP + 192)                                                  \
            : "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", \
              "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167")
        if constexpr (sizeof(E) == 2 && ET<E>::DT == WAE_F16) {
          TS_KSTEP(0, "f16");
          if constexpr (TS_KT == 32) TS_KSTEP(16, "f16");
        } else {
          TS_KSTEP(0, "bf16");
          if constexpr (TS_KT == 32) TS_KSTEP(16, "bf16");
        }
#undef TS_KSTEP
      }
      slot_c = slot_c + 1 == TS_NS ? 0 : slot_c + 1;
      --ahead;
      cur_adv(cc);
#ifdef WAE_TS_STAMPS
      asm volatile("s_nop 0" ::: "memory");
      TS_STAMP(s4);
      TS_ACC(k_wait, s0, s1); TS_ACC(k_bar, s1, s2); TS_ACC(k_issue, s2, s3); TS_ACC(k_mma, s3, s4);
      ++k_iter;
#endif
    }
    if constexpr (PACED) {
      if (wave == 0) ts_row_wait(row);           // no scalar load left in flight across the epilogue
      publish(seg_pos + (1 << 20));              // this member is done with the segment
    }

    // the MFMAs above are opaque to the compiler's hazard recogniser: cover the MFMA-result -> VALU-read wait states here
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int lne = threadIdx.x & 63;
    asm volatile("" : "+v"(lne));   // the 128 output addresses are formed here, not hoisted above the slab loop
    const int nl = lne & 31, hh = lne >> 5;
    // ---- C += alpha * acc   (lane = column n, registers = rows m); fp32 atomics: other workgroups own other
    //      time ranges of the same region ------------------------------------------------------------------------
    if (active && !TS_ABL(8)) {
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
#ifdef WAE_TS_STAMPS
    ++k_seg;
#endif
  }
#ifdef WAE_TS_STAMPS
  if (threadIdx.x == 0) {
    int64_t* o = (int64_t*)p.pace + (int64_t)blockIdx.x * 8;
    o[0] = ts_clock() - k_t0; o[1] = ts_wallclock() - k_w0; o[2] = k_wait; o[3] = k_bar; o[4] = k_issue; o[5] = k_mma; o[6] = k_iter;
    o[7] = k_seg;
  }
#endif
}

extern "C" int wae_gemm_tn_stream(int32_t dtype, const wae_ts_job* jobs_dev, const wae_ts_seg* segs_dev, const int32_t* team_seg_dev,
                                  int32_t nteams, int32_t team_size, int32_t nwg, int32_t B, int32_t T, int32_t* pace,
                                  int32_t window, int32_t pace_from, void* stream) {
  WAE_REQUIRE(dtype == WAE_BF16 || dtype == WAE_F16, "gemm_tn_stream: 16-bit operands only (fp32 runs take wae_gemm_tn_tiles)");
  WAE_REQUIRE(jobs_dev && segs_dev && team_seg_dev && nteams > 0 && team_size > 0 && nwg >= nteams * team_size && B > 0 && T > 0,
              "gemm_tn_stream: bad arguments");
  WAE_REQUIRE(B <= 64, "gemm_tn_stream: at most 64 clips per launch (one all-ones column per clip)");
  static_assert(sizeof(wae_ts_job) == sizeof(TsJob), "wae_ts_job and TsJob must have the same layout");
  static_assert(sizeof(wae_ts_seg) == sizeof(TsSeg), "wae_ts_seg and TsSeg must have the same layout");
  static_assert(TS_NPP * 1024 == TS_SP && TS_NPQ * 1024 == TS_SQ, "slab images must be whole DMA pieces");
  TsArgs a;
  a.jobs = (const TsJob*)jobs_dev;
  a.segs = (const TsSeg*)segs_dev;
  a.team_seg = team_seg_dev;
  a.nteams = nteams; a.team_size = team_size;
  a.B = B; a.T = T; a.spc = TS_RATIO * ((T + 31) / 32);
  a.pace = window > 0 ? pace : nullptr;
  a.window = window;
  a.pace_from = pace_from;
#ifdef WAE_TS_STAMPS
  a.pace = pace;      // the stamp array (int64 [nwg][8]); pacing is off in this build
  window = 0;
#endif
  const size_t lds = (size_t)TS_NS * TS_SLOT;
  const bool paced = window > 0 && a.pace != nullptr && team_size > 1 && team_size <= 8 && pace_from > 0 && pace_from < team_size;
  auto go = [&](auto kernel) -> int {
    static WaeLdsCache lds_cache;   // one per instantiation of this lambda's call operator, i.e. per kernel
    if (int rc = wae_ensure_lds((const void*)kernel, lds_cache, lds, "gemm_tn_stream"); rc != WAE_OK) return rc;
    hipLaunchKernelGGL(kernel, dim3(nwg), dim3(TS_NW * 64), lds, as_stream(stream), a);
    return WAE_OK;
  };
  int rc;
  if (dtype == WAE_F16) rc = paced ? go(gemm_tn_stream_kernel<f16, true>) : go(gemm_tn_stream_kernel<f16, false>);
  else rc = paced ? go(gemm_tn_stream_kernel<__bf16, true>) : go(gemm_tn_stream_kernel<__bf16, false>);
  if (rc != WAE_OK) return rc;
  return wae_check_launch("gemm_tn_stream");
}
