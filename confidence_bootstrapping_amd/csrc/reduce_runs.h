// Device helpers shared by the score-model kernels (tp_conv.hip, tp_conv_bf16.hip) and the confidence-model kernel
// (fctp_conv.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace cbd {

// Makes a register value opaque to the optimiser at this point of the program (an empty volatile asm that reads and writes it): used to
// keep addresses computed BEFORE an MFMA chain from being recomputed inside it, and to tie side-effect-free FMAs of fully unrolled tile
// loops to their tile.
template <class P>
__device__ __forceinline__ void pin(P& p) { asm volatile("" : "+v"(p)); }

// A wave-uniform pointer kept opaque and in SGPRs, and the global-address-space pointer type for loads of the form
// `global_load v, v_lane_offset, s[base:base+1] offset:imm` (tp_conv_dev.h explains why the weight streams are read that way).
template <class P>
__device__ __forceinline__ void pin_s(P& p) { asm volatile("" : "+s"(p)); }
template <class T>
using GPtr = const T __attribute__((address_space(1)))*;

// ReLU in ONE VALU instruction (v_med3_f32 with a finite upper bound).  fmaxf(x, 0) costs two -- hipcc first canonicalises the operand
// with v_max_f32 x, x -- and med3(x, 0, +inf) is folded back into that pair.  (An inline-asm v_max_f32 is one instruction too, but
// the hazard recogniser does not see its operands and drops the wait states between an MFMA and the read of its result.)
__device__ __forceinline__ float relu1(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, 3.0e38f); }

// Run-length sums of one 32-edge message tile without atomics (bitwise reproducible).  `msg` = LDS tile [col][OUT_STR] of the
// 32 edges' messages, `sl` = LDS [32] aggregating node of every edge (sorted; -1 for the lanes past the end of the group, which only
// follow valid ones).  Lane = column.  A run that starts at the tile's first edge goes to first_sum[tile] (`fs`), one that reaches
// edge 31 to last_sum[tile] (`ls`), any other run (strictly inside the tile) is the node's only contribution from this group and is
// stored directly to run_acc[node]; conv_finalize adds the pieces in tile order.  A run that ends at the last edge of the group's
// partial tile has no other tile either: stored as interior.
// The run boundaries are the same for every column, so they are found once (ballot over sl) and kept in a scalar mask.
// Round 5 (from the register-stationary kernel's reduction, tp_conv_bf16s.hip): ONE pass for all columns -- lane l sums column
// col_lo + l and, where there are more than 64 columns, column col_lo + 64 + l as well, so the scalar run bookkeeping (two thirds of the
// instructions of a pass) is paid once instead of twice; the tile is read two edges at a time (ds_read_b64; OUT_STR = 34 keeps the 32
// lanes of a read on 64 distinct banks) when the stride is even; and a group of eight edges without a run boundary -- most groups: a
// tile holds one to three runs -- is summed without a branch.  Same additions in the same order as before: results bitwise unchanged.
template <int NODE_STR, int OUT_STR, bool TWO>
__device__ __forceinline__ void reduce_runs_impl(const float* __restrict__ msg, const int s_me, const unsigned starts, const int last, int lane,
                                                 int out_dim, float* __restrict__ fs, float* __restrict__ ls, float* __restrict__ run_acc,
                                                 int col_lo) {
  const int col = col_lo + lane, col2 = col + 64;
  const bool on = col < out_dim, two = TWO && col2 < out_dim;
  const float* oc = msg + (on ? col : col_lo) * OUT_STR;
  const float* oc2 = msg + (two ? col2 : col_lo) * OUT_STR;
  float v[32], w[TWO ? 32 : 1];
  if constexpr (OUT_STR % 2 == 0) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const f32x2 a = reinterpret_cast<const f32x2*>(oc)[k];
      v[2 * k] = a.x; v[2 * k + 1] = a.y;
      if constexpr (TWO) {
        const f32x2 b = reinterpret_cast<const f32x2*>(oc2)[k];
        w[2 * k] = b.x; w[2 * k + 1] = b.y;
      }
    }
  } else {
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) {
      v[jj] = oc[jj];
      if constexpr (TWO) w[jj] = oc2[jj];
    }
  }
  float sum = 0.f, sum2 = 0.f;
  int a0 = 0;
#pragma unroll
  for (int g8 = 0; g8 < 4; ++g8) {
    if (((starts >> (8 * g8)) & 0xffu) == 0u) {
#pragma unroll
      for (int jj = 8 * g8; jj < 8 * g8 + 8; ++jj) {
        sum += v[jj];
        if constexpr (TWO) sum2 += w[jj];
      }
    } else {
#pragma unroll
      for (int jj = 8 * g8; jj < 8 * g8 + 8; ++jj) {
        if (jj > 0 && ((starts >> jj) & 1u)) {   // run [a0, jj-1] is complete
          const int node = __builtin_amdgcn_readlane(s_me, a0);
          float* dst = a0 == 0 ? fs : run_acc + (size_t)node * NODE_STR;
          if (on) dst[col] = sum;
          if (two) dst[col2] = sum2;
          sum = 0.f; sum2 = 0.f;
          a0 = jj;
        }
        sum += v[jj];
        if constexpr (TWO) sum2 += w[jj];
      }
    }
  }
  if (last >= 0) {   // run that reaches edge 31 of a full tile
    float* dst = a0 == 0 ? fs : ls;
    if (on) dst[col] = sum;
    if (two) dst[col2] = sum2;
  }
}

// MAX_COLS: compile-time bound of out_dim - col_lo at the call site.  The single pass covers at most 128 columns from col_lo (lane l:
// col_lo + l and col_lo + 64 + l) -- a wider layer would silently lose its tail, so the bound is part of the signature.  MSG_OFF: offset
// (floats) of `msg` from the 16-byte aligned LDS base; with an even stride the tile is read as f32x2, which needs it 8-byte aligned.
template <int NODE_STR, int OUT_STR, int MAX_COLS, int MSG_OFF>
__device__ __forceinline__ void reduce_runs(const float* __restrict__ msg, const int* __restrict__ sl, int lane, int out_dim,
                                            float* __restrict__ fs, float* __restrict__ ls, float* __restrict__ run_acc, int col_lo = 0) {
  static_assert(MAX_COLS <= 128, "reduce_runs covers at most 128 columns in its single pass");
  static_assert(OUT_STR % 2 != 0 || MSG_OFF % 2 == 0, "ds_read_b64 of the message tile needs an 8-byte aligned tile base");
  const int jl = lane & 31;
  const int s_me = sl[jl], s_prev = sl[jl > 0 ? jl - 1 : 0];
  const unsigned starts = (unsigned)__ballot(lane < 32 && s_me != s_prev);   // bit jj: edge jj starts a new run (bit 0 is never set)
  const int last = __builtin_amdgcn_readlane(s_me, 31);
  if (col_lo + 64 < out_dim) reduce_runs_impl<NODE_STR, OUT_STR, true>(msg, s_me, starts, last, lane, out_dim, fs, ls, run_acc, col_lo);      // wave-uniform
  else reduce_runs_impl<NODE_STR, OUT_STR, false>(msg, s_me, starts, last, lane, out_dim, fs, ls, run_acc, col_lo);
}

}  // namespace cbd
