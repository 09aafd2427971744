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
// The run boundaries are the same for every column, so they are found once (ballot over sl) and kept in a scalar mask; a column's 32
// values are then fetched with 32 independent LDS reads and summed under scalar branches -- the former loop read sl[jj] and the value
// from LDS inside a data-dependent chain and cost ~24 k cycles per tile (in-kernel stamps), a quarter of a bf16 wave's lifetime.
// Same additions in the same order as before: results are bitwise unchanged.
template <int NODE_STR, int OUT_STR>
__device__ __forceinline__ void reduce_runs(const float* __restrict__ msg, const int* __restrict__ sl, int lane, int out_dim,
                                            float* __restrict__ fs, float* __restrict__ ls, float* __restrict__ run_acc, int col_lo = 0) {
  const int jl = lane & 31;
  const int s_me = sl[jl], s_prev = sl[jl > 0 ? jl - 1 : 0];
  const unsigned starts = (unsigned)__ballot(lane < 32 && s_me != s_prev);   // bit jj: edge jj starts a new run (bit 0 is never set)
  const int last = __builtin_amdgcn_readlane(s_me, 31);
  for (int col = col_lo + lane; col < out_dim; col += 64) {      // [col_lo, out_dim): the columns this wave's tile slice produces
    const float* oc = msg + col * OUT_STR;
    float v[32];
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) v[jj] = oc[jj];
    float sum = 0.f;
    int a0 = 0;
#pragma unroll
    for (int jj = 0; jj < 32; ++jj) {
      if (jj > 0 && ((starts >> jj) & 1u)) {   // run [a0, jj-1] is complete
        const int node = __builtin_amdgcn_readlane(s_me, a0);
        float* dst = a0 == 0 ? fs : run_acc + (size_t)node * NODE_STR;
        dst[col] = sum;
        sum = 0.f;
        a0 = jj;
      }
      sum += v[jj];
    }
    if (last >= 0) (a0 == 0 ? fs : ls)[col] = sum;   // run that reaches edge 31 of a full tile
  }
}

}  // namespace cbd
