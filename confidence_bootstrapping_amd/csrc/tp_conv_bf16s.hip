// bf16 tensor-product message passing with REGISTER-STATIONARY weights (BASELINE.json configs[3]; options bf16 + bf16_stationary).
//
// Same math as tp_conv_bf16.hip (FCBlock -> FasterTensorProduct -> segmented sum; reference models/tensor_layers.py:195-206, 66-117) for
// the 74 -> 74 layers, organised the other way round.  The streaming kernel gives every wave 64 edges and makes it read its group's whole
// FCBlock (56 tiles x 6 KB = 336 KB) per 64 edges: at the bf16 MFMA rate that is the CU's whole vector-memory return path (64 B/clk), so
// the matrix pipe and that path saturate together at ~0.5 (PMC round 4), and a quarter of every wave's lifetime (index -> gather round
// trips, first Linear, reduction) holds no MFMA at all.  Sharing tiles through LDS lost three times to the per-tile rendezvous, the role
// split with LDS-resident tiles to the per-slice repetition of those fixed phases.  Here
//   * a PERSISTENT workgroup of four waves (one per SIMD, up to 512 registers each) is bound to one FCBlock and keeps ALL of its second
//     Linear in registers for the whole launch: 53 tiles x 24 registers, 14 / 14 / 14 / 11 tiles per wave.  No weight is fetched per
//     edge any more; the vector-memory path carries only the gathers (0.7 KB per edge instead of 5.9);
//   * every wave processes ALL 32 edges of a unit, but only its own tiles: the fixed phases are paid once per unit by the workgroup,
//     not once per wave or slice.  The unit's gathered destination rows (mids) and its hidden activations h (the B operand) are shared
//     through LDS; ONE barrier per unit;
//   * waves 0..2 (0e tiles [0,14), [14,28), [28,38) + the scalar-mid tiles of block 1o) write partial output sums to LDS; wave 3 works
//     one unit behind them: it adds the three partials, runs the remaining vector tiles, finishes the message tile.  Wave 3 also runs
//     the first Linear (its three tiles stay in LDS) one unit AHEAD, wave 2 the run-length reduction two units behind, and all four
//     waves share the gathers three units ahead -- a software pipeline over units, every stage double- or quad-buffered in LDS;
//   * inside a wave the CG epilogue of tile k sits between the MFMAs of tile k + 1 (two accumulator sets), since a wave that is alone on
//     its SIMD has nobody else to fill its MFMA shadows.
// Results: the same pieces (first_sum / last_sum / run_acc per 32-edge tile) as the streaming kernel; the order in which a message's
// tile contributions are added differs (partials per wave), so the two kernels agree to fp32 rounding of those sums, not bitwise.
// Deterministic: no atomics, fixed orders.
#include <cstdlib>

#include "kernels.h"
#include "tp_conv_dev.h"
#include "tp_conv_bf16_dev.h"

namespace cbd {

constexpr int SW_WAVES = 4;
constexpr int SU = 32;                                   // edges per unit = one reduction tile
constexpr ConvShape SS = conv_shape(3, 3, true);         // the bf16 stream's layout (merged vector tails)
static_assert(SS.ntiles == 56 && SS.t0e == 38 && SS.t1o == 9 && SS.t1e - SS.vmerged == 3 && SS.t0o == 3 && SS.vmerged == 1,
              "the wave programs below are written for the 74 -> 74 layer shape");
constexpr int S_MAX_ROLES = 8;
constexpr int RED_WAVE = 0;                              // the wave that runs the run-length reduction

// ---- LDS map (floats)
constexpr int L_X_SLOT = 76 * 32;                        // gathered destination rows, transposed [col][32 edges]
constexpr int L_X = 0;                                   // [4 slots]
constexpr int L_FRAG_SLOT = V2_NFRAG * 64 * 4;           // one B operand: [6 k-steps][64 lanes][8 bf16] = 6 KB
constexpr int L_BX = L_X + 4 * L_X_SLOT;                 // [2 slots] first-Linear input (edge_attr | x_src[:32] | x_dst[:32]) as bf16 B operand
constexpr int L_H = L_BX + 2 * L_FRAG_SLOT;              // [4 slots] hidden activations h as bf16 B operand
constexpr int L_P_WAVE = 20 * 64;                        // partial sums of one wave: 16 (0e) + 3 (1o scalar sums) registers x 64 lanes
constexpr int L_P_SLOT = 3 * L_P_WAVE;
constexpr int L_P = L_H + 4 * L_FRAG_SLOT;               // [2 slots][3 waves]
constexpr int S_OSTR = 36;                               // message tile stride: lane = column reads four edges per ds_read_b128 without bank conflicts
constexpr int L_O_SLOT = NODE_DIM * S_OSTR;              // message tile [74][36]
constexpr int L_O = L_P + 2 * L_P_SLOT;                  // [2 slots]
constexpr int L_BIAS = L_O + 2 * L_O_SLOT;               // [57][32] fp32 bias rows of the stream
constexpr int L_FLW = L_BIAS + (SS.ntiles + 1) * 32;     // the three first-Linear tiles
constexpr int L_TOTAL = L_FLW + 3 * L_FRAG_SLOT;
constexpr int S_LDS_BYTES = L_TOTAL * 4;
static_assert(S_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(L_BX % 4 == 0 && L_H % 4 == 0 && L_P % 4 == 0 && L_O % 4 == 0 && L_BIAS % 4 == 0 && L_FLW % 4 == 0, "16-byte alignment");

struct RoleTableS {
  int n_roles;
  unsigned char role_of[CONV_MAX_GROUPS];      // role of every entry of ConvArgs::g
  const float* wstream[S_MAX_ROLES];
};

__device__ __forceinline__ int s_wave_sum(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// LDS-only barrier: the gathers of later units stay in flight across it (__syncthreads would wait for them)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// One tile: acc = C + W_tile * B, six dependent MFMAs; `epi(q)` is issued behind MFMA q (the previous tile's epilogue slices).
// The weights are never rewritten, so the in-flight-operand hazard of the streaming kernels (tp_conv_dev.h) does not exist here.
template <bool ZERO_C, class Epi>
__device__ __forceinline__ void s_chain(const bf16x8 (&w)[V2_NFRAG], const Act6& B, const f32x16& c, f32x16& acc, Epi epi) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) {
    if (q == 0) {
      if constexpr (ZERO_C) {
        const f32x16 zero = {};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[q], B.v[q], zero, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[q], B.v[q], c, 0, 0, 0);
      }
    } else {
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[q], B.v[q], acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    epi(q);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Two tiles at once: two independent chains interleaved (a dependent MFMA issues ~40 cycles after its predecessor, an independent one
// after 32: one chain alone keeps the matrix pipe at 0.8), `epi(s)` behind MFMA s = 0 .. 11 (the previous pair's epilogue slices).
template <class Epi>
__device__ __forceinline__ void s_chain2(const bf16x8 (&wa)[V2_NFRAG], const bf16x8 (&wb)[V2_NFRAG], const Act6& B, f32x16& acca, f32x16& accb,
                                         Epi epi) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) {
    if (q == 0) {
      const f32x16 zero = {};
      acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[q], B.v[q], zero, 0, 0, 0);
    } else {
      acca = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[q], B.v[q], acca, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    epi(2 * q);
    __builtin_amdgcn_sched_barrier(0);
    if (q == 0) {
      const f32x16 zero = {};
      accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[q], B.v[q], zero, 0, 0, 0);
    } else {
      accb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[q], B.v[q], accb, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    epi(2 * q + 1);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// bias rows of tile T as the C operand (accumulator layout: register r of lane half hf = row (r & 3) + 8 (r >> 2) + 4 hf)
__device__ __forceinline__ void s_load_bias(const float* bias_lds, int T, int hf, f32x16& cb) {
  const f32x4* b4 = reinterpret_cast<const f32x4*>(bias_lds + T * 32);
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const f32x4 b = b4[hf + 2 * qq];
    cb[4 * qq + 0] = b.x; cb[4 * qq + 1] = b.y; cb[4 * qq + 2] = b.z; cb[4 * qq + 3] = b.w;
  }
}

__device__ __forceinline__ void s_load_act(const float* slot, int lane, Act6& h) {
  const bf16x8* p = reinterpret_cast<const bf16x8*>(slot) + lane;
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) h.v[q] = p[q * 64];
}

constexpr int S_EPI_LO[V2_NFRAG + 1] = {0, 3, 6, 9, 12, 14, 16};

// Run-length sums of one message tile: reduce_runs (reduce_runs.h) with the aggregating-node ids in a register instead of LDS, and ONE
// pass over the 32 edges for all 74 columns -- lane l sums column l and, for l < 10, column 64 + l as well (the scalar run bookkeeping
// is shared); the tile is read with ds_read_b128 (stride 36), and a group of eight edges without a run boundary -- most of them: a unit
// holds one to three runs -- is summed without a single branch.  Same additions in the same order as reduce_runs.
template <int NODE_STR>
__device__ __forceinline__ void s_reduce_runs(const float* __restrict__ msg, int s_me, int lane, float* __restrict__ fs,
                                              float* __restrict__ ls, float* __restrict__ run_acc) {
  constexpr int REST = NODE_DIM - 64;
  const int s_up = __shfl_up(s_me, 1, 64);
  const int s_prev = (lane & 31) > 0 ? s_up : s_me;
  const unsigned starts = (unsigned)__ballot(lane < 32 && s_me != s_prev);
  const int last = __builtin_amdgcn_readlane(s_me, 31);
  const bool two = lane < REST;
  const f32x4* oc = reinterpret_cast<const f32x4*>(msg + lane * S_OSTR);
  const f32x4* oc2 = reinterpret_cast<const f32x4*>(msg + (64 + (two ? lane : 0)) * S_OSTR);
  float v[32], w[32];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const f32x4 a = oc[k], b = oc2[k];
    v[4 * k] = a.x; v[4 * k + 1] = a.y; v[4 * k + 2] = a.z; v[4 * k + 3] = a.w;
    w[4 * k] = b.x; w[4 * k + 1] = b.y; w[4 * k + 2] = b.z; w[4 * k + 3] = b.w;
  }
  float sum = 0.f, sum2 = 0.f;
  int a0 = 0;
#pragma unroll
  for (int g8 = 0; g8 < 4; ++g8) {
    if (((starts >> (8 * g8)) & 0xffu) == 0u) {
#pragma unroll
      for (int jj = 8 * g8; jj < 8 * g8 + 8; ++jj) { sum += v[jj]; sum2 += w[jj]; }
    } else {
#pragma unroll
      for (int jj = 8 * g8; jj < 8 * g8 + 8; ++jj) {
        if (jj > 0 && ((starts >> jj) & 1u)) {   // run [a0, jj-1] is complete
          const int node = __builtin_amdgcn_readlane(s_me, a0);
          float* dst = a0 == 0 ? fs : run_acc + (size_t)node * NODE_STR;
          dst[lane] = sum;
          if (two) dst[64 + lane] = sum2;
          sum = 0.f; sum2 = 0.f;
          a0 = jj;
        }
        sum += v[jj]; sum2 += w[jj];
      }
    }
  }
  if (last >= 0) {
    float* dst = a0 == 0 ? fs : ls;
    dst[lane] = sum;
    if (two) dst[64 + lane] = sum2;
  }
}

// ---- biased tiles (vector / pseudoscalar blocks) as software-pipelined steps: the CG epilogue of step s - 1 sits between the MFMAs of
//      step s (two accumulators, two sets of raw operands), the bias rows of step s + 1 are fetched from LDS behind MFMA 1 of step s
//      (MFMA 0, which read them as its C operand, has completed by then).  Accumulator register 3 q + o of lane half hf = (mid slot q,
//      output 3 hf + o) (common.h).  Mid kinds as in tp_conv_dev.h::mid1o / mid1e / mid0o; the raw operands are read from LDS one step
//      ahead and the cross / dot products formed in the epilogue slice that uses them.
struct VOut { float s1o[3], k1o[9], s1e[3], k1e[9], k0o[3]; };
typedef float RawT[VEC_TILE_I][3];
enum { VK_1O = 1, VK_1E = 2, VK_0O = 3 };
constexpr int S_T1O = 3 + SS.t0e, S_T1E = S_T1O + SS.t1o, S_T0O = S_T1E + SS.t1e - 1;      // stream indices of the blocks' first tiles
constexpr int S_OWN1E = VEC_TILE_I * (SS.t1e - 1);                                          // block 1e's own mids (the rest are guests of 0o's last tile)

template <int KIND, int T>
__device__ __forceinline__ void v_raw(const float* xc, RawT& r) {
#pragma unroll
  for (int q = 0; q < VEC_TILE_I; ++q) {
    const int i = VEC_TILE_I * T + q;
    int col = -1, n = 0;                 // first LDS row and number of rows (1 scalar, 3 vector)
    if (KIND == VK_1O) {
      if (i < NS) { col = i; n = 1; } else if (i < NS + SS.n1o) { col = COL_1O + 3 * (i - NS); n = 3; }
      else if (i < SS.fan1o) { col = COL_1E + 3 * (i - NS - SS.n1o); n = 3; }
    } else if (KIND == VK_1E) {
      if (i < SS.n1o) { col = COL_1O + 3 * i; n = 3; } else if (i < SS.n1o + SS.n1e) { col = COL_1E + 3 * (i - SS.n1o); n = 3; }
      else if (i < S_OWN1E) { col = COL_0O + (i - SS.n1o - SS.n1e); n = 1; }
    } else {
      if (i < SS.n1e) { col = COL_1E + 3 * i; n = 3; } else if (i < SS.fan0o) { col = COL_0O + (i - SS.n1e); n = 1; }
      else if (i - SS.fan0o < SS.fan1e - S_OWN1E) { col = COL_0O + (S_OWN1E + (i - SS.fan0o) - SS.n1o - SS.n1e); n = 1; }      // guests: block 1e's tail mids
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < n) r[q][c] = xc[(col + c) * 32];
  }
}

template <int KIND, int T>
__device__ __forceinline__ void v_epi(int qs, const f32x16& a, const RawT& r, const float (&v)[3], VOut& o) {
#pragma unroll
  for (int q = 0; q < VEC_TILE_I; ++q) {
    if (q != qs) continue;
    const int i = VEC_TILE_I * T + q;
    // what the slot contributes to: 0 nothing, 1 scalar sum (direction applied once per block), 2 vector (copy), 3 vector (cross with the
    // direction), 4 pseudoscalar from a dot with the direction, 5 pseudoscalar copy, 6 guest of block 1e (scalar sum)
    int what = 0;
    if (KIND == VK_1O) what = i < NS ? 1 : i < NS + SS.n1o ? 2 : i < SS.fan1o ? 3 : 0;
    else if (KIND == VK_1E) what = i < SS.n1o ? 3 : i < SS.n1o + SS.n1e ? 2 : i < S_OWN1E ? 1 : 0;
    else what = i < SS.n1e ? 4 : i < SS.fan0o ? 5 : (i - SS.fan0o < SS.fan1e - S_OWN1E) ? 6 : 0;
    float m[3] = {r[q][0], r[q][1], r[q][2]};
    if (what == 3) {
      const float a0 = r[q][0], a1 = r[q][1], a2 = r[q][2];
      m[0] = a1 * v[2] - a2 * v[1]; m[1] = a2 * v[0] - a0 * v[2]; m[2] = a0 * v[1] - a1 * v[0];
    }
    if (what == 4) m[0] = r[q][0] * v[0] + r[q][1] * v[1] + r[q][2] * v[2];
#pragma unroll
    for (int oo = 0; oo < 3; ++oo) {
      const float wa = a[3 * q + oo];
      if (what == 1) { if (KIND == VK_1O) o.s1o[oo] = fmaf(m[0], wa, o.s1o[oo]); else o.s1e[oo] = fmaf(m[0], wa, o.s1e[oo]); }
      if (what == 6) o.s1e[oo] = fmaf(m[0], wa, o.s1e[oo]);
      if (what == 4 || what == 5) o.k0o[oo] = fmaf(m[0], wa, o.k0o[oo]);
      if (what == 2 || what == 3) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if (KIND == VK_1O) o.k1o[3 * oo + c] = fmaf(m[c], wa, o.k1o[3 * oo + c]); else o.k1e[3 * oo + c] = fmaf(m[c], wa, o.k1e[3 * oo + c]);
        }
      }
    }
  }
}

// one step: raw operands of THIS tile from LDS, its chain, and between the MFMAs the previous step's epilogue + the next step's bias rows
template <int KIND, int T, int PKIND, int PT>
__device__ __forceinline__ void v_step(const bf16x8 (&w)[V2_NFRAG], const Act6& h, f32x16& cb, const float* bias_lds, const int next_bias, const int hf,
                                       const float* xc, f32x16& acc_cur, const f32x16& acc_prev, RawT& raw_cur, const RawT& raw_prev,
                                       const float (&v)[3], VOut& o) {
  v_raw<KIND, T>(xc, raw_cur);
  s_chain<false>(w, h, cb, acc_cur, [&](int q) __attribute__((always_inline)) {
    if (q == 1 && next_bias >= 0) s_load_bias(bias_lds, next_bias, hf, cb);
    if constexpr (PKIND != 0) { if (q < VEC_TILE_I) v_epi<PKIND, PT>(q, acc_prev, raw_prev, v, o); }
  });
}
template <int KIND, int T>
__device__ __forceinline__ void v_drain(const f32x16& acc, const RawT& raw, const float (&v)[3], VOut& o) {
#pragma unroll
  for (int q = 0; q < VEC_TILE_I; ++q) v_epi<KIND, T>(q, acc, raw, v, o);
}

// ---- the program of wave W over the units [u0, u0 + n) of one group entry ------------------------------------------------------
// Pipeline over tau = 0 .. n + 4 (one barrier per iteration):  unit tau: edge indices; tau-1: gathers issued at the start of the iteration,
// written to LDS at its end (X rows, first-Linear input);  tau-2: first Linear (wave 3) -> H;  tau-3: waves 0..2 tiles -> partials P;
// tau-4: wave 3 tiles -> message tile O;  tau-5: wave 2 reduction -> global memory.
template <int W, int NT, int DIAG>
__device__ __forceinline__ void s_segment(const ConvGroup& G, const int cnt, const int u0, const int n, float* const lds,
                                          const bf16x8 (&wt)[NT][V2_NFRAG], const bf16x8 (&ab0e)[3], const int lane,
                                          unsigned long long (&clk)[6]) {
  const int j = lane & 31, hf = lane >> 5;
  const int q4 = 2 * W + hf;                       // this lane's 16-byte granule (columns 4 q4 ..) of the 32-column segments
  const float* const bias_lds = lds + L_BIAS;
  int i_dst = 0, i_src = 0, i_attr = 0;            // indices of edge j of the unit whose gathers are issued next
  f32x4 gx[3], gf[2];                              // gathers in flight: X-row granules q4, q4 + 8, q4 + 16; attr / x_src granule q4
  float vn[3] = {0.f, 0.f, 0.f};                   // edge direction of the unit this wave's tiles process next iteration
  int red_src = -1;

#pragma unroll 1
  for (int tau = 0; tau < n + 5; ++tau) {
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    if constexpr (DIAG == 4) c0 = stamp();
    // ================= gathers of unit tau - 1: issue
    const int ug = tau - 1;
    const bool g_on = ug >= 0 && ug < n;
    if (g_on) {
      const float* rowd = G.node_in + (size_t)i_dst * NODE_STRIDE;
      gx[0] = *reinterpret_cast<const f32x4*>(rowd + 4 * q4);
      gx[1] = *reinterpret_cast<const f32x4*>(rowd + 4 * (q4 + 8));
      if constexpr (W <= 1) gx[2] = *reinterpret_cast<const f32x4*>(rowd + 4 * (W == 0 ? q4 + 16 : 18));
      gf[0] = *reinterpret_cast<const f32x4*>(G.attr + (size_t)i_attr * 32 + 4 * q4);
      gf[1] = *reinterpret_cast<const f32x4*>(G.node_in + (size_t)i_src * NODE_STRIDE + 4 * q4);
    }
    // ================= edge indices of unit tau (used at the start of the next iteration)
    if (tau < n) {
      const int e = (u0 + tau) * SU + j;
      const int ec = e < cnt ? e : cnt - 1;
      i_dst = G.dst[ec]; i_src = G.src[ec]; i_attr = G.attr_idx[ec];
    }
    // edge directions / aggregating nodes for the stages of the NEXT iteration
    float v[3] = {vn[0], vn[1], vn[2]};
    if constexpr (W >= 2) {
      const int un = tau + 1 - (W == 2 ? 3 : 4);
      if (un >= 0 && un < n) {
        const int e = (u0 + un) * SU + j;
        const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[e < cnt ? e : cnt - 1];
        vn[0] = vv.x; vn[1] = vv.y; vn[2] = vv.z;
      }
    }
    int s_red = red_src;
    if constexpr (W == RED_WAVE) {
      const int un = tau + 1 - 5;
      if (un >= 0 && un < n) {
        const int e = (u0 + un) * SU + j;
        red_src = G.src[e < cnt ? e : cnt - 1];        // lanes past the end of the group are masked at the use (no wait on the load here)
      }
    }
    if constexpr (DIAG == 4) c1 = stamp();

    // ================= waves 0 .. 2: hidden tile W of the first Linear of unit tau - 2 -> H (h = ReLU(W1 x + b1) as the bf16 B operand)
    if constexpr (W <= 2) {
      const int uf = tau - 2;
      if (uf >= 0 && uf < n) {
        Act6 Bx;
        s_load_act(lds + L_BX + (uf & 1) * L_FRAG_SLOT, lane, Bx);
        bf16x8* const hs = reinterpret_cast<bf16x8*>(lds + L_H + (uf & 3) * L_FRAG_SLOT) + lane;
        const bf16x8* const fw = reinterpret_cast<const bf16x8*>(lds + L_FLW) + W * V2_TILE_FRAGS + lane;
        bf16x8 fa[V2_NFRAG];
#pragma unroll
        for (int q = 0; q < V2_NFRAG; ++q) fa[q] = fw[q * 64];
        f32x16 cb, fac;
        s_load_bias(bias_lds, W, hf, cb);
        s_chain<false>(fa, Bx, cb, fac, [](int) {});
        bf16x8 h0, h1;
#pragma unroll
        for (int r = 0; r < 8; ++r) { h0[r] = (__bf16)relu1(fac[r]); h1[r] = (__bf16)relu1(fac[8 + r]); }
        hs[(2 * W) * 64] = h0;
        hs[(2 * W + 1) * 64] = h1;
      }
    }
    if constexpr (DIAG == 4) c2 = stamp();

    // ================= tiles
    const int ut = tau - (W == 3 ? 4 : 3);
    if (ut >= 0 && ut < n) {
      const float* const xc = lds + L_X + (ut & 3) * L_X_SLOT + j;
      Act6 h;
      s_load_act(lds + L_H + (ut & 3) * L_FRAG_SLOT, lane, h);
      float* const pw = lds + L_P + (ut & 1) * L_P_SLOT;
      f32x16 o0e;

      if constexpr (W <= 2) {
        // ---- 0e tiles [I_LO, I_LO + N0E): tile = one mid index x 32 output scalars
        constexpr int I_LO = 14 * W, N0E = W == 2 ? 10 : 14;
        // The 0e tiles carry no bias: sum_i m_i (w_i + b_i) = sum_i m_i w_i + sum_i b_i m_i, and the second sum over ALL 38 mids is one
        // [32 x 48] . [48 x 32] bf16 product (tp_conv_bf16.hip) -- k-steps 0 and 1 (the 32 scalar mids) start wave 1's partial sums,
        // k-step 2 (the six dot mids, which need the edge direction) wave 2's
        {
          const f32x16 zero = {};
          o0e = zero;
        }
        if constexpr (W == 1) {
#pragma unroll
          for (int s3 = 0; s3 < 2; ++s3) {
            bf16x8 bm;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) bm[jj] = (__bf16)xc[(16 * s3 + 8 * hf + jj) * 32];
            o0e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab0e[s3], bm, o0e, 0, 0, 0);
          }
        }
        if constexpr (W == 2) {
          bf16x8 bm;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            float m = 0.f;
            if (jj < SS.n1o) {
              const float* p = xc + (COL_1O + 3 * jj) * 32;
              m = p[0] * v[0] + p[32] * v[1] + p[64] * v[2];
            }
            if (hf) m = 0.f;
            bm[jj] = (__bf16)m;
          }
          o0e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab0e[2], bm, o0e, 0, 0, 0);
        }
        auto mid_of = [&](int i) __attribute__((always_inline)) -> float {
          if (i < NS) return xc[i * 32];
          const float* p = xc + (COL_1O + 3 * (i - NS)) * 32;
          return p[0] * v[0] + p[32] * v[1] + p[64] * v[2];
        };
        // pairs of tiles as two interleaved chains; the epilogue of pair p (16 FMAs per tile) sits behind the 12 MFMAs of pair p + 1:
        // slices 0 .. 5 the first tile of the pair, 6 .. 11 the second
        static_assert(N0E % 2 == 0, "0e tiles are processed in pairs");
        f32x16 acc4[4];
        float mid4[4];
        mid4[0] = mid_of(I_LO); mid4[1] = mid_of(I_LO + 1);
        s_chain2(wt[0], wt[1], h, acc4[0], acc4[1], [](int) {});
#pragma unroll
        for (int p = 1; p < N0E / 2; ++p) {
          const int cur = 2 * (p & 1), prv = 2 * ((p - 1) & 1);
          mid4[cur] = mid_of(I_LO + 2 * p); mid4[cur + 1] = mid_of(I_LO + 2 * p + 1);
          const f32x16& ya = acc4[prv];
          const f32x16& yb = acc4[prv + 1];
          const float ma = mid4[prv], mb = mid4[prv + 1];
          s_chain2(wt[2 * p], wt[2 * p + 1], h, acc4[cur], acc4[cur + 1], [&](int sl) __attribute__((always_inline)) {
            const int q = sl < 6 ? sl : sl - 6;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              if (r >= S_EPI_LO[q] && r < S_EPI_LO[q + 1]) { if (sl < 6) o0e[r] = fmaf(ma, ya[r], o0e[r]); else o0e[r] = fmaf(mb, yb[r], o0e[r]); }
          });
        }
        {
          constexpr int prv = 2 * ((N0E / 2 - 1) & 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) o0e[r] = fmaf(mid4[prv], acc4[prv][r], o0e[r]);
#pragma unroll
          for (int r = 0; r < 16; ++r) o0e[r] = fmaf(mid4[prv + 1], acc4[prv + 1][r], o0e[r]);
        }
        VOut vo;
        vo.s1o[0] = vo.s1o[1] = vo.s1o[2] = 0.f;
        if constexpr (W == 2) {
          // ---- block 1o, tiles 0 .. 3: mids 0 .. 19 are (scalar feature) x (edge direction): sum_i x_i w_io, the direction is applied by wave 3
          f32x16 cb, a2[2];
          RawT raw2[2];
          s_load_bias(bias_lds, S_T1O, hf, cb);
          v_step<VK_1O, 0, 0, 0>(wt[N0E + 0], h, cb, bias_lds, S_T1O + 1, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
          v_step<VK_1O, 1, VK_1O, 0>(wt[N0E + 1], h, cb, bias_lds, S_T1O + 2, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
          v_step<VK_1O, 2, VK_1O, 1>(wt[N0E + 2], h, cb, bias_lds, S_T1O + 3, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
          v_step<VK_1O, 3, VK_1O, 2>(wt[N0E + 3], h, cb, bias_lds, -1, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
          v_drain<VK_1O, 3>(a2[1], raw2[1], v, vo);
        }
        // ---- partial sums -> LDS (lane-linear float4 rows)
        f32x4* const p4 = reinterpret_cast<f32x4*>(pw + W * L_P_WAVE) + lane;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) p4[qq * 64] = f32x4{o0e[4 * qq], o0e[4 * qq + 1], o0e[4 * qq + 2], o0e[4 * qq + 3]};
        if constexpr (W == 2) p4[4 * 64] = f32x4{vo.s1o[0], vo.s1o[1], vo.s1o[2], 0.f};
      } else {
        // ================= wave 3: sum of the partials, remaining vector tiles, message tile
        {
          const f32x4* const p4 = reinterpret_cast<const f32x4*>(pw) + lane;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const f32x4 a = p4[qq * 64], b = p4[(L_P_WAVE / 4) + qq * 64], c = p4[2 * (L_P_WAVE / 4) + qq * 64];
            o0e[4 * qq + 0] = (a.x + b.x) + c.x; o0e[4 * qq + 1] = (a.y + b.y) + c.y;
            o0e[4 * qq + 2] = (a.z + b.z) + c.z; o0e[4 * qq + 3] = (a.w + b.w) + c.w;
          }
        }
        const f32x4 sp = (reinterpret_cast<const f32x4*>(pw) + lane)[2 * (L_P_WAVE / 4) + 4 * 64];
        VOut vo;
        vo.s1o[0] = sp.x; vo.s1o[1] = sp.y; vo.s1o[2] = sp.z;
        vo.s1e[0] = vo.s1e[1] = vo.s1e[2] = 0.f;
        vo.k0o[0] = vo.k0o[1] = vo.k0o[2] = 0.f;
#pragma unroll
        for (int r = 0; r < 9; ++r) { vo.k1o[r] = 0.f; vo.k1e[r] = 0.f; }
        f32x16 cb, a2[2];
        RawT raw2[2];
        s_load_bias(bias_lds, S_T1O + 4, hf, cb);
        // block 1o tiles 4 .. 8 (mids 20 .. 43 + one padded slot), block 1e's own three tiles, block 0o's three (with 1e's tail mids as guests)
        v_step<VK_1O, 4, 0, 0>(wt[0], h, cb, bias_lds, S_T1O + 5, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
        v_step<VK_1O, 5, VK_1O, 4>(wt[1], h, cb, bias_lds, S_T1O + 6, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
        v_step<VK_1O, 6, VK_1O, 5>(wt[2], h, cb, bias_lds, S_T1O + 7, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
        v_step<VK_1O, 7, VK_1O, 6>(wt[3], h, cb, bias_lds, S_T1O + 8, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
        v_step<VK_1O, 8, VK_1O, 7>(wt[4], h, cb, bias_lds, S_T1E + 0, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
        v_step<VK_1E, 0, VK_1O, 8>(wt[5], h, cb, bias_lds, S_T1E + 1, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
        v_step<VK_1E, 1, VK_1E, 0>(wt[6], h, cb, bias_lds, S_T1E + 2, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
        v_step<VK_1E, 2, VK_1E, 1>(wt[7], h, cb, bias_lds, S_T0O + 0, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
        v_step<VK_0O, 0, VK_1E, 2>(wt[8], h, cb, bias_lds, S_T0O + 1, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
        v_step<VK_0O, 1, VK_0O, 0>(wt[9], h, cb, bias_lds, S_T0O + 2, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
        v_step<VK_0O, 2, VK_0O, 1>(wt[10], h, cb, bias_lds, -1, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
        v_drain<VK_0O, 2>(a2[0], raw2[0], v, vo);
        float k1o[9], k1e[9], k0o[3];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          k0o[o] = vo.k0o[o];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            k1o[3 * o + c] = fmaf(v[c], vo.s1o[o], vo.k1o[3 * o + c]);
            k1e[3 * o + c] = fmaf(v[c], vo.s1e[o], vo.k1e[3 * o + c]);
          }
        }
        // ---- message tile [col][36]
        float* const om = lds + L_O + (ut & 1) * L_O_SLOT;
#pragma unroll
        for (int r = 0; r < 16; ++r) om[((r & 3) + 8 * (r >> 2) + 4 * hf) * S_OSTR + j] = o0e[r];
#pragma unroll
        for (int o = 0; o < 3; ++o) {
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            om[(COL_1O + 3 * (3 * hf + o) + c) * S_OSTR + j] = k1o[3 * o + c];
            om[(COL_1E + 3 * (3 * hf + o) + c) * S_OSTR + j] = k1e[3 * o + c];
          }
          om[(COL_0O + 3 * hf + o) * S_OSTR + j] = k0o[o];
        }
      }
    }
    if constexpr (DIAG == 4) c3 = stamp();

    // ================= reduction wave: run-length sums of unit tau - 5 -> global memory
    if constexpr (W == RED_WAVE) {
      const int ur = tau - 5;
      if (ur >= 0 && ur < n) {
        const float* const om = lds + L_O + (ur & 1) * L_O_SLOT;
        const size_t tile = (size_t)(u0 + ur);
        const int sm = (u0 + ur) * SU + j < cnt ? s_red : -1;      // lanes past the end of the group
        s_reduce_runs<NODE_STRIDE>(om, sm, lane, G.first_sum + tile * NODE_STRIDE, G.last_sum + tile * NODE_STRIDE, G.run_acc);
      }
    }
    if constexpr (DIAG == 4) c4 = stamp();

    // ================= gathers of unit tau - 1: write to LDS
    if (g_on) {
      float* const X = lds + L_X + (ug & 3) * L_X_SLOT;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k == 2 && W > 1) continue;
        const int gran = k < 2 ? q4 + 8 * k : (W == 0 ? q4 + 16 : 18);
        if (k == 2 && W == 1 && hf) continue;
        const f32x4 r = gx[k];
        float* o = X + (4 * gran) * 32 + j;
        o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
      }
      // first-Linear input: granule q4 of part seg (0 edge_attr, 1 x_src, 2 x_dst) -> k-step 2 seg + (q >> 1), elements 4 (q & 1) .. + 3 of
      // lane (j, hb), where q4 = 4 hb + q (the layout of v2_set_in, tp_conv_bf16_dev.h)
      __bf16* const bx = reinterpret_cast<__bf16*>(lds + L_BX + (ug & 1) * L_FRAG_SLOT);
      const int hb = q4 >> 2, q = q4 & 3;
#pragma unroll
      for (int seg = 0; seg < 3; ++seg) {
        const f32x4 x = seg == 0 ? gf[0] : seg == 1 ? gf[1] : gx[0];
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 pk;
        pk[0] = (__bf16)x.x; pk[1] = (__bf16)x.y; pk[2] = (__bf16)x.z; pk[3] = (__bf16)x.w;
        *reinterpret_cast<bf16x4*>(bx + (((2 * seg + (q >> 1)) * 64 + 32 * hb + j) * 8 + 4 * (q & 1))) = pk;
      }
    }
    unsigned long long c5 = 0;
    if constexpr (DIAG == 4) c5 = stamp();
    lds_barrier();
    if constexpr (DIAG == 4) {
      const unsigned long long c6 = stamp();
      clk[0] += c1 - c0; clk[1] += c2 - c1; clk[2] += c3 - c2; clk[3] += c4 - c3; clk[4] += c5 - c4; clk[5] += c6 - c5;
    }
  }
}

template <int W, int DIAG>
__device__ __forceinline__ void s_wave_prog(const ConvArgs& args, float* const lds, const float* const wstream, const int lane,
                                            const int u_lo, const int u_hi, const int mine, const int incl) {
  constexpr int T0 = W == 0 ? 3 : W == 1 ? 17 : W == 2 ? 31 : 3 + SS.t0e + 4;
  constexpr int NT = W == 3 ? 11 : 14;
  const GFrag gp = (GFrag)reinterpret_cast<const bf16x8*>(wstream);
  bf16x8 wt[NT][V2_NFRAG];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < V2_NFRAG; ++q) {
      wt[t][q] = gp[(size_t)(T0 + t) * V2_TILE_FRAGS + q * 64 + lane];
      // ten tiles live in the accumulation half of the register file (256 registers; the MFMA reads srcA from it directly), the rest
      // in ordinary VGPRs -- pinning more than fit makes hipcc copy the overflow in front of every MFMA that uses it
      if (t < 10) asm volatile("" : "+a"(wt[t][q])); else asm volatile("" : "+v"(wt[t][q]));
    }
  bf16x8 ab0e[3];
  if constexpr (W == 1 || W == 2) {
    const GFrag gb0e = (GFrag)reinterpret_cast<const bf16x8*>(reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(wstream) + (size_t)(SS.ntiles + 1) * V2_TILE_FRAGS) + (size_t)(SS.ntiles + 1) * 32);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) ab0e[s3] = gb0e[s3 * 64 + lane];
  }
  unsigned long long clk[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long st0 = 0, sr0 = 0;
  if constexpr (DIAG == 4) { st0 = stamp(); sr0 = __builtin_amdgcn_s_memrealtime(); }
#pragma unroll 1
  for (int g = 0; g < args.n_groups; ++g) {
    const int mg = __builtin_amdgcn_readlane(mine, g);
    if (mg == 0) continue;
    const int end = __builtin_amdgcn_readlane(incl, g), start = end - mg;
    const int a = u_lo > start ? u_lo : start, b = u_hi < end ? u_hi : end;
    if (a >= b) continue;
    const ConvGroup G = args.g[g];
    const int cnt = *G.count;
    s_segment<W, NT, DIAG>(G, cnt, a - start, b - a, lds, wt, ab0e, lane, clk);
  }
  if constexpr (DIAG == 4) {
    const int rec = blockIdx.x * SW_WAVES + W;
    if (lane == 0 && args.stamps && rec < 8192 / 2) {
      unsigned long long* o = args.stamps + (size_t)rec * 16;
      o[0] = st0; o[1] = sr0; o[2] = stamp(); o[3] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
      for (int k = 0; k < 6; ++k) o[4 + k] = clk[k];
      o[10] = (unsigned long long)(u_hi - u_lo);
    }
  }
}

template <int DIAG = 0>
__global__ __launch_bounds__(SW_WAVES * 64, 1) void tp_conv64s_kernel(ConvArgs args, RoleTableS rt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

  // ---- roles: 32-edge units per entry (lane g <-> entry g), workgroups per role, this workgroup's role and unit range
  int units = 0, my_role = -1;
  if (lane < args.n_groups) {
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const int* cp = *reinterpret_cast<const int* const*>(ka + offsetof(ConvArgs, g) + (size_t)lane * sizeof(ConvGroup) + offsetof(ConvGroup, count));
    units = (*cp + SU - 1) / SU;
    my_role = *reinterpret_cast<const unsigned char*>(ka + ((sizeof(ConvArgs) + alignof(RoleTableS) - 1) / alignof(RoleTableS)) * alignof(RoleTableS) + offsetof(RoleTableS, role_of) + lane);
  }
  const int total = s_wave_sum(units);
  if (total == 0) return;
  const int n_wg = gridDim.x;
  int role = -1, rank = 0, n_role_wg = 1, role_units = 0, acc_units = 0, wg_lo = 0;
#pragma unroll 1
  for (int r = 0; r < rt.n_roles; ++r) {
    const int w = s_wave_sum(my_role == r ? units : 0);
    acc_units += w;
    int wg_hi = (int)((long long)n_wg * acc_units / total);
    if (w > 0 && wg_hi <= wg_lo) wg_hi = wg_lo + 1;                 // every role with work gets a workgroup
    if (r == rt.n_roles - 1 || wg_hi > n_wg) wg_hi = n_wg;
    if (role < 0 && (int)blockIdx.x >= wg_lo && (int)blockIdx.x < wg_hi && w > 0) { role = r; rank = blockIdx.x - wg_lo; n_role_wg = wg_hi - wg_lo; role_units = w; }
    wg_lo = wg_hi;
  }
  if (role < 0) return;
  const int u_lo = (int)((long long)role_units * rank / n_role_wg), u_hi = (int)((long long)role_units * (rank + 1) / n_role_wg);
  if (u_lo >= u_hi) return;
  const float* const wstream = rt.wstream[role];

  // ---- bias rows and the three first-Linear tiles -> LDS, once
  {
    const float* gb = reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(wstream) + (size_t)(SS.ntiles + 1) * V2_TILE_FRAGS);
    for (int k = threadIdx.x; k < (SS.ntiles + 1) * 32; k += SW_WAVES * 64) lds[L_BIAS + k] = gb[k];
    const f32x4* src = reinterpret_cast<const f32x4*>(wstream);
    f32x4* dst = reinterpret_cast<f32x4*>(lds + L_FLW);
    for (int k = threadIdx.x; k < 3 * V2_TILE_FRAGS; k += SW_WAVES * 64) dst[k] = src[k];
  }
  __syncthreads();

  // units of this role per entry and their inclusive prefix (unit -> entry)
  const int mine = my_role == role ? units : 0;
  int incl = mine;
#pragma unroll
  for (int d = 1; d < CONV_MAX_GROUPS; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if (lane >= d) incl += v;
  }
  if (wave == 0) s_wave_prog<0, DIAG>(args, lds, wstream, lane, u_lo, u_hi, mine, incl);
  else if (wave == 1) s_wave_prog<1, DIAG>(args, lds, wstream, lane, u_lo, u_hi, mine, incl);
  else if (wave == 2) s_wave_prog<2, DIAG>(args, lds, wstream, lane, u_lo, u_hi, mine, incl);
  else s_wave_prog<3, DIAG>(args, lds, wstream, lane, u_lo, u_hi, mine, incl);
}

// a: the edge groups of a 74 -> 74 layer (whole tile chains); n_wg: workgroups (<= CUs).  Roles = distinct weight streams.
hipError_t launch_tp_conv_bf16s(const ConvArgs& a, int n_wg, hipStream_t s) {
  if (a.n_groups <= 0) return hipSuccess;
  RoleTableS rt{};
  for (int g = 0; g < a.n_groups; ++g) {
    const ConvGroup& G = a.g[g];
    if (!G.vec_on || G.i0e_lo != 0 || G.i0e_hi != SS.t0e) return hipErrorInvalidValue;      // no virtual slices here
    int r = -1;
    for (int k = 0; k < rt.n_roles; ++k)
      if (rt.wstream[k] == G.wstream) r = k;
    if (r < 0) {
      if (rt.n_roles == S_MAX_ROLES) return hipErrorInvalidValue;
      r = rt.n_roles++;
      rt.wstream[r] = G.wstream;
    }
    rt.role_of[g] = (unsigned char)r;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64s_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64s_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  static const int diag = getenv("CBD_BF16_DIAG") ? atoi(getenv("CBD_BF16_DIAG")) : 0;
  if (diag == 4) hipLaunchKernelGGL((tp_conv64s_kernel<4>), dim3(n_wg), dim3(SW_WAVES * 64), S_LDS_BYTES, s, a, rt);
  else hipLaunchKernelGGL((tp_conv64s_kernel<0>), dim3(n_wg), dim3(SW_WAVES * 64), S_LDS_BYTES, s, a, rt);
  return hipGetLastError();
}

}  // namespace cbd
