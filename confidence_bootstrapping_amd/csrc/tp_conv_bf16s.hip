// bf16 tensor-product message passing with REGISTER-STATIONARY weights (BASELINE.json configs[3]; options bf16 + bf16_stationary).
//
// Same math as tp_conv_bf16.hip (FCBlock -> FasterTensorProduct -> segmented sum; reference models/tensor_layers.py:195-206, 66-117) for
// the 74 -> 74 layers, organised the other way round.  The streaming kernel gives every wave 64 edges and makes it read its group's whole
// FCBlock (56 tiles x 6 KB = 336 KB) per 64 edges: at the bf16 MFMA rate that is the CU's whole vector-memory return path (64 B/clk), so
// the matrix pipe and that path saturate together at ~0.5 (PMC round 4), and a quarter of every wave's lifetime (index -> gather round
// trips, first Linear, reduction) holds no MFMA at all.  Sharing tiles through LDS lost three times to the per-tile rendezvous, the role
// split with LDS-resident tiles to the per-slice repetition of those fixed phases.  Here
//   * a PERSISTENT workgroup of four waves (one per SIMD, up to 512 registers each) is bound to one FCBlock and keeps its second Linear
//     on the CU for the whole launch: 49 of the 53 tiles in registers (13 / 12 / 13 / 11 per wave x 24 registers; ten per wave in the
//     accumulation half of the register file, read by the MFMA as srcA directly -- the accumulators are forced into VGPRs for that,
//     -mllvm --amdgpu-mfma-vgpr-form=1), four tiles and the three first-Linear tiles in LDS.  No weight is fetched per edge any more;
//     the vector-memory path carries only the gathers (0.7 KB per edge instead of 5.9);
//   * every wave processes ALL 32 edges of a unit, but only its own tiles: the fixed phases are paid once per unit by the workgroup,
//     not once per wave or slice.  The unit's gathered destination rows (mids) and its hidden activations h (the B operand) are shared
//     through LDS; ONE barrier per unit;
//   * waves 0..2 (0e mids [0,14), [14,28), [28,38) + the scalar-mid tiles of block 1o) write partial output sums to LDS; wave 3 works
//     one unit behind them: it adds the three partials, runs the remaining vector tiles as software-pipelined steps, finishes the
//     message tile.  Waves 0..2 also compute one hidden tile each of the first Linear one unit AHEAD (its chain interleaved into their
//     first tile's), waves 0 / 1 half of the run-length reduction each two units behind, and all four waves share the gathers three
//     units ahead -- a software pipeline over units, every stage double- or quad-buffered in LDS;
//   * inside waves 0..2 the chains of consecutive tiles are staggered by half a tile (three accumulators: no MFMA waits for its
//     predecessor) and the CG epilogue of tile k - 2, the first-Linear chain, the gather issue and the LDS writes sit in the issue slots
//     between the MFMAs -- a wave that is alone on its SIMD has nobody else to fill its MFMA shadows;
//   * the units of a launch, in role-major order, are cut into equal pieces per workgroup (a workgroup may straddle a role boundary and
//     then reloads its weights once): dealing whole workgroups to roles cost up to 2x (profiles/r05_a_bf16_stationary_experiment.txt).
// Measured (C4 64 x 40): 2 .. 4 % faster than the streaming kernel; what bounds it: DESIGN.md section 5 and the profile file above.
// Results: the same pieces (first_sum / last_sum / run_acc per 32-edge tile) as the streaming kernel; the order in which a message's
// tile contributions are added differs (partials per wave), so the two kernels agree to fp32 rounding of those sums, not bitwise.
// Deterministic: no atomics, fixed orders.
#include <atomic>
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
constexpr int RED_COLS = (NODE_DIM + 1) / 2;               // waves 0 and 1 reduce the message columns [0, 37) and [37, 74)

// ---- LDS map (floats)
constexpr int L_X_SLOT = 76 * 32;                        // gathered destination rows, transposed [col][32 edges]
constexpr int L_X = 0;                                   // [4 slots]
constexpr int L_FRAG_SLOT = V2_NFRAG * 64 * 4;           // one B operand: [6 k-steps][64 lanes][8 bf16] = 6 KB
constexpr int L_BX = L_X + 4 * L_X_SLOT;                 // [2 slots] first-Linear input (edge_attr | x_src[:32] | x_dst[:32]) as bf16 B operand
constexpr int L_H = L_BX + 2 * L_FRAG_SLOT;              // [3 slots] hidden activations h as bf16 B operand (slot = unit mod 3)
constexpr int L_P_WAVE = 16 * 64;                        // partial sums of one wave: 16 (0e) registers x 64 lanes; wave 2: + one float4 row (1o scalar sums)
constexpr int L_P_SLOT = 3 * L_P_WAVE + 4 * 64;
constexpr int L_P = L_H + 3 * L_FRAG_SLOT;               // [2 slots][3 waves]
constexpr int S_OSTR = 36;                               // message tile stride: lane = column reads four edges per ds_read_b128 without bank conflicts
constexpr int L_O_SLOT = NODE_DIM * S_OSTR;              // message tile [74][36]
constexpr int L_O = L_P + 2 * L_P_SLOT;                  // [2 slots]
constexpr int S_BIAS_ROWS = 3 + SS.t1o + SS.t1e - 1 + SS.t0o;   // the tiles that take a bias as their C operand: first Linear, blocks 1o / 1e / 0o
constexpr int L_BIAS = L_O + 2 * L_O_SLOT;               // [18][32] fp32 bias rows (bias_row())
constexpr int L_FLW = L_BIAS + S_BIAS_ROWS * 32;         // the three first-Linear tiles
constexpr int S_LDS_TILES = 4;                           // second-Linear tiles read from LDS instead of held in registers: the last 0e tile of waves 0 and 2, the last two of wave 1
constexpr int L_WT = L_FLW + 3 * L_FRAG_SLOT;
constexpr int L_TOTAL = L_WT + S_LDS_TILES * L_FRAG_SLOT;
__host__ __device__ constexpr int lds_tile_stream(int k) { return k == 0 ? 16 : k == 1 ? 23 : k == 2 ? 30 : 40; }      // stream index of LDS tile k
__host__ __device__ constexpr int bias_row(int T) { return T < 3 ? T : 3 + (T - (3 + SS.t0e)); }                        // row of stream tile T in the LDS bias table
constexpr int S_LDS_BYTES = L_TOTAL * 4;
static_assert(S_LDS_BYTES <= 160 * 1024, "one workgroup per CU");
static_assert(L_BX % 4 == 0 && L_H % 4 == 0 && L_P % 4 == 0 && L_O % 4 == 0 && L_BIAS % 4 == 0 && L_FLW % 4 == 0 && L_WT % 4 == 0, "16-byte alignment");
__device__ __forceinline__ int h_slot(int u) { return (u + 9) % 3; }      // u >= -9

struct RoleTableS {
  int n_roles;
  unsigned char role_of[CONV_MAX_GROUPS];      // role of every entry of ConvArgs::g
  unsigned char weight[S_MAX_ROLES];           // cost of one unit of the role in 1/64 (ConvGroup::cost_w of its groups)
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

// bias rows of tile T as the C operand (accumulator layout: register r of lane half hf = row (r & 3) + 8 (r >> 2) + 4 hf)
__device__ __forceinline__ void s_load_bias(const float* bias_lds, int T, int hf, f32x16& cb) {
  const f32x4* b4 = reinterpret_cast<const f32x4*>(bias_lds + bias_row(T) * 32);
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

// Run-length sums of one message tile for the columns [C_LO, C_LO + C_N) (reduce_runs.h: segmented sums without atomics, a run that
// touches the tile's first / last edge goes to first_sum / last_sum, any other to run_acc[node]); lane l sums column C_LO + l, waves 0 and
// 1 each take half of the 74 columns.  The branch-free bulk is taken out: `gs[g]` = the sum of the eight edges of group g of this lane's
// column, formed between the MFMAs of the tile loop (two ds_read_b128 + seven adds per group, no control flow; the aggregating-node ids
// and the boundary mask come in registers).  Here only the bookkeeping is left: a
// group without a run boundary adds its sum; a group WITH one (one or two of four in a tile of two or three runs) is re-read and walked
// edge by edge.  (A run's sum is therefore associated by groups of eight -- deterministic, not the association of reduce_runs.)
template <int NODE_STR, int C_LO, int C_N>
__device__ __forceinline__ void s_reduce_groups(const float* __restrict__ msg, const float (&gs)[4], const int s_me, const unsigned starts,
                                                const int last, int lane, float* __restrict__ fs, float* __restrict__ ls,
                                                float* __restrict__ run_acc) {
  const bool on = lane < C_N;
  const int col = C_LO + (on ? lane : 0);
  const f32x4* oc = reinterpret_cast<const f32x4*>(msg + col * S_OSTR);
  float sum = 0.f;
  int a0 = 0;
#pragma unroll
  for (int g8 = 0; g8 < 4; ++g8) {
    if (((starts >> (8 * g8)) & 0xffu) == 0u) {
      sum += gs[g8];
    } else {
      const f32x4 a = oc[2 * g8], b = oc[2 * g8 + 1];
      const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int jj = 8 * g8 + k;
        if (jj > 0 && ((starts >> jj) & 1u)) {   // run [a0, jj-1] is complete
          const int node = __builtin_amdgcn_readlane(s_me, a0);
          float* dst = a0 == 0 ? fs : run_acc + (size_t)node * NODE_STR;
          if (on) dst[col] = sum;
          sum = 0.f;
          a0 = jj;
        }
        sum += v[k];
      }
    }
  }
  if (last >= 0 && on) (a0 == 0 ? fs : ls)[col] = sum;
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
template <int KIND, int T, int PKIND, int PT, class Side>
__device__ __forceinline__ void v_step(const bf16x8 (&w)[V2_NFRAG], const Act6& h, f32x16& cb, const float* bias_lds, const int next_bias, const int hf,
                                       const float* xc, f32x16& acc_cur, const f32x16& acc_prev, RawT& raw_cur, const RawT& raw_prev,
                                       const float (&v)[3], VOut& o, Side side) {
  v_raw<KIND, T>(xc, raw_cur);
  s_chain<false>(w, h, cb, acc_cur, [&](int q) __attribute__((always_inline)) {
    if (q == 1 && next_bias >= 0) s_load_bias(bias_lds, next_bias, hf, cb);
    if constexpr (PKIND != 0) { if (q < VEC_TILE_I) v_epi<PKIND, PT>(q, acc_prev, raw_prev, v, o); }
    side(q);
  });
}
template <int KIND, int T, int PKIND, int PT>
__device__ __forceinline__ void v_step(const bf16x8 (&w)[V2_NFRAG], const Act6& h, f32x16& cb, const float* bias_lds, const int next_bias, const int hf,
                                       const float* xc, f32x16& acc_cur, const f32x16& acc_prev, RawT& raw_cur, const RawT& raw_prev,
                                       const float (&v)[3], VOut& o) {
  v_step<KIND, T, PKIND, PT>(w, h, cb, bias_lds, next_bias, hf, xc, acc_cur, acc_prev, raw_cur, raw_prev, v, o, [](int) {});
}
template <int KIND, int T>
__device__ __forceinline__ void v_drain(const f32x16& acc, const RawT& raw, const float (&v)[3], VOut& o) {
#pragma unroll
  for (int q = 0; q < VEC_TILE_I; ++q) v_epi<KIND, T>(q, acc, raw, v, o);
}

// Which of a wave's 0e tiles (position k = mid index - I_LO) are read from LDS, and where the others sit in its register array.
// Wave 1 has two LDS tiles; they are kept apart (positions 6 and 13) so that only one LDS tile is in flight in the staggered chains.
template <int W>
struct WMap {
  static constexpr int N0E = W == 2 ? 10 : 14;
  static constexpr int I_LO = 14 * W;
  static constexpr bool is_lds(int k) { return W == 0 ? k == 13 : W == 1 ? (k == 6 || k == 13) : k == 9; }
  static constexpr int ridx(int k) { return W == 1 ? (k < 6 ? k : k - 1) : k; }
  static constexpr int lidx(int k) { return W == 0 ? 0 : W == 1 ? (k == 6 ? 1 : 2) : 3; }
  static constexpr int NREG0E = W == 0 ? 13 : W == 1 ? 12 : 9;      // register-resident 0e tiles; wave 2: wt[NREG0E + t] = block 1o tile t
};

// ---- waves 0 .. 2: one fused, branch-free instruction stream per unit iteration ----------------------------------------------------
// A wave that is alone on its SIMD has nobody to fill its stalls, and every instruction it issues next to an MFMA costs matrix-pipe
// time only if it WAITS there.  So everything that is not a tile -- the first-Linear chain of this wave's hidden tile (one unit ahead),
// the gather issue (two units ahead), the LDS writes of the gathered data, the run-length reduction (wave 0, two units behind) -- is cut
// into slices that sit in the issue slots between the MFMAs of the 0e pairs, their LDS / memory operands requested several slots
// earlier.  Stages whose unit is outside [0, n) run on whatever the LDS slots hold (no branches; the slots they write are not live,
// section "fill / drain" in DESIGN.md) -- only global stores are guarded, global loads use clamped indices.
template <int W, int NT, int DIAG>
__device__ __forceinline__ void s_segment_a(const ConvGroup& G, const int cnt, const int u0, const int n, float* const lds,
                                            const bf16x8 (&wt)[NT][V2_NFRAG], const bf16x8 (&ab0e)[3], const int lane,
                                            unsigned long long (&clk)[6]) {
  static_assert(W <= 2, "waves 0 .. 2");
  constexpr int I_LO = 14 * W;                              // this wave's 0e tiles: mids [I_LO, I_LO + WMap<W>::N0E)
  using M = WMap<W>;
  constexpr int V0 = M::NREG0E;                             // wave 2: wt[V0 + t] = block 1o tile t
  const int j = lane & 31, hf = lane >> 5;
  const int q4 = 2 * W + hf;                       // this lane's 16-byte granule (columns 4 q4 ..) of the 32-column segments
  const float* const bias_lds = lds + L_BIAS;
  const bf16x8* const lw = reinterpret_cast<const bf16x8*>(lds + L_WT) + lane;      // LDS tile t: lw + t * V2_TILE_FRAGS
  const GPtr<float> g_node = (GPtr<float>)G.node_in;
  const GPtr<float> g_attr = (GPtr<float>)G.attr;
  int i_dst = 0, i_src = 0, i_attr = 0;            // indices of edge j of the unit whose gathers are issued next
  float vn[3] = {0.f, 0.f, 0.f};                   // edge direction of the unit this wave's tiles process next iteration (wave 2)
  int red_src = 0, red_prev = 0;                   // aggregating node of edge j (and of edge j - 1) of the unit reduced next iteration (waves 0 / 1)
  auto edge_of = [&](int u) __attribute__((always_inline)) -> int {      // clamped edge index of lane j in unit u (any u)
    int e = (u0 + u) * SU + j;
    e = e < cnt ? e : cnt - 1;
    return e > 0 ? e : 0;
  };

#pragma unroll 1
  for (int tau = 0; tau < n + 5; ++tau) {
    unsigned long long c0 = 0;
    if constexpr (DIAG == 4 || DIAG == 6) c0 = stamp();
    const int ug = tau - 1, uf = tau - 2, ut = tau - 3, ur = tau - 5;
    // ================= LDS operands of this iteration: h of the tiles' unit, k-step 0 of the first-Linear chain
    const float* const xc = lds + L_X + (ut & 3) * L_X_SLOT + j;
    Act6 h;
    s_load_act(lds + L_H + h_slot(ut) * L_FRAG_SLOT, lane, h);
    const bf16x8* const bxp = reinterpret_cast<const bf16x8*>(lds + L_BX + (uf & 1) * L_FRAG_SLOT) + lane;
    const bf16x8* const fwp = reinterpret_cast<const bf16x8*>(lds + L_FLW) + W * V2_TILE_FRAGS + lane;
    bf16x8 fa[2], fb[2];
    fa[0] = fwp[0]; fb[0] = bxp[0];
    f32x16 fac;
    s_load_bias(bias_lds, W, hf, fac);
    float dots[6];
    float v[3] = {vn[0], vn[1], vn[2]};
    auto mid_ld = [&](int i) __attribute__((always_inline)) -> float { return i < NS ? xc[i * 32] : dots[i - NS]; };
    if constexpr (W == 2) {   // the six 1o . direction mids (0e mids 32 .. 37): tiles of this wave and k-step 2 of the bias product
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float* p = xc + (COL_1O + 3 * k) * 32;
        dots[k] = p[0] * v[0] + p[32] * v[1] + p[64] * v[2];
      }
    }
    __builtin_amdgcn_sched_barrier(0);      // the LDS reads above are in flight while the gather addresses below are formed
    // ================= gathers of unit tau - 1: issue (written to LDS between the MFMAs of pair 2)
    f32x4 gx[W == 2 ? 3 : 2], gf[2];
    {
      const GPtr<f32x4> rowd = (GPtr<f32x4>)(g_node + (size_t)i_dst * NODE_STRIDE);
      gx[0] = rowd[q4];
      gx[1] = rowd[q4 + 8];
      if constexpr (W == 2) gx[2] = rowd[q4 + 12];      // granules 16, 17 (wave 3: 18)
      gf[0] = ((GPtr<f32x4>)(g_attr + (size_t)i_attr * 32))[q4];
      gf[1] = ((GPtr<f32x4>)(g_node + (size_t)i_src * NODE_STRIDE))[q4];
    }
    // edge indices of unit tau (used at the start of the next iteration); direction / aggregating node for the next iteration's stages
    {
      const int ec = edge_of(tau);
      i_dst = G.dst[ec]; i_src = G.src[ec]; i_attr = G.attr_idx[ec];
    }
    if constexpr (W == 2) {
      const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[edge_of(ut + 1)];
      vn[0] = vv.x; vn[1] = vv.y; vn[2] = vv.z;
    }
    const bool red_on = W <= 1 && ur >= 0 && ur < n;
    // run boundaries of the unit reduced in this iteration -- the same for every column: found HERE (scalar mask), from the aggregating
    // nodes of edge j and of edge j - 1 loaded during the previous iteration (no LDS shuffle in the tail)
    int r_sme = -1, r_last = -1;
    unsigned r_starts = 0;
    if constexpr (W <= 1) {
      const int e_me = (u0 + ur) * SU + j;
      r_sme = e_me < cnt ? red_src : -1;                     // lanes past the end of the group: -1 (one run that is never stored)
      const int r_prv = e_me - 1 < cnt ? red_prev : -1;
      r_starts = (unsigned)__ballot(lane < 32 && j > 0 && r_sme != r_prv);
      r_last = __builtin_amdgcn_readlane(r_sme, 31);
      const int en = edge_of(ur + 1);
      red_src = G.src[en];
      red_prev = G.src[en > 0 ? en - 1 : 0];
    }

    // reduction of unit tau - 5 (waves 0 / 1): this lane's column of the message tile; its four group sums are formed in the tile loop
    const float* const r_om = lds + L_O + (ur & 1) * L_O_SLOT;
    const f32x4* const r_oc = reinterpret_cast<const f32x4*>(r_om + ((W == 0 ? 0 : RED_COLS) + (lane < (W == 0 ? RED_COLS : NODE_DIM - RED_COLS) ? lane : 0)) * S_OSTR);
    f32x4 r_v[2];
    float r_gs[4] = {0.f, 0.f, 0.f, 0.f};
    // ================= the side work of pair p, slot s (0 .. 11)
    f32x16 o0e;
    {
      const f32x16 zero = {};
      o0e = zero;
    }
    bf16x8 bm, aw[3];
    float bsrc[8];
    f32x16 ob;      // wave 1: accumulator of the bias product of the 0e block's 32 scalar mids (sum_i b_i m_i as two bf16 MFMAs, tp_conv_bf16.hip)
    {
      const f32x16 zero = {};
      ob = zero;
    }
    float* const xw = lds + L_X + (ug & 3) * L_X_SLOT;
    __bf16* const bxw = reinterpret_cast<__bf16*>(lds + L_BX + (ug & 1) * L_FRAG_SLOT);
    bf16x8* const hs = reinterpret_cast<bf16x8*>(lds + L_H + h_slot(uf) * L_FRAG_SLOT) + lane;
    // Side work of half-step hs_ (0 .. N0E), slot sl (0 .. 5).  Half-step s issues MFMAs 3, 4, 5 of tile s - 1 and MFMAs 0, 1, 2 of tile s,
    // alternating (slot 2 i follows MFMA 3 + i of tile s - 1, slot 2 i + 1 MFMA i of tile s): two chains in flight, staggered by half a
    // tile, so no MFMA waits for its predecessor; the epilogue of tile s - 2 -- finished one half-step ago -- fills the same slots.
    auto side = [&](int hs_, int sl) __attribute__((always_inline)) {
      if (hs_ <= 1 && (sl & 1)) {
        // ---- first-Linear chain of hidden tile W (unit tau - 2): MFMA f = 3 hs_ + sl / 2, operands of f + 1 requested in front of it
        const int f = 3 * hs_ + (sl >> 1);
        if (f + 1 < V2_NFRAG) { fa[(f + 1) & 1] = fwp[(f + 1) * 64]; fb[(f + 1) & 1] = bxp[(f + 1) * 64]; }
        fac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[f & 1], fb[f & 1], fac, 0, 0, 0);
      }
      if (hs_ == 2) {
        // ---- hidden tile: ReLU, bf16, store as the B operand of the second Linear
        if (sl == 1 || sl == 3) {
          const int s2 = sl >> 1;
          bf16x8 hh;
#pragma unroll
          for (int r = 0; r < 8; ++r) hh[r] = (__bf16)relu1(fac[8 * s2 + r]);
          hs[(2 * W + s2) * 64] = hh;
        }
      }
      if constexpr (W == 1) {
        // ---- bias product, k-steps 0 and 1: operands from LDS at slot 0, converted at slot 3, MFMA at slot 5 of half-steps 12 and 13
        //      (into an accumulator of its own: the epilogue FMAs of the tiles write o0e in the same half-steps)
        if ((hs_ == 12 || hs_ == 13) && sl == 0) {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) bsrc[jj] = xc[(16 * (hs_ - 12) + 8 * hf + jj) * 32];
        }
        if ((hs_ == 12 || hs_ == 13) && sl == 3) {
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) bm[jj] = (__bf16)bsrc[jj];
        }
        if ((hs_ == 12 || hs_ == 13) && sl == 5) ob = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab0e[hs_ - 12], bm, ob, 0, 0, 0);
      }
      if constexpr (W <= 1) {
        // ---- reduction, branch-free part: group g8 of eight edges is read in half-step 4 + 2 g8 and summed one half-step later
        if (hs_ >= 4 && hs_ <= 11) {
          const int g8 = (hs_ - 4) >> 1;
          if (((hs_ - 4) & 1) == 0 && sl == 0) { r_v[0] = r_oc[2 * g8]; r_v[1] = r_oc[2 * g8 + 1]; }
          if (((hs_ - 4) & 1) == 1 && sl == 2)
            r_gs[g8] = ((((((r_v[0].x + r_v[0].y) + r_v[0].z) + r_v[0].w) + r_v[1].x) + r_v[1].y) + r_v[1].z) + r_v[1].w;
        }
      }
      // first two fragments of an LDS-resident tile, one half-step ahead of its first MFMA
      if (hs_ + 1 < M::N0E && M::is_lds(hs_ + 1) && sl == 3) { aw[0] = lw[M::lidx(hs_ + 1) * V2_TILE_FRAGS]; aw[1] = lw[M::lidx(hs_ + 1) * V2_TILE_FRAGS + 64]; }
      if (hs_ == 5 || hs_ == 6) {
        // ---- gathered rows of unit tau - 1 -> X (transposed), first-Linear input -> Bx (layout of v2_set_in, tp_conv_bf16_dev.h)
        if (hs_ == 5 && sl >= 1 && sl <= 3) {
          const int kk = sl - 1;
          if (!(kk == 2 && W != 2)) {
            const int gran = kk < 2 ? q4 + 8 * kk : q4 + 12;
            const f32x4 r = gx[kk < 2 ? kk : (W == 2 ? 2 : 0)];
            float* o = xw + (4 * gran) * 32 + j;
            o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
          }
        }
        if (hs_ == 6 && sl >= 1 && sl <= 3) {
          const int seg = sl - 1;
          const int hb = q4 >> 2, qq = q4 & 3;
          const f32x4 x = seg == 0 ? gf[0] : seg == 1 ? gf[1] : gx[0];
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          bf16x4 pk;
          pk[0] = (__bf16)x.x; pk[1] = (__bf16)x.y; pk[2] = (__bf16)x.z; pk[3] = (__bf16)x.w;
          *reinterpret_cast<bf16x4*>(bxw + (((2 * seg + (qq >> 1)) * 64 + 32 * hb + j) * 8 + 4 * (qq & 1))) = pk;
        }
      }
    };

    // ================= 0e tiles: staggered chains (three accumulators: two tiles in flight, one in its epilogue)
    f32x16 acc3[3];
    float mid3[3];
    auto mf = [&](int k, int q) __attribute__((always_inline)) {      // MFMA q of this wave's tile k
      const f32x16 zero = {};
      f32x16& a = acc3[k % 3];
      if (!M::is_lds(k)) {
        if (q == 0) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[M::ridx(k)][q], h.v[q], zero, 0, 0, 0);
        else a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[M::ridx(k)][q], h.v[q], a, 0, 0, 0);
      } else {
        // fragment q + 2 into the register fragment q - 1 was read from: that MFMA has completed (MFMA q depends on it and an MFMA of
        // the other tile lies in between)
        if (q + 2 < V2_NFRAG) aw[(q + 2) % 3] = lw[M::lidx(k) * V2_TILE_FRAGS + (q + 2) * 64];
        if (q == 0) a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q % 3], h.v[q], zero, 0, 0, 0);
        else a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aw[q % 3], h.v[q], a, 0, 0, 0);
      }
    };
    auto slot = [&](int hs_, int sl) __attribute__((always_inline)) {
      if (hs_ >= 2) {      // epilogue slice of tile hs_ - 2
        const f32x16& ya = acc3[(hs_ - 2) % 3];
        const float ma = mid3[(hs_ - 2) % 3];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (r >= S_EPI_LO[sl] && r < S_EPI_LO[sl + 1]) o0e[r] = fmaf(ma, ya[r], o0e[r]);
      }
      side(hs_, sl);
    };
    unsigned long long d_pre = 0, d_a = 0, d_b = 0, d_c = 0;      // DIAG 6: finer clocks of the fused iteration (prologue | half-steps 0-2 | 3-7 | 8-14 | tail | barrier)
    if constexpr (DIAG == 6) d_pre = stamp();
#pragma unroll
    for (int hs_ = 0; hs_ <= M::N0E; ++hs_) {
      if constexpr (DIAG == 6) { if (hs_ == 3) d_a = stamp(); if (hs_ == 8) d_b = stamp(); }
      if (hs_ >= 1) mid3[(hs_ - 1) % 3] = mid_ld(M::I_LO + hs_ - 1);      // used one half-step from now
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        if (hs_ >= 1) {
          mf(hs_ - 1, 3 + i);
          __builtin_amdgcn_sched_barrier(0);
          slot(hs_, 2 * i);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (hs_ < M::N0E) {
          mf(hs_, i);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (hs_ < M::N0E || hs_ >= 1) {      // (the last half-step has no second chain: both slices behind the one MFMA)
          slot(hs_, 2 * i + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (DIAG == 6) d_c = stamp();
    {
      const f32x16& ya = acc3[(M::N0E - 1) % 3];
      const float ma = mid3[(M::N0E - 1) % 3];
#pragma unroll
      for (int r = 0; r < 16; ++r) o0e[r] = fmaf(ma, ya[r], o0e[r]);
    }
    if constexpr (W == 2) {   // bias product k-step 2: the six dot mids (lower lane half; the upper half's slice is padding), onto the finished sums
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) bm[jj] = (__bf16)((jj < SS.n1o && hf == 0) ? dots[jj < 6 ? jj : 0] : 0.f);
      o0e = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab0e[2], bm, o0e, 0, 0, 0);
    }
    if constexpr (W == 1) {      // the bias product (accumulated between the MFMAs of half-steps 12 / 13, `ob`) onto the finished partial sums
#pragma unroll
      for (int r = 0; r < 16; ++r) o0e[r] += ob[r];
    }
    VOut vo;
    vo.s1o[0] = vo.s1o[1] = vo.s1o[2] = 0.f;
    if constexpr (W == 2) {
      // ---- block 1o, tiles 0 .. 3: mids 0 .. 19 are (scalar feature) x (edge direction): sum_i x_i w_io, the direction is applied by wave 3
      f32x16 cb, a2[2];
      RawT raw2[2];
      s_load_bias(bias_lds, S_T1O, hf, cb);
      v_step<VK_1O, 0, 0, 0>(wt[V0 + 0], h, cb, bias_lds, S_T1O + 1, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
      v_step<VK_1O, 1, VK_1O, 0>(wt[V0 + 1], h, cb, bias_lds, S_T1O + 2, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
      v_step<VK_1O, 2, VK_1O, 1>(wt[V0 + 2], h, cb, bias_lds, S_T1O + 3, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
      v_step<VK_1O, 3, VK_1O, 2>(wt[V0 + 3], h, cb, bias_lds, -1, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
      v_drain<VK_1O, 3>(a2[1], raw2[1], v, vo);
    }
    // ---- partial sums -> LDS (lane-linear float4 rows)
    {
      f32x4* const p4 = reinterpret_cast<f32x4*>(lds + L_P + (ut & 1) * L_P_SLOT + W * L_P_WAVE) + lane;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) p4[qq * 64] = f32x4{o0e[4 * qq], o0e[4 * qq + 1], o0e[4 * qq + 2], o0e[4 * qq + 3]};
      if constexpr (W == 2) p4[4 * 64] = f32x4{vo.s1o[0], vo.s1o[1], vo.s1o[2], 0.f};
    }
    unsigned long long c4 = 0, c5 = 0;
    if constexpr (DIAG == 4 || DIAG == 6) c4 = stamp();
    // ================= waves 0 and 1: run-length sums of unit tau - 5 -> global memory, half of the columns each
    if constexpr (W <= 1) {
      if (red_on) {
        const float* const om = lds + L_O + (ur & 1) * L_O_SLOT;
        const size_t tile = (size_t)(u0 + ur);
        s_reduce_groups<NODE_STRIDE, W * RED_COLS, W == 0 ? RED_COLS : NODE_DIM - RED_COLS>(om, r_gs, r_sme, r_starts, r_last, lane,
                                                                                            G.first_sum + tile * NODE_STRIDE,
                                                                                            G.last_sum + tile * NODE_STRIDE, G.run_acc);
      }
    }
    if constexpr (DIAG == 4 || DIAG == 6) c5 = stamp();
    lds_barrier();
    if constexpr (DIAG == 4) {
      const unsigned long long c6 = stamp();
      clk[2] += c4 - c0; clk[3] += c5 - c4; clk[5] += c6 - c5;
    }
    if constexpr (DIAG == 6) {      // prologue | half-steps 0-2 | 3-7 | 8-14 | last epilogue, partial sums, reduction | barrier
      const unsigned long long c6 = stamp();
      clk[0] += d_pre - c0; clk[1] += d_a - d_pre; clk[2] += d_b - d_a; clk[3] += d_c - d_b; clk[4] += c5 - d_c; clk[5] += c6 - c5;
    }
  }
}

// ---- wave 3 over the units [u0, u0 + n) of one group entry ---------------------------------------------------------------------
// Pipeline over tau = 0 .. n + 4 (one barrier per iteration):  unit tau: edge indices; tau-1: gathers (all four waves; X rows, first-Linear
// input);  tau-2: first Linear (waves 0 .. 2, one hidden tile each) -> H;  tau-3: waves 0 .. 2 tiles -> partials P;  tau-4: wave 3 -> message
// tile O;  tau-5: wave 0 reduction -> global memory.  Wave 3: its share of the gathers, the sum of the partials, block 1o tiles 4 .. 8,
// blocks 1e and 0o as software-pipelined steps (v_step), the message tile.
template <int NT, int DIAG>
__device__ __forceinline__ void s_segment_b(const ConvGroup& G, const int cnt, const int u0, const int n, float* const lds,
                                            const bf16x8 (&wt)[NT][V2_NFRAG], const bf16x8 (&ab0e)[3], const int lane,
                                            unsigned long long (&clk)[6]) {
  constexpr int W = 3;
  const int j = lane & 31, hf = lane >> 5;
  const int q4 = 2 * W + hf;                       // this lane's 16-byte granule (columns 4 q4 ..) of the 32-column segments
  const float* const bias_lds = lds + L_BIAS;
  const GPtr<float> g_node = (GPtr<float>)G.node_in;
  const GPtr<float> g_attr = (GPtr<float>)G.attr;
  int i_dst = 0, i_src = 0, i_attr = 0;            // indices of edge j of the unit whose gathers are issued next
  float vn[3] = {0.f, 0.f, 0.f};                   // edge direction of the unit this wave's tiles process next iteration
  auto edge_of = [&](int u) __attribute__((always_inline)) -> int {      // clamped edge index of lane j in unit u (any u)
    int e = (u0 + u) * SU + j;
    e = e < cnt ? e : cnt - 1;
    return e > 0 ? e : 0;
  };

#pragma unroll 1
  for (int tau = 0; tau < n + 5; ++tau) {
    unsigned long long c0 = 0, e_pre = 0, e_a = 0, e_b = 0, e_c = 0;      // DIAG 6: prologue | block 1o steps | block 1e steps | block 0o steps | message tile | barrier
    if constexpr (DIAG == 4 || DIAG == 6) c0 = stamp();
    const int ug = tau - 1, ut = tau - 4;
    // ================= LDS operands: h, the three partial sums of the 0e block, wave 2's scalar sums of block 1o
    const float* const xc = lds + L_X + (ut & 3) * L_X_SLOT + j;
    Act6 h;
    s_load_act(lds + L_H + h_slot(ut) * L_FRAG_SLOT, lane, h);
    const float* const pw = lds + L_P + (ut & 1) * L_P_SLOT;
    const f32x4 sp = (reinterpret_cast<const f32x4*>(pw) + lane)[3 * (L_P_WAVE / 4)];
    f32x16 cb;
    s_load_bias(bias_lds, S_T1O + 4, hf, cb);
    __builtin_amdgcn_sched_barrier(0);      // the LDS reads above are in flight while the gather addresses below are formed
    // ================= gathers of unit tau - 1: issue (written to LDS between the MFMAs of steps 5 and 6)
    f32x4 gx[3], gf[2];
    {
      const GPtr<f32x4> rowd = (GPtr<f32x4>)(g_node + (size_t)i_dst * NODE_STRIDE);
      gx[0] = rowd[q4];
      gx[1] = rowd[q4 + 8];
      gx[2] = rowd[18];                              // granule 18 (columns 72 .. 75)
      gf[0] = ((GPtr<f32x4>)(g_attr + (size_t)i_attr * 32))[q4];
      gf[1] = ((GPtr<f32x4>)(g_node + (size_t)i_src * NODE_STRIDE))[q4];
    }
    {
      const int ec = edge_of(tau);
      i_dst = G.dst[ec]; i_src = G.src[ec]; i_attr = G.attr_idx[ec];
    }
    float v[3] = {vn[0], vn[1], vn[2]};
    {
      const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[edge_of(ut + 1)];
      vn[0] = vv.x; vn[1] = vv.y; vn[2] = vv.z;
    }
    f32x16 o0e;
    {
      const f32x4* const p4 = reinterpret_cast<const f32x4*>(pw) + lane;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 a = p4[qq * 64], b = p4[(L_P_WAVE / 4) + qq * 64], c = p4[2 * (L_P_WAVE / 4) + qq * 64];
        o0e[4 * qq + 0] = (a.x + b.x) + c.x; o0e[4 * qq + 1] = (a.y + b.y) + c.y;
        o0e[4 * qq + 2] = (a.z + b.z) + c.z; o0e[4 * qq + 3] = (a.w + b.w) + c.w;
      }
    }
    float* const xw = lds + L_X + (ug & 3) * L_X_SLOT;
    __bf16* const bxw = reinterpret_cast<__bf16*>(lds + L_BX + (ug & 1) * L_FRAG_SLOT);
    auto gather_write = [&](int part, int q) __attribute__((always_inline)) {
      if (part == 0 && q >= 1 && q <= 3) {
        const int kk = q - 1;
        const int gran = kk < 2 ? q4 + 8 * kk : 18;
        const f32x4 r = gx[kk];
        float* o = xw + (4 * gran) * 32 + j;
        o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;      // (granule 18: both lane halves hold and store the same values -- no predicate, no control flow inside the chain)
      }
      if (part == 1 && q >= 1 && q <= 3) {
        const int seg = q - 1;
        const int hb = q4 >> 2, qq = q4 & 3;
        const f32x4 x = seg == 0 ? gf[0] : seg == 1 ? gf[1] : gx[0];
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 pk;
        pk[0] = (__bf16)x.x; pk[1] = (__bf16)x.y; pk[2] = (__bf16)x.z; pk[3] = (__bf16)x.w;
        *reinterpret_cast<bf16x4*>(bxw + (((2 * seg + (qq >> 1)) * 64 + 32 * hb + j) * 8 + 4 * (qq & 1))) = pk;
      }
    };

    // ================= block 1o tiles 4 .. 8 (mids 20 .. 43 + one padded slot), block 1e's own three tiles, block 0o's three (with 1e's
    //                   tail mids as guests) as software-pipelined steps
    VOut vo;
    vo.s1o[0] = sp.x; vo.s1o[1] = sp.y; vo.s1o[2] = sp.z;
    vo.s1e[0] = vo.s1e[1] = vo.s1e[2] = 0.f;
    vo.k0o[0] = vo.k0o[1] = vo.k0o[2] = 0.f;
#pragma unroll
    for (int r = 0; r < 9; ++r) { vo.k1o[r] = 0.f; vo.k1e[r] = 0.f; }
    f32x16 a2[2];
    RawT raw2[2];
    if constexpr (DIAG == 6) e_pre = stamp();
    v_step<VK_1O, 4, 0, 0>(wt[0], h, cb, bias_lds, S_T1O + 5, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
    v_step<VK_1O, 5, VK_1O, 4>(wt[1], h, cb, bias_lds, S_T1O + 6, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
    v_step<VK_1O, 6, VK_1O, 5>(wt[2], h, cb, bias_lds, S_T1O + 7, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
    v_step<VK_1O, 7, VK_1O, 6>(wt[3], h, cb, bias_lds, S_T1O + 8, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
    v_step<VK_1O, 8, VK_1O, 7>(wt[4], h, cb, bias_lds, S_T1E + 0, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
    if constexpr (DIAG == 6) e_a = stamp();
    v_step<VK_1E, 0, VK_1O, 8>(wt[5], h, cb, bias_lds, S_T1E + 1, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo,
                               [&](int q) __attribute__((always_inline)) { gather_write(0, q); });
    v_step<VK_1E, 1, VK_1E, 0>(wt[6], h, cb, bias_lds, S_T1E + 2, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo,
                               [&](int q) __attribute__((always_inline)) { gather_write(1, q); });
    v_step<VK_1E, 2, VK_1E, 1>(wt[7], h, cb, bias_lds, S_T0O + 0, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
    if constexpr (DIAG == 6) e_b = stamp();
    v_step<VK_0O, 0, VK_1E, 2>(wt[8], h, cb, bias_lds, S_T0O + 1, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
    v_step<VK_0O, 1, VK_0O, 0>(wt[9], h, cb, bias_lds, S_T0O + 2, hf, xc, a2[1], a2[0], raw2[1], raw2[0], v, vo);
    v_step<VK_0O, 2, VK_0O, 1>(wt[10], h, cb, bias_lds, -1, hf, xc, a2[0], a2[1], raw2[0], raw2[1], v, vo);
    v_drain<VK_0O, 2>(a2[0], raw2[0], v, vo);
    if constexpr (DIAG == 6) e_c = stamp();
    // ---- message tile [col][36]
    {
      float* const om = lds + L_O + (ut & 1) * L_O_SLOT;
#pragma unroll
      for (int r = 0; r < 16; ++r) om[((r & 3) + 8 * (r >> 2) + 4 * hf) * S_OSTR + j] = o0e[r];
#pragma unroll
      for (int o = 0; o < 3; ++o) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          om[(COL_1O + 3 * (3 * hf + o) + c) * S_OSTR + j] = fmaf(v[c], vo.s1o[o], vo.k1o[3 * o + c]);
          om[(COL_1E + 3 * (3 * hf + o) + c) * S_OSTR + j] = fmaf(v[c], vo.s1e[o], vo.k1e[3 * o + c]);
        }
        om[(COL_0O + 3 * hf + o) * S_OSTR + j] = vo.k0o[o];
      }
    }
    unsigned long long c5 = 0;
    if constexpr (DIAG == 4 || DIAG == 6) c5 = stamp();
    lds_barrier();
    if constexpr (DIAG == 4) {
      const unsigned long long c6 = stamp();
      clk[2] += c5 - c0; clk[5] += c6 - c5;
    }
    if constexpr (DIAG == 6) {
      const unsigned long long c6 = stamp();
      clk[0] += e_pre - c0; clk[1] += e_a - e_pre; clk[2] += e_b - e_a; clk[3] += e_c - e_b; clk[4] += c5 - e_c; clk[5] += c6 - c5;
    }
  }
}

template <int W, int DIAG>
__device__ __forceinline__ void s_wave_prog(const ConvArgs& args, float* const lds, const float* const wstream, const int lane,
                                            const int u_lo, const int u_hi, const int mine, const int incl, const int role) {
  // stationary tiles (stream indices): wave 0: 3 .. 15 (0e mids 0 .. 12; mid 13 = tile 16 is read from LDS), wave 1: 17 .. 22 and 24 .. 29
  // (23, 30 from LDS), wave 2: 31 .. 39 (0e mids 28 .. 36; 37 = tile 40 from LDS) and 41 .. 44 (block 1o tiles 0 .. 3), wave 3: 45 .. 55
  constexpr int NT = W == 3 ? 11 : W == 1 ? 12 : 13;
  const GFrag gp = (GFrag)reinterpret_cast<const bf16x8*>(wstream);
  bf16x8 wt[NT][V2_NFRAG];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int T = W == 0 ? 3 + t : W == 1 ? (t < 6 ? 17 + t : 18 + t) : W == 2 ? (t < 9 ? 31 + t : 41 + (t - 9)) : 45 + t;
#pragma unroll
    for (int q = 0; q < V2_NFRAG; ++q) wt[t][q] = gp[(size_t)T * V2_TILE_FRAGS + q * 64 + lane];
  }
  // Ten tiles live in the accumulation half of the register file (256 registers; the MFMA reads srcA from it directly), the rest in
  // ordinary VGPRs -- pinning more than fit makes hipcc copy the overflow in front of every MFMA that uses it.  The pins come AFTER all
  // the loads have been issued: a pin right behind its load made every one of the 78 loads wait for its own data (vmcnt(0) each,
  // tens of microseconds per workgroup and launch).
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < V2_NFRAG; ++q) {
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
  if constexpr (DIAG >= 4 && DIAG <= 6) { st0 = stamp(); sr0 = __builtin_amdgcn_s_memrealtime(); }      // 4: phase clocks, 5: lifetimes only, 6: finer clocks of waves 0 .. 2
#pragma unroll 1
  for (int g = 0; g < args.n_groups; ++g) {
    const int mg = __builtin_amdgcn_readlane(mine, g);
    if (mg == 0) continue;
    const int end = __builtin_amdgcn_readlane(incl, g), start = end - mg;
    const int a = u_lo > start ? u_lo : start, b = u_hi < end ? u_hi : end;
    if (a >= b) continue;
    const ConvGroup G = args.g[g];
    const int cnt = *G.count;
    if constexpr (W <= 2) s_segment_a<W, NT, DIAG>(G, cnt, a - start, b - a, lds, wt, ab0e, lane, clk);
    else s_segment_b<NT, DIAG>(G, cnt, a - start, b - a, lds, wt, ab0e, lane, clk);
  }
  if constexpr (DIAG >= 4 && DIAG <= 6) {
    const int rec = blockIdx.x * SW_WAVES + W;
    if (lane == 0 && args.stamps && rec < 8192 / 2) {
      unsigned long long* o = args.stamps + (size_t)rec * 16;
      o[0] = st0; o[1] = sr0; o[2] = stamp(); o[3] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
      for (int k = 0; k < 6; ++k) o[4 + k] = clk[k];
      o[10] = (unsigned long long)(u_hi - u_lo);
      o[11] = (unsigned long long)role;
    }
  }
}

template <int DIAG = 0>
__global__ __launch_bounds__(SW_WAVES * 64, 1) void tp_conv64s_kernel(ConvArgs args, RoleTableS rt) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;

  // ---- work split: 32-edge units per entry (lane g <-> entry g); the units of the launch in ROLE-MAJOR order form one range that is cut
  //      into gridDim.x equal pieces.  A workgroup whose piece straddles a role boundary finishes the first role, reloads its registers
  //      and LDS with the next role's FCBlock and goes on (a few microseconds, at most n_roles - 1 workgroups per launch).  Dealing whole
  //      workgroups to roles in proportion to their units looked simpler and cost up to 2x: the small ligand-ligand role deserved 1.8
  //      workgroups, got 1, and that workgroup finished last (profiles/r05_*).
  int units = 0, my_role = -1;
  if (lane < args.n_groups) {
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const int* cp = *reinterpret_cast<const int* const*>(ka + offsetof(ConvArgs, g) + (size_t)lane * sizeof(ConvGroup) + offsetof(ConvGroup, count));
    units = (*cp + SU - 1) / SU;
    my_role = *reinterpret_cast<const unsigned char*>(ka + ((sizeof(ConvArgs) + alignof(RoleTableS) - 1) / alignof(RoleTableS)) * alignof(RoleTableS) + offsetof(RoleTableS, role_of) + lane);
  }
  //      Round 6: equal pieces of WORK, not of units.  A unit's cost depends on its role -- the run-length reduction walks every eight-edge
  //      group that holds a run boundary, and the groups differ in run length: ligand-ligand units (runs of ~17 edges) cost 8 % more than
  //      ligand->receptor units (runs of hundreds), receptor units 3-4 % more (per-workgroup lifetimes, tools/conv_span_wg.py:
  //      2484 / 2292 / 2384 / 2364 ns per unit for ll / lr / rr / rl) -- and with equal unit counts the workgroups inside the expensive
  //      roles finished 6-7 % after the mean and set the launch time.  Every role's units are weighed with RoleTableS::weight (1/64;
  //      70 / 64 / 68 / 66 from a sweep on the C4 leg: 220.9 -> 228.7 .. 233 poses/s on one box, profiles/r06_n_split_weights.txt).
  const int total = s_wave_sum(units);
  if (total == 0) return;
  const int n_wg = gridDim.x;
  long long total_cost = 0;
#pragma unroll 1
  for (int role = 0; role < rt.n_roles; ++role) total_cost += (long long)s_wave_sum(my_role == role ? units : 0) * rt.weight[role];
  const long long C_lo = total_cost * blockIdx.x / n_wg, C_hi = total_cost * (blockIdx.x + 1) / n_wg;
  if (C_lo >= C_hi) return;
  long long acc_cost = 0;
#pragma unroll 1
  for (int role = 0; role < rt.n_roles; ++role) {
    const int w = s_wave_sum(my_role == role ? units : 0);
    const int W = rt.weight[role];
    const long long A_r = acc_cost;
    acc_cost += (long long)w * W;
    const long long a = C_lo > A_r ? C_lo : A_r, b = C_hi < acc_cost ? C_hi : acc_cost;
    if (a >= b) continue;
    // unit boundaries = floor(cost offset / W): the piece boundaries of neighbouring workgroups coincide, so the units tile exactly
    const int u_lo = (int)((a - A_r) / W), u_hi = (int)((b - A_r) / W);
    if (u_lo >= u_hi) continue;
    const float* const wstream = rt.wstream[role];
    // ---- bias rows, the three first-Linear tiles and one second-Linear tile per wave 0 .. 2 -> LDS
    __syncthreads();
    {
      const float* gb = reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(wstream) + (size_t)(SS.ntiles + 1) * V2_TILE_FRAGS);
      for (int k = threadIdx.x; k < S_BIAS_ROWS * 32; k += SW_WAVES * 64) {
        const int row = k >> 5;
        lds[L_BIAS + k] = gb[(row < 3 ? row : 3 + SS.t0e + (row - 3)) * 32 + (k & 31)];
      }
      const f32x4* src = reinterpret_cast<const f32x4*>(wstream);
      f32x4* dst = reinterpret_cast<f32x4*>(lds + L_FLW);
      for (int k = threadIdx.x; k < 3 * V2_TILE_FRAGS; k += SW_WAVES * 64) dst[k] = src[k];
      // stream tiles 16 | 23, 30 | 40 = 0e tiles of waves 0 | 1 | 2 (mids 13 | 20, 27 | 37), WMap
      f32x4* dw = reinterpret_cast<f32x4*>(lds + L_WT);
      for (int k = threadIdx.x; k < S_LDS_TILES * V2_TILE_FRAGS; k += SW_WAVES * 64) {
        const int ww = k / V2_TILE_FRAGS, r = k - ww * V2_TILE_FRAGS;
        dw[k] = src[(size_t)lds_tile_stream(ww) * V2_TILE_FRAGS + r];
      }
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
    if (wave == 0) s_wave_prog<0, DIAG>(args, lds, wstream, lane, u_lo, u_hi, mine, incl, role);
    else if (wave == 1) s_wave_prog<1, DIAG>(args, lds, wstream, lane, u_lo, u_hi, mine, incl, role);
    else if (wave == 2) s_wave_prog<2, DIAG>(args, lds, wstream, lane, u_lo, u_hi, mine, incl, role);
    else s_wave_prog<3, DIAG>(args, lds, wstream, lane, u_lo, u_hi, mine, incl, role);
  }
}

// a: the edge groups of a 74 -> 74 layer (whole tile chains); n_wg: workgroups (<= CUs).  Roles = distinct weight streams.
// Role table of a launch: one role per distinct FCBlock (weight stream) among its groups.  false: the launch does not fit this kernel --
// virtual slices of the role split, or more than S_MAX_ROLES FCBlocks (cannot happen through cbd_sample_multi, which wants shared weights:
// four roles per launch; the check keeps the kernel's table safe whatever builds the arguments).
static bool s_role_table(const ConvArgs& a, RoleTableS& rt) {
  for (int g = 0; g < a.n_groups; ++g) {
    const ConvGroup& G = a.g[g];
    if (!G.vec_on || G.i0e_lo != 0 || G.i0e_hi != SS.t0e) return false;
    int r = -1;
    for (int k = 0; k < rt.n_roles; ++k)
      if (rt.wstream[k] == G.wstream) r = k;
    if (r < 0) {
      if (rt.n_roles == S_MAX_ROLES) return false;
      r = rt.n_roles++;
      rt.wstream[r] = G.wstream;
      rt.weight[r] = (unsigned char)(G.cost_w > 0 && G.cost_w < 256 ? G.cost_w : 64);
#ifdef CBD_DIAG      // diagnostic library only: CBD_S_EQUAL_UNITS=1 restores the equal-unit split of round 5 for A/B runs in one process / on one box
      static const bool equal_units = getenv("CBD_S_EQUAL_UNITS") && atoi(getenv("CBD_S_EQUAL_UNITS")) != 0;
      if (equal_units) rt.weight[r] = 64;
#endif
    }
    rt.role_of[g] = (unsigned char)r;
  }
  return true;
}

bool tp_conv_bf16s_fits(const ConvArgs& a) {
  RoleTableS rt{};
  return s_role_table(a, rt);
}

hipError_t launch_tp_conv_bf16s(const ConvArgs& a, int n_wg, hipStream_t s) {
  if (a.n_groups <= 0) return hipSuccess;
  RoleTableS rt{};
  if (!s_role_table(a, rt)) return hipErrorInvalidValue;
  // the opt-in LDS size is a per-DEVICE function attribute: one flag per device, not one per process
  static std::atomic<unsigned long long> attr_set{0};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const unsigned long long bit = 1ull << (dev & 63);
  if (!(attr_set.load(std::memory_order_acquire) & bit)) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64s_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    if (e != hipSuccess) return e;
#ifdef CBD_DIAG
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64s_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64s_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64s_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
    if (e != hipSuccess) return e;
#endif
    attr_set.fetch_or(bit, std::memory_order_release);
  }
#ifdef CBD_DIAG      // diagnostic library only (tools/diag_lib.py): 4 = per-wave phase clocks, 5 = workgroup lifetimes, 6 = finer clocks (correct results)
  static const int diag_env = getenv("CBD_BF16_DIAG") ? atoi(getenv("CBD_BF16_DIAG")) : 0;
  static const int diag_min_roles = getenv("CBD_DIAG_MIN_ROLES") ? atoi(getenv("CBD_DIAG_MIN_ROLES")) : 0;      // stamps only from launches with at least so many roles
  const int diag = rt.n_roles >= diag_min_roles ? diag_env : 0;
  if (diag == 4) hipLaunchKernelGGL((tp_conv64s_kernel<4>), dim3(n_wg), dim3(SW_WAVES * 64), S_LDS_BYTES, s, a, rt);
  else if (diag == 5) hipLaunchKernelGGL((tp_conv64s_kernel<5>), dim3(n_wg), dim3(SW_WAVES * 64), S_LDS_BYTES, s, a, rt);
  else if (diag == 6) hipLaunchKernelGGL((tp_conv64s_kernel<6>), dim3(n_wg), dim3(SW_WAVES * 64), S_LDS_BYTES, s, a, rt);
  else
#endif
  hipLaunchKernelGGL((tp_conv64s_kernel<0>), dim3(n_wg), dim3(SW_WAVES * 64), S_LDS_BYTES, s, a, rt);
  return hipGetLastError();
}

}  // namespace cbd
