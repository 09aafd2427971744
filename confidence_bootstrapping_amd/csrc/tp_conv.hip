// Fused tensor-product message passing for gfx950 (MI355X):
//   per edge:  h = ReLU(W1 [edge_attr | x_src[:32] | x_dst[:32]] + b1)            (FCBlock layer 1)
//              w = W2 h + b2                                                       (FCBlock layer 2, 1216..1660 wide)
//              msg = FasterTensorProduct(x_dst, sh(edge_vec), w)                   (lmax = 1 CG paths)
//   per node:  acc[src] += msg                                                     (mean/BN/residual: finalize kernel)
// replacing reference models/tensor_layers.py:195-206 (FCBlock -> FasterTensorProduct.forward:66-117 ->
// torch_scatter.scatter) without ever materialising the [E, W] per-edge weight tensor in HBM.
//
// Mapping onto the matrix cores (described for the exact-fp32 policy OpsF32, v_mfma_f32_32x32x2_f32; OpsBf16 below is the
// bf16-operand policy of the same kernel):
//   D[row = weight column, col = edge] = sum_k  A[row][k] * B[k][edge]
//   * B (activations) lives in registers for the whole tile: lane (j = lane&31, hf = lane>>5) holds
//     act[edge j][k(s, hf)] for the 48 k-steps s.  The accumulator of the first Linear (after bias+ReLU) IS the
//     B operand of the second Linear -- the k order of W2 is permuted at weight-packing time to the C/D layout
//     row(reg, hf) = (reg&3) + 8*(reg>>2) + 4*hf, so nothing moves between the two GEMMs.
//   * A (weights) is streamed as 32-row tiles [12 x 64 lanes x float4] straight from L2 into registers in MFMA
//     operand order, one tile ahead of its use (every wave streams the whole 0.7 MB of its group's FCBlock, which
//     is L2 resident: no LDS staging, no barrier in the tile loop, waves fully independent -> one wave per workgroup).
//   * Each finished 32x32 tile of w is consumed immediately by the CG contraction on the VALU: lane (j, hf)
//     multiplies its 16 accumulator rows with the "mid" value(s) of edge j (scalar, dot, cross or outer
//     products of x_dst and the unit edge vector, read from a per-wave LDS copy of the gathered row) and adds
//     into per-lane output accumulators.  1/sqrt(fan_in), sqrt(3) (sh scale) and 1/sqrt(2) factors are folded
//     into the packed weights.
//   * Rows of a vector-block tile hold 5 mid indices x 6 outputs: accumulator register reg < 15 of lane half hf is
//     (i = 5*tile + reg/3, o = 3*hf + reg%3), so both the mid index and the output slot are compile-time functions of
//     the register index, each half owns three of the six outputs, and 30 of the 32 MFMA rows carry weights.
//   * Segmented reduction without atomics: edges are sorted by aggregating node; each wave run-length sums its 32
//     messages in LDS and stores per-tile pieces (first run / last run / interior runs) that conv_finalize_kernel adds
//     in a fixed order -> bitwise reproducible results.
#include <cstdlib>

#include <type_traits>

#include "kernels.h"
#include "tp_conv_dev.h"

namespace cbd {


template <int IN, int OUT, int VAR, class Ops>
__global__ __launch_bounds__(64, 2) void tp_conv_kernel(ConvArgs args) {
  constexpr ConvShape S = conv_shape(IN, OUT, true);   // merged vector tails (common.h)
  constexpr bool STAMPS = VAR == 8 || VAR == 13;     // diagnostics: 13 = the stamps of 8 on the gather pattern of 12
  constexpr int GV = VAR == 13 ? 12 : VAR;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bias_l = lds;                                 // [ntiles][32]
  float* xT = lds + S.ntiles * 32;                     // [80][32] gathered destination rows, transposed
  int* srcl = reinterpret_cast<int*>(xT + XT_FLOATS);  // [32]
  const int lane = threadIdx.x;
  const int j = lane & 31, hf = lane >> 5;

  // ---- which group / edge range does this wave own?  (edge counts live on the device)
  int grp = -1, e0 = 0, cnt = 0, tile_local = 0;
  if (!find_group(args, blockIdx.x, lane, CONV_WG_EDGES, grp, tile_local, cnt)) return;
  e0 = tile_local * CONV_WG_EDGES;
  const ConvGroup G = args.g[grp];
  unsigned long long st_t0 = 0, st_r0 = 0, st_t1 = 0, st_t2 = 0, st_t3 = 0, st_g0 = 0;
  if constexpr (STAMPS) { st_t0 = stamp(); st_r0 = __builtin_amdgcn_s_memrealtime(); }

  // ---- start the weight stream: tile 0 fragments + the bias table of the group
  using Frag = typename Ops::Frag;
  // NODE_PROJ policies: wave-uniform stream base, tile T fragment sg of this lane = gu[T * TILE_FRAGS + sg * 64 + lane]
  const GPtr<Frag> gu = (GPtr<Frag>)reinterpret_cast<const Frag*>(G.wstream);
  const Frag* gp = reinterpret_cast<const Frag*>(G.wstream) + lane;   // per-lane form (policies without NODE_PROJ)
  Frag a[Ops::NFRAG];
  if constexpr (Ops::NODE_PROJ) {
    // the live fragments of all three first-Linear tiles (f32_split: tiles 0 and 1 here, tile 2 behind the row gather below -- all 18
    // of its fragments in flight next to the gathers cost a spill and a wait on it at the very start of the wave)
    if constexpr (std::is_same<Ops, OpsBf16x3>::value) Ops::template load_first_part<0, 2>(a, gu, lane);
    else Ops::load_first_u(a, gu, lane);
  } else {
#pragma unroll
    for (int sg = 0; sg < Ops::NFRAG; ++sg) a[sg] = gp[sg * 64];
  }
  {  // bias table -> LDS: fixed number of unconditional, clamped loads (a counted loop compiles to a load/wait waterfall)
    const f32x4* gb = reinterpret_cast<const f32x4*>(reinterpret_cast<const Frag*>(G.wstream) + (size_t)(S.ntiles + 1) * Ops::TILE_FRAGS);
    constexpr int NB4 = S.ntiles * 8, NBI = (NB4 + 63) / 64;
    f32x4 bt[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) { const int k = lane + 64 * i; bt[i] = gb[k < NB4 ? k : NB4 - 1]; }
#pragma unroll
    for (int i = 0; i < NBI; ++i) { const int k = lane + 64 * i; reinterpret_cast<f32x4*>(bias_l)[k < NB4 ? k : NB4 - 1] = bt[i]; }
  }

  // ---- gather the edge's inputs.  Lanes past the end of the group read the group's last edge (unconditional loads:
  //      a per-lane `valid ? load : 0` makes hipcc branch around every load and wait vmcnt(0) each time) and are
  //      dropped at the end through src = -1.
  const int e = e0 + j;
  const bool valid = e < cnt;
  const int ec = valid ? e : cnt - 1;
  const int src_r = G.src[ec], dst = G.dst[ec], aidx = G.attr_idx[ec];
  const int src = valid ? src_r : -1;
  const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[ec];
  const float v[3] = {vv.x, vv.y, vv.z};
  if (hf == 0) srcl[j] = src;

  f32x16 acc1[Ops::NODE_PROJ ? 3 : 1];
  typename Ops::Act Bx;  // first-Linear input on the matrix cores: edge_attr(32), lane half hf holds cols 16hf..16hf+15.  The
                         // x_src[:32] / x_dst[:32] parts arrive as per-node projections through the accumulator (G.psrc / G.pdst)
  {
    // VAR 14 (diagnostic, correct results): the per-edge gathers as non-temporal loads, so that they do not push the weight tiles out of L2
    auto gl = [](const f32x4* p) __attribute__((always_inline)) { return VAR == 14 ? __builtin_nontemporal_load(p) : *p; };
    const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx * 32 + 16 * hf);
#pragma unroll
    for (int q = 0; q < 4; ++q) Ops::set_in(Bx, 0, q, gl(pa + q));
    if constexpr (!Ops::NODE_PROJ) {
      const f32x4* ps = reinterpret_cast<const f32x4*>(G.node_in + (size_t)src_r * NODE_STRIDE + 16 * hf);
      const f32x4* pd = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dst * NODE_STRIDE + 16 * hf);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        Ops::set_in(Bx, 1, q, ps[q]);
        Ops::set_in(Bx, 2, q, pd[q]);
      }
    }
    if constexpr (Ops::NODE_PROJ) {
      // accumulator start values of the three first-Linear tiles, gathered HERE with everything else the edge needs (the first Linear
      // then waits on nothing): b1 + W1s x_src + W1d x_dst for rows 32m + (r&3) + 8(r>>2) + 4hf; the bias rows straight from the
      // stream's table in global memory (its LDS copy is still being written)
      const f32x4* const gb = reinterpret_cast<const f32x4*>(reinterpret_cast<const Frag*>(G.wstream) + (size_t)(S.ntiles + 1) * Ops::TILE_FRAGS);
      const f32x4* const p_s = reinterpret_cast<const f32x4*>(G.psrc + (size_t)src_r * KDIM + 4 * hf);
      // diagnostics with WRONG results (timing only): VAR 10 reads the destination projection at the aggregating node's row (run-
      // coherent instead of random), VAR 11 skips both projection gathers, VAR 12 also gathers the node row at the aggregating node
      const f32x4* const p_d = reinterpret_cast<const f32x4*>(G.pdst + (size_t)(GV >= 10 && GV < 14 ? src_r : dst) * KDIM + 4 * hf);
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 b = gb[8 * m + 2 * q + hf];
          f32x4 u = b, w = b;
          if constexpr (GV < 11 || GV == 14) { u = gl(p_s + 8 * m + 2 * q); w = gl(p_d + 8 * m + 2 * q); }
          acc1[m][4 * q + 0] = b.x + u.x + w.x; acc1[m][4 * q + 1] = b.y + u.y + w.y;
          acc1[m][4 * q + 2] = b.z + u.z + w.z; acc1[m][4 * q + 3] = b.w + u.w + w.w;
        }
    }
    // full destination row -> transposed LDS copy xT[col][j]; lane half hf copies cols 40hf .. 40hf+39
    // (f32_split: its 18-fragment weight tile + three-plane operands leave no room for the row's 40 registers NEXT TO the 48 start
    //  values above -- hipcc spilled 12 registers here; a scheduling fence makes the row gather a round of its own.  One extra memory
    //  round trip per wave, this policy only: the exact-fp32 kernel's code is unchanged.)
    if constexpr (std::is_same<Ops, OpsBf16x3>::value) {
      __builtin_amdgcn_sched_barrier(0);
      Ops::template load_first_part<2, 3>(a, gu, lane);
    }
    const f32x4* pr = reinterpret_cast<const f32x4*>(G.node_in + (size_t)(GV == 12 ? src_r : dst) * NODE_STRIDE + 40 * hf);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const f32x4 r = gl(pr + q);
      float* o = xT + (40 * hf + 4 * q) * 32 + j;
      o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
    }
  }
  __syncthreads();   // single-wave workgroup: orders the LDS writes above before the reads below
  if constexpr (STAMPS) st_t1 = stamp();

  int T = 0;
  f32x16 acc;
  typename Ops::Act h1;
  // the stream carries one zero tile after the last real one, so the prefetch of tile T+1 is always in bounds
#define CBD_TILE(BOP, NEXT)                                                                         \
  {                                                                                                 \
    const int tn_ = (NEXT);                                                                         \
    if constexpr (Ops::NODE_PROJ) gemm_tile_u<Ops>(a, gu + (VAR == 9 ? (size_t)0 : (size_t)tn_ * Ops::TILE_FRAGS), lane, bias_l + T * 32, BOP, acc, hf); \
    else gemm_tile<Ops>(a, gp + (VAR == 9 ? (size_t)0 : (size_t)tn_ * Ops::TILE_FRAGS), bias_l + T * 32, BOP, acc, hf); \
    T = tn_;                                                                                        \
  }

  // ---- first Linear (3 tiles): h1 = ReLU(W1 x + b1), kept in the C/D register layout
  // A group may be a VIRTUAL slice of an edge group (ConvGroup::i0e_lo/hi, vec_on): the same edges, but only the 0e tiles
  // [lo, hi) and/or the vector blocks -- several waves then share one 32-edge tile's weight-tile chain (short launches)
  const int i_lo = G.i0e_lo, i_hi = G.i0e_hi;
  const bool vec_on = G.vec_on != 0;
  const int T_vec = 3 + S.t0e;
  if constexpr (!Ops::NODE_PROJ) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      CBD_TILE(Bx, m < 2 ? T + 1 : (i_lo < i_hi ? 3 + i_lo : T_vec));
      if constexpr (!Ops::EXACT_F32) { if (m == 2) mfma_operand_guard(); }   // the first-Linear operands die here without a refill
      Ops::set_hidden(h1, m, acc);
      if constexpr (STAMPS) { if (m == 0) st_g0 = stamp(); }
    }
  } else {
    // K = 32 edge-attribute product on top of acc1[m]; tile m's registers are refilled with the first second-Linear tile's fragments
    const int tn_ = i_lo < i_hi ? 3 + i_lo : T_vec;
    const GPtr<Frag> next = gu + (VAR == 9 ? (size_t)0 : (size_t)tn_ * Ops::TILE_FRAGS);
    Ops::template gemm_first_u<0>(a, next, lane, Bx, acc1[0]);
    Ops::set_hidden(h1, 0, acc1[0]);
    if constexpr (STAMPS) st_g0 = stamp();
    Ops::template gemm_first_u<1>(a, next, lane, Bx, acc1[1]);
    Ops::set_hidden(h1, 1, acc1[1]);
    Ops::template gemm_first_u<2>(a, next, lane, Bx, acc1[2]);
    if constexpr (!Ops::EXACT_F32) mfma_operand_guard();   // Bx dies here without a refill
    Ops::set_hidden(h1, 2, acc1[2]);
    T = tn_;
  }

  if constexpr (STAMPS) st_t2 = stamp();
  const float* xc = xT + j;
  // ---- block 0e: one tile per mid index, 32 output scalars
  float o0e[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) o0e[r] = 0.f;
  // the mid is read from LDS BEFORE the MFMA chain (the scheduling fences of the chain keep the read above it; its wait lands at the
  // first use below): the LDS latency is covered by the tile's MFMAs instead of being exposed after them.  Two loops, one per kind of
  // mid (the scalar features themselves, then the 1o . direction dot products), so that neither body branches on the mid index.
  auto tile0e = [&](int i, float m) __attribute__((always_inline)) {
    CBD_TILE(h1, i + 1 < i_hi ? T + 1 : (vec_on ? T_vec : S.ntiles));
#pragma unroll
    for (int r = 0; r < 16; ++r) o0e[r] = fmaf(m, acc[r], o0e[r]);
  };
  const int i_mid = i_hi < NS ? i_hi : NS;
#pragma unroll 1
  for (int i = i_lo; i < i_mid; ++i) tile0e(i, xc[i * 32]);
  if constexpr (IN >= 1) {
#pragma unroll 1
    for (int i = i_lo > NS ? i_lo : NS; i < i_hi; ++i) {
      const float* p = xc + (COL_1O + 3 * (i - NS)) * 32;
      tile0e(i, p[0] * v[0] + p[32] * v[1] + p[64] * v[2]);
    }
  }

  // ---- vector / pseudoscalar blocks: tile = 5 mid indices x 6 outputs; lane half hf owns outputs 3hf..3hf+2, register
  //      reg < 15 holds (i = 5t + reg/3, o = 3hf + reg%3) -- no cross-lane traffic, mids identical in both halves
  float k1o[9], k1e[9], k0o[3];
#pragma unroll
  for (int r = 0; r < 9; ++r) { k1o[r] = 0.f; k1e[r] = 0.f; }
  k0o[0] = k0o[1] = k0o[2] = 0.f;

  // The tile loops of the vector blocks are FULLY unrolled: the mid index is then a compile-time constant, the kind of every mid
  // (scalar x direction, copy, cross product, padding) is resolved by the compiler and the LDS reads of a tile's mids are issued
  // together.  Rolled, every mid was a chain of scalar branches around LDS reads that were each waited for in turn (44 s_waitcnt and
  // ~3.6 k cycles per vector tile, in-kernel stamps of the bf16 kernel, round 2).
  // Mids of the form (scalar feature) x (edge direction) -- the first NS mids of block 1o, the last n0o of block 1e -- are not
  // multiplied out: sum_i (x_i v_c) w_io = v_c sum_i x_i w_io, one FMA per output instead of three, the direction applied once per
  // block (`is_scalar(i)` / `scalar_of(x, i)`); padded slots i >= fan are skipped.
  auto vec_block = [&](auto mid_fn, auto is_scalar, auto scalar_of, auto ntile_c, auto fan_c, float (&keep)[9]) __attribute__((always_inline)) {
    constexpr int ntile = decltype(ntile_c)::value, fan = decltype(fan_c)::value;
    float sc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < ntile; ++t) {
      float m[VEC_TILE_I][3], xs[VEC_TILE_I];   // the tile's mids, evaluated (LDS reads + cross products) before the MFMA chain
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const int i = VEC_TILE_I * t + q;
        if (i >= fan) continue;
        if (is_scalar(i)) xs[q] = scalar_of(xc, i);
        else mid_fn(xc, i, v, m[q]);
      }
      CBD_TILE(h1, T + 1);
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const int i = VEC_TILE_I * t + q;
        if (i >= fan) continue;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const float w = acc[3 * q + o];
          if (is_scalar(i)) {
            sc[o] = fmaf(xs[q], w, sc[o]);
          } else {
            keep[3 * o + 0] = fmaf(m[q][0], w, keep[3 * o + 0]);
            keep[3 * o + 1] = fmaf(m[q][1], w, keep[3 * o + 1]);
            keep[3 * o + 2] = fmaf(m[q][2], w, keep[3 * o + 2]);
          }
        }
      }
      // pure FMAs are not tied to their place in the unrolled code: pin the running sums to the tile (otherwise hipcc parks the
      // accumulator in scratch and does the FMAs tiles later)
#pragma unroll
      for (int o = 0; o < 3; ++o) pin(sc[o]);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int c = 0; c < 3; ++c) keep[3 * o + c] = fmaf(v[c], sc[o], keep[3 * o + c]);
  };

  if (vec_on) {
    vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, vv, m); },
              [](int i) { return i < NS; }, [](const float* x, int i) { return x[i * 32]; },
              std::integral_constant<int, S.t1o>{}, std::integral_constant<int, S.fan1o>{}, k1o);
    // merged tails (ConvShape::vmerged): block 1e runs its first 5 (t1e - 1) mids; the others are guests of block 0o's last tile
    constexpr int OWN1E = S.vmerged ? VEC_TILE_I * (S.t1e - 1) : S.fan1e, GUESTS = S.fan1e - OWN1E;
    if constexpr (OUT >= 2)
      vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, vv, m); },
                [](int i) { return i >= S.n1o + S.n1e; }, [](const float* x, int i) { return x[(COL_0O + (i - S.n1o - S.n1e)) * 32]; },
                std::integral_constant<int, S.t1e - S.vmerged>{}, std::integral_constant<int, OWN1E>{}, k1e);
    if constexpr (OUT >= 3) {
#pragma unroll
      for (int t = 0; t < S.t0o; ++t) {
        float m[VEC_TILE_I];
        float gm[GUESTS > 0 ? GUESTS : 1][3];     // block 1e's tail mids (vectors; a scalar x direction mid is multiplied out here)
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          if (VEC_TILE_I * t + q >= S.fan0o) continue;
          m[q] = mid0o<IN>(xc, VEC_TILE_I * t + q, v);
        }
        if (t == S.t0o - 1) {
#pragma unroll
          for (int g = 0; g < GUESTS; ++g) {
            const int i = OWN1E + g;
            if (i >= S.n1o + S.n1e) {
              const float x = xc[(COL_0O + (i - S.n1o - S.n1e)) * 32];
              gm[g][0] = x * v[0]; gm[g][1] = x * v[1]; gm[g][2] = x * v[2];
            } else mid1e<IN>(xc, i, v, gm[g]);
          }
        }
        CBD_TILE(h1, T + 1);
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          if (VEC_TILE_I * t + q >= S.fan0o) continue;
#pragma unroll
          for (int o = 0; o < 3; ++o) k0o[o] = fmaf(m[q], acc[3 * q + o], k0o[o]);
        }
        if (t == S.t0o - 1) {
          constexpr int SLOT0 = S.fan0o - VEC_TILE_I * (S.t0o - 1);     // block 0o's own mids in its last tile
#pragma unroll
          for (int g = 0; g < GUESTS; ++g)
#pragma unroll
            for (int o = 0; o < 3; ++o) {
              const float w = acc[3 * (SLOT0 + g) + o];
              k1e[3 * o + 0] = fmaf(gm[g][0], w, k1e[3 * o + 0]);
              k1e[3 * o + 1] = fmaf(gm[g][1], w, k1e[3 * o + 1]);
              k1e[3 * o + 2] = fmaf(gm[g][2], w, k1e[3 * o + 2]);
            }
        }
#pragma unroll
        for (int o = 0; o < 3; ++o) pin(k0o[o]);
      }
    }
  }

#undef CBD_TILE
  if constexpr (STAMPS) st_t3 = stamp();
  // ---- messages -> LDS (re-using the gathered-row tile, stride 34: reduce_runs.h),
  //      then run-length sum per aggregating node
  __syncthreads();   // every read of xT (mids) is complete before it is overwritten
#pragma unroll
  for (int r = 0; r < 16; ++r) xT[((r & 3) + 8 * (r >> 2) + 4 * hf) * OUT_STRIDE + j] = o0e[r];
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xT[(COL_1O + 3 * (3 * hf + o) + c) * OUT_STRIDE + j] = k1o[3 * o + c];
      if constexpr (OUT >= 2) xT[(COL_1E + 3 * (3 * hf + o) + c) * OUT_STRIDE + j] = k1e[3 * o + c];
    }
  if constexpr (OUT >= 3) {
#pragma unroll
    for (int o = 0; o < 3; ++o) xT[(COL_0O + 3 * hf + o) * OUT_STRIDE + j] = k0o[o];
  }
  __syncthreads();
  // Run-length sums per aggregating node (reduce_runs, tp_conv_dev.h)
  reduce_runs<NODE_STRIDE, OUT_STRIDE, S.out_dim, S.ntiles * 32>(xT, srcl, lane, S.out_dim, G.first_sum + (size_t)tile_local * NODE_STRIDE, G.last_sum + (size_t)tile_local * NODE_STRIDE,
              G.run_acc);
  if constexpr (STAMPS) {
    if (lane == 0 && args.stamps && blockIdx.x < 8192) {
      unsigned long long* o = args.stamps + (size_t)blockIdx.x * 8;
      o[0] = st_t0; o[1] = st_r0; o[2] = stamp(); o[3] = __builtin_amdgcn_s_memrealtime();
      o[4] = st_t1; o[5] = st_t2; o[6] = st_t3; o[7] = st_g0;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// segmented sum (fixed order) -> mean -> e3nn BatchNorm (eval) -> residual, reference tensor_layers.py:206-216.
//   out[n][c] = bn(sum_{edges into n} msg / max(deg[n],1)) + (c < in_dim ? node_in[n][c] : 0)
// The message sums were left by tp_conv as per-tile pieces (see the end of tp_conv_kernel); they are combined here group by
// group, tile by tile -- a fixed order, so results are bitwise reproducible.  bn_scale[c] = weight * rsqrt(running_var +
// eps) per column; bn_mean / bn_bias are zero outside the 0e columns.
__device__ __forceinline__ void finalize_one(const FinArgs& fa, const float* __restrict__ node_in, float* __restrict__ node_out,
                                             const float* __restrict__ bn_scale, const float* __restrict__ bn_mean,
                                             const float* __restrict__ bn_bias, int i, int c, int in_dim, int out_dim, int node_off) {
  const size_t o = (size_t)(i + node_off) * NODE_STRIDE + c;
  float r = 0.f;
  if (c < out_dim) {
    float sum = 0.f;
    int deg = 0;
    for (int g = 0; g < fa.n_groups; ++g) {
      const FinGroup& G = fa.g[g];
      const int k = G.node_mod > 0 ? i % G.node_mod : i;
      const int s = G.start[k], n = G.cnt[k];
      deg += G.deg_weight ? n : 0;
      if (n <= 0 || (G.col_hi > 0 && c >= G.col_hi)) continue;
      const int e = s + n, t0 = s / WAVE_EDGES, t1 = (e - 1) / WAVE_EDGES;
      const bool at_start = (s % WAVE_EDGES) == 0;
      if (t0 == t1) {
        if (at_start) sum += G.first_sum[(size_t)t0 * NODE_STRIDE + c];
        else if ((e % WAVE_EDGES) == 0) sum += G.last_sum[(size_t)t0 * NODE_STRIDE + c];
        else sum += G.run_acc[(size_t)(k + node_off) * NODE_STRIDE + c];
      } else {
        sum += (at_start ? G.first_sum : G.last_sum)[(size_t)t0 * NODE_STRIDE + c];
        for (int t = t0 + 1; t <= t1; ++t) sum += G.first_sum[(size_t)t * NODE_STRIDE + c];
      }
    }
    float m = sum / (float)(deg > 1 ? deg : 1);
    m = (m - bn_mean[c]) * bn_scale[c] + bn_bias[c];
    r = m + (c < in_dim ? node_in[o] : 0.f);
  }
  node_out[o] = r;
}

__global__ void conv_finalize_kernel(FinArgs fa, const float* __restrict__ node_in, float* __restrict__ node_out,
                                     const float* __restrict__ bn_scale, const float* __restrict__ bn_mean,
                                     const float* __restrict__ bn_bias, int n_nodes, int in_dim, int out_dim, int node_off) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = idx / NODE_STRIDE, c = idx % NODE_STRIDE;
  if (i >= n_nodes) return;
  finalize_one(fa, node_in, node_out, bn_scale, bn_mean, bn_bias, i, c, in_dim, out_dim, node_off);
}

// One layer of every co-scheduled batch in one launch.  kind (FinKind): FIN_EMB ligand rows with the embedding layer's slices;
// FIN_FIRST ligand rows + receptor rows with the shared layer-0 receptor group; FIN_MID ligand + receptor rows; FIN_LAST ligand
// rows only (the receptor rows of the last interaction layer are never read, reference quirk 3).
__device__ __forceinline__ const PoseBatch& fin_locate(const Multi& m, int& local) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < MAX_COSCHED; ++i) k = (i < m.n && (int)blockIdx.x >= m.off[i]) ? i : k;
  local = blockIdx.x - m.off[k];
  return *m.d[k];
}

__global__ void conv_finalize_multi_kernel(Multi mm, int kind, int xi_in, int xi_out, const float* __restrict__ bn_scale,
                                           const float* __restrict__ bn_mean, const float* __restrict__ bn_bias, int in_dim, int out_dim) {
  int blk;
  const PoseBatch& PB = fin_locate(mm, blk);
  const int idx = blk * blockDim.x + threadIdx.x;
  const int i = idx / NODE_STRIDE, c = idx % NODE_STRIDE;
  const bool roles = kind >= FIN_FIRST_R;          // bf16 role split: the cross / receptor groups come with a second 0e slice
  const int base = roles ? kind - (FIN_FIRST_R - FIN_FIRST) : kind;
  const int n0 = PB.B * PB.gs.Nl, n1 = (base == FIN_FIRST || base == FIN_MID) ? PB.B * PB.gs.Nr : 0;
  if (i >= n0 + n1) return;
  const float* node_in = PB.X[xi_in];
  float* node_out = PB.X[xi_out];
#ifdef CBD_EXPERIMENTS
  if (i < n0) finalize_one(base == FIN_EMB ? PB.fin_emb : roles ? PB.fin_lig_r : PB.fin_lig, node_in, node_out, bn_scale, bn_mean, bn_bias, i, c, in_dim,
                           out_dim, 0);
  else finalize_one(base == FIN_FIRST ? (roles ? PB.fin_rec_shared_r : PB.fin_rec_shared) : (roles ? PB.fin_rec_r : PB.fin_rec), node_in, node_out,
                    bn_scale, bn_mean, bn_bias, i - n0, c, in_dim, out_dim, PB.gs.rec_off);
#else
  if (i < n0) finalize_one(base == FIN_EMB ? PB.fin_emb : PB.fin_lig, node_in, node_out, bn_scale, bn_mean, bn_bias, i, c, in_dim, out_dim, 0);
  else finalize_one(base == FIN_FIRST ? PB.fin_rec_shared : PB.fin_rec, node_in, node_out, bn_scale, bn_mean, bn_bias, i - n0, c, in_dim, out_dim,
                    PB.gs.rec_off);
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Torsion head on the matrix cores (reference models/score_model.py:431-448, 650-664): one wave per (sample, rotatable
// bond); its <= 32 neighbour atoms (radius(lig_pos, bond_pos, 5), bond_nb_kernel) are the 32 MFMA columns.
//   per edge: attr = final_edge_embedding(gauss(d)); h = ReLU(W1 [attr | x_atom[:32] | (x_u + x_v)[:32]] + b1); w = W2 h + b2
//             msg[0:32] (0o) = sum_u w_B[u][.] d_B[u],  msg[32:64] (0e) = sum_u w_A[u][.] d_A[u]
//             d_A[u] = x1o[u] . T1 / sqrt(3) / sqrt(6), d_B[u] = x1e[u] . T1 / sqrt(3) / sqrt(6),
//             T1 = (3/sqrt2)(b b^T - I/3)(sqrt3 v)  (the two live e3nn paths, tests/test_kernel_math.py::tor_t1)
//   per bond: mean over its edges -> BatchNorm -> tor_final_layer -> * sqrt(torus score norm)
// Weight stream: 3 tiles of W1, then one tile per mid index (6 of path A, 6 of path B), 32 outputs each.
constexpr int BOND_TILES = 15;
__global__ __launch_bounds__(64) void bond_conv_kernel(BondHead h, Multi mm, int xi, const float* __restrict__ wstream, float tor_norm_sqrt) {
  __shared__ __attribute__((aligned(16))) float bias_l[BOND_TILES * 32];
  __shared__ float msg_l[64 * 33];
  __shared__ float s_feat[64];
  const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
  int bond;
  const PoseBatch& PB = fin_locate(mm, bond);
  const GraphStatic gs = PB.gs;
  const float* __restrict__ pos = PB.gd.pos;
  const float* __restrict__ node = PB.X[xi];
  const int* __restrict__ nb = PB.tor_nb;
  const int* __restrict__ nb_cnt = PB.tor_nb_cnt;
  float* __restrict__ tor_out = PB.tor_out;
  float* __restrict__ dbg_feat = PB.dbg_torfeat;
  const int ne = nb_cnt[bond];
  const int b = bond / gs.R, rho = bond % gs.R, Nl = gs.Nl;
  const float* P = pos + (size_t)b * Nl * 3;
  const int u = gs.rot_u[rho], v = gs.rot_v[rho];

  const GPtr<f32x4> gp = (GPtr<f32x4>)reinterpret_cast<const f32x4*>(wstream);   // wave-uniform stream base (tp_conv_dev.h: gemm_u)
  f32x4 a[KSTEPS / 4];
#pragma unroll
  for (int sg = 0; sg < KSTEPS / 4; ++sg) a[sg] = gp[sg * 64 + lane];
  {
    const f32x4* gb = reinterpret_cast<const f32x4*>(wstream + (size_t)(BOND_TILES + 1) * TILE_W_FLOATS);
    constexpr int NB4 = BOND_TILES * 8, NBI = (NB4 + 63) / 64;
#pragma unroll
    for (int i = 0; i < NBI; ++i) { const int k = lane + 64 * i; reinterpret_cast<f32x4*>(bias_l)[k < NB4 ? k : NB4 - 1] = gb[k < NB4 ? k : NB4 - 1]; }
  }
  const bool valid = j < ne;
  const int at = valid ? nb[(size_t)bond * 32 + j] : u;
  const float bx = (P[3 * u] + P[3 * v]) / 2, by = (P[3 * u + 1] + P[3 * v + 1]) / 2, bz = (P[3 * u + 2] + P[3 * v + 2]) / 2;
  float bux, buy, buz, bn, ux, uy, uz, d;
  {
    float x = P[3 * v] - P[3 * u], y = P[3 * v + 1] - P[3 * u + 1], z = P[3 * v + 2] - P[3 * u + 2];
    bn = sqrtf(x * x + y * y + z * z);
    const float inv = 1.0f / fmaxf(bn, 1e-12f);
    bux = x * inv; buy = y * inv; buz = z * inv;
    x = P[3 * at] - bx; y = P[3 * at + 1] - by; z = P[3 * at + 2] - bz;
    d = sqrtf(x * x + y * y + z * z);
    const float inv2 = 1.0f / fmaxf(d, 1e-12f);
    ux = x * inv2; uy = y * inv2; uz = z * inv2;
  }
  const float* xa = node + (size_t)(b * Nl + at) * NODE_STRIDE;
  const float* xu = node + (size_t)(b * Nl + u) * NODE_STRIDE;
  const float* xv = node + (size_t)(b * Nl + v) * NODE_STRIDE;
  // the 12 CG intermediates of this edge (zero for empty slots)
  float md[12];
  {
    const float bv = bux * ux + buy * uy + buz * uz;
    const float c = 3.6742346141747673f;   // (3/sqrt2) * sqrt3
    const float t1[3] = {c * (bux * bv - ux / 3.f), c * (buy * bv - uy / 3.f), c * (buz * bv - uz / 3.f)};
    const float k = valid ? 0.57735026918962576f * 0.40824829046386302f : 0.f;   // 1/sqrt3 * sqrt(1/6)
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const float* p = xa + (q < 6 ? COL_1O + 3 * q : COL_1E + 3 * (q - 6));
      md[q] = (p[0] * t1[0] + p[1] * t1[1] + p[2] * t1[2]) * k;
    }
  }
  // first-Linear input, lane half hf holds columns 16hf..16hf+15 of each part
  OpsF32::Act BxA;
  float (&Bx)[KSTEPS] = BxA.v;
  {
    // final_edge_embedding(GaussianSmearing(d)): every lane evaluates the hidden layer, then its 16 outputs
    float hid[32];
#pragma unroll
    for (int o = 0; o < 32; ++o) hid[o] = h.fe.part[o];
    for (int k = 0; k < 32; ++k) {
      const float t = d - h.fe.offset[k];
      const float gk = expf(h.fe.coeff * (t * t));
#pragma unroll
      for (int o = 0; o < 32; ++o) hid[o] = fmaf(h.fe.WgT[k * 32 + o], gk, hid[o]);
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) Bx[o] = h.fe.b1[16 * hf + o];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const float hk = fmaxf(hid[k], 0.f);
#pragma unroll
      for (int o = 0; o < 16; ++o) Bx[o] = fmaf(h.fe.W1T[k * 32 + 16 * hf + o], hk, Bx[o]);
    }
    const f32x4* pa = reinterpret_cast<const f32x4*>(xa + 16 * hf);
    const f32x4* pu = reinterpret_cast<const f32x4*>(xu + 16 * hf);
    const f32x4* pv = reinterpret_cast<const f32x4*>(xv + 16 * hf);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 s = pa[q], c0 = pu[q], c1 = pv[q];
      Bx[16 + 4 * q + 0] = s.x; Bx[16 + 4 * q + 1] = s.y; Bx[16 + 4 * q + 2] = s.z; Bx[16 + 4 * q + 3] = s.w;
      Bx[32 + 4 * q + 0] = c0.x + c1.x; Bx[32 + 4 * q + 1] = c0.y + c1.y; Bx[32 + 4 * q + 2] = c0.z + c1.z; Bx[32 + 4 * q + 3] = c0.w + c1.w;
    }
  }
  __syncthreads();
  int T = 0;
  f32x16 acc;
  OpsF32::Act h1;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    gemm_tile_u<OpsF32>(a, gp + (size_t)(T + 1) * (TILE_W_FLOATS / 4), lane, bias_l + T * 32, BxA, acc, hf);
    ++T;
    OpsF32::set_hidden(h1, m, acc);
  }
  float oA[16], oB[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) { oA[r] = 0.f; oB[r] = 0.f; }
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    gemm_tile_u<OpsF32>(a, gp + (size_t)(T + 1) * (TILE_W_FLOATS / 4), lane, bias_l + T * 32, h1, acc, hf);
    ++T;
    if (q < 6) {
#pragma unroll
      for (int r = 0; r < 16; ++r) oA[r] = fmaf(md[q], acc[r], oA[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) oB[r] = fmaf(md[q], acc[r], oB[r]);
    }
  }
  // feature order of the reference layer output: [32x0o (path B) | 32x0e (path A)]
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int w = (r & 3) + 8 * (r >> 2) + 4 * hf;
    msg_l[w * 33 + j] = oB[r];
    msg_l[(32 + w) * 33 + j] = oA[r];
  }
  __syncthreads();
  {
    float sum = 0.f;
    for (int k = 0; k < 32; ++k) sum += msg_l[lane * 33 + k];   // empty slots hold exact zeros
    const float mean = sum / (float)(ne > 1 ? ne : 1);
    const float f = (mean - h.bn_mean[lane]) * h.bn_scale[lane] + h.bn_bias[lane];
    s_feat[lane] = f;
    if (dbg_feat) dbg_feat[(size_t)bond * 64 + lane] = f;
  }
  __syncthreads();
  if (lane < 32) {
    float t = 0.f;
    for (int q = 0; q < 64; ++q) t = fmaf(h.tf_w0[lane * 64 + q], s_feat[q], t);
    t = tanhf(t) * h.tf_w1[lane];
    for (int off = 16; off > 0; off >>= 1) t += __shfl_xor(t, off);
    if (lane == 0) tor_out[bond] = t * tor_norm_sqrt;
  }
}

// m: B * R workgroups per batch
hipError_t launch_bond_conv(const BondHead& h, const Multi& m, int xi, const float* wstream, float tor_norm_sqrt, hipStream_t s) {
  if (m.off[m.n] <= 0) return hipSuccess;
  hipLaunchKernelGGL(bond_conv_kernel, dim3(m.off[m.n]), dim3(64), 0, s, h, m, xi, wstream, tor_norm_sqrt);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- host launchers
template <int IN, int OUT>
static hipError_t launch_one(const ConvArgs& a, int grid, hipStream_t s) {
  constexpr int lds_bytes = conv_lds_floats(conv_shape(IN, OUT, true).ntiles) * 4;
#ifdef CBD_DIAG
  // Diagnostic library only (tools/diag_lib.py builds experiments/libcbdock_diag.so with -DCBD_DIAG; the product library holds the
  // VAR = 0 kernels alone and never reads this variable).  CBD_CONV_VARIANT=8: the 74->74 kernel stamps s_memtime / s_memrealtime
  // (correct results); 14: non-temporal gathers (correct results); 9 .. 13: timing-only bounds with WRONG results -- 9: every tile
  // re-reads weight tile 0 (the L2 -> register weight stream becomes L1-resident: what a perfect weight-reuse scheme could gain),
  // 10 .. 12: run-coherent / skipped gathers, 13: the stamps of 8 on the gather pattern of 12.
  static const int var = getenv("CBD_CONV_VARIANT") ? atoi(getenv("CBD_CONV_VARIANT")) : 0;
  if (IN == 3 && var == 8) hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, (IN == 3 ? 8 : 0), OpsF32>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && var == 9) hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, (IN == 3 ? 9 : 0), OpsF32>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && var == 10) hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, (IN == 3 ? 10 : 0), OpsF32>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && var == 11) hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, (IN == 3 ? 11 : 0), OpsF32>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && var == 14) hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, (IN == 3 ? 14 : 0), OpsF32>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && var == 13) hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, (IN == 3 ? 13 : 0), OpsF32>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && var == 12) hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, (IN == 3 ? 12 : 0), OpsF32>), dim3(grid), dim3(64), lds_bytes, s, a);
  else
#endif
  hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, 0, OpsF32>), dim3(grid), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}

// The bf16-operand variant (cbd_set_option("bf16", 1), BASELINE.json configs[3]) lives in tp_conv_bf16.hip (64 edges per wave).
template <int IN, int OUT>
static hipError_t launch_one_x3(const ConvArgs& a, int grid, hipStream_t s) {
  constexpr int lds_bytes = conv_lds_floats(conv_shape(IN, OUT, true).ntiles) * 4;
  hipLaunchKernelGGL((tp_conv_kernel<IN, OUT, 0, OpsBf16x3>), dim3(grid), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}

hipError_t launch_tp_conv_x3(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s) {
  if (grid <= 0) return hipSuccess;
  if (in_level == 0 && out_level == 1) return launch_one_x3<0, 1>(a, grid, s);
  if (in_level == 1 && out_level == 2) return launch_one_x3<1, 2>(a, grid, s);
  if (in_level == 2 && out_level == 3) return launch_one_x3<2, 3>(a, grid, s);
  if (in_level == 3 && out_level == 3) return launch_one_x3<3, 3>(a, grid, s);
  return hipErrorInvalidValue;
}

hipError_t launch_tp_conv(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s) {
  if (grid <= 0) return hipSuccess;
  if (in_level == 0 && out_level == 1) return launch_one<0, 1>(a, grid, s);
  if (in_level == 1 && out_level == 2) return launch_one<1, 2>(a, grid, s);
  if (in_level == 2 && out_level == 3) return launch_one<2, 3>(a, grid, s);
  if (in_level == 3 && out_level == 3) return launch_one<3, 3>(a, grid, s);
  return hipErrorInvalidValue;
}

hipError_t launch_conv_finalize(const FinArgs& fa, const float* node_in, float* node_out, const float* bn_scale,
                                const float* bn_mean, const float* bn_bias, int n_nodes, int in_dim, int out_dim,
                                int node_off, hipStream_t s) {
  if (n_nodes <= 0) return hipSuccess;
  const int total = n_nodes * NODE_STRIDE;
  hipLaunchKernelGGL(conv_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, fa, node_in, node_out, bn_scale,
                     bn_mean, bn_bias, n_nodes, in_dim, out_dim, node_off);
  return hipGetLastError();
}

// m: ceil((n0 + n1) * NODE_STRIDE / 256) workgroups per batch with n0 = B * Nl and n1 = B * Nr for the kinds that include receptor rows
hipError_t launch_conv_finalize_multi(const Multi& m, int kind, int xi_in, int xi_out, const float* bn_scale, const float* bn_mean,
                                      const float* bn_bias, int in_dim, int out_dim, hipStream_t s) {
  if (m.off[m.n] <= 0) return hipSuccess;
  hipLaunchKernelGGL(conv_finalize_multi_kernel, dim3(m.off[m.n]), dim3(256), 0, s, m, kind, xi_in, xi_out, bn_scale, bn_mean, bn_bias,
                     in_dim, out_dim);
  return hipGetLastError();
}

}  // namespace cbd
