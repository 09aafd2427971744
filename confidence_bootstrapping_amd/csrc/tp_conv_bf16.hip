// bf16-operand tensor-product message passing for gfx950, second generation (BASELINE.json configs[3]; cbd_set_option("bf16", 1)).
//
// Same math as tp_conv.hip (FCBlock -> FasterTensorProduct -> segmented sum, reference models/tensor_layers.py:195-206,66-117),
// re-tiled for the bf16 matrix pipe, where the fp32-era layout is no longer bound by the MFMAs (one 32x32x16 bf16 MFMA is 8 passes =
// 32 cycles against 64 for the fp32 k=2 form):
//   * ONE WAVE OWNS 64 EDGES: every weight fragment is used for two MFMAs (two 32-edge sub-tiles, two independent accumulator
//     chains), which halves the weight stream per FLOP and the per-tile fixed costs;
//   * the bias (fp32) is the C operand of a tile's first MFMA pair, kept in 16 registers that are re-loaded in place: 6 k-steps and
//     6 KB per tile, no accumulator initialisation (an earlier version spent a 7th k-step on it);
//   * what bounds this kernel is not the matrix core but everything next to it (in-kernel stamps and timing-only diagnostics,
//     CBD_BF16_DIAG=1..4): every VALU instruction costs matrix-pipe time, so the CG epilogue was trimmed --
//     -fno-slp-vectorize (packed fp32 FMAs are the most expensive kind), scalar x direction mids factored out of the sums, one-
//     instruction ReLU, padded slots skipped -- and the per-wave latency phases shortened: gathers in two rounds, vector-block tile
//     loops fully unrolled so that mids are compile-time constants and their LDS reads batch, run-length reduction with scalar run
//     masks.  Round 1 137.6 -> round 2 195 poses/s on C4 (64 x 40), 0.27 -> 0.39 of the bf16 peak.  Tried and NOT kept: sharing every
//     tile between the 4 or 8 waves of a workgroup through an LDS ring with one barrier per tile (122 / 108 poses/s), 4 independent
//     waves per workgroup (no change), 128 edges per wave at one wave per SIMD (114), global instead of FLAT loads (no change), wave
//     priorities (round 3: static s_setprio by workgroup number, raised inside or outside the MFMA chains: 196.3-199.1 vs 197.7-198.2
//     poses/s, profiles/r03_b_prio_sweep.txt -- the two residents of a SIMD are not phase-locked), and again round 3: workgroup-shared
//     tiles through an LDS ring filled by LDS-DMA with one barrier per tile (8-wave workgroups: 159 vs 214 poses/s,
//     profiles/r03_k_bf16_shared_weights_experiment.txt);
//   * LDS per wave = two transposed row tiles of 9.8 KB (the gathered destination rows, later the message tiles): 8 waves per CU.
// fp32 everywhere outside the two Linears (gathered rows, CG contraction, messages, reduction), like the first generation.
// Weight stream (pack_conv_stream_bf16, engine.hip): (ntiles + 1) tiles of [6 k-steps][64 lanes][8 bf16] = 6 KB, then the fp32 bias
// rows [ntiles + 1][32].
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "tp_conv_dev.h"
#include "tp_conv_bf16_dev.h"

namespace cbd {

// DIAG (timing only, WRONG results; CBD_BF16_DIAG=n): 1 = every tile re-reads weight tile 0 (the weight stream becomes L1-resident),
// 2 = no CG epilogue (the accumulators are only summed up), 3 = both; 4 = correct results + phase stamps (tools/conv_clock.py bf16);
// 8 = the bias registers are never re-loaded, 9 = 8 + 1, 16 = no weight or bias re-loads at all (the first tile's registers serve every
// tile: the kernel without its weight stream), 24 = 16 + 8, 32 = the bias bpermutes are not waited for
template <int IN, int OUT, int DIAG = 0>
__global__ __launch_bounds__(64, 2) void tp_conv64_kernel(ConvArgs args) {
  constexpr ConvShape S = conv_shape(IN, OUT, true);   // merged vector tails (common.h), as in tp_conv.hip
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const xT0 = lds;                                   // sub-tile 0: [col][32] gathered rows, later [col][34] messages
  float* const xT1 = lds + V2_SUB_FLOATS;
  int* const srcl = reinterpret_cast<int*>(lds + 2 * V2_SUB_FLOATS);   // [2][32]
  const int lane = threadIdx.x;
  const int j = lane & 31, hf = lane >> 5;
  const int lane4hf = 16 * hf;   // byte address (4 x lane) of lane 4 hf: base of the bias bpermutes (v2_gemm)

  // ---- which group / edge range does this wave own?  (edge counts live on the device; a wave owns 64 edges = two reduction tiles)
  int grp = -1, e0 = 0, cnt = 0, tile_local = 0;
  {
    int wave_in_group = 0;
    if (!find_group(args, blockIdx.x, lane, 64, grp, wave_in_group, cnt)) return;
    e0 = wave_in_group * 64;
    tile_local = 2 * wave_in_group;
  }
  const ConvGroup G = args.g[grp];
  unsigned long long st_t0 = 0, st_r0 = 0, st_t1 = 0, st_t2 = 0, st_t3 = 0, st_0e = 0;
  if constexpr (DIAG == 4) { st_t0 = stamp(); st_r0 = __builtin_amdgcn_s_memrealtime(); }

  // ---- start the weight stream
  const GFrag gp = (GFrag)reinterpret_cast<const bf16x8*>(G.wstream);   // uniform; tile T fragment q of this lane: gp[T * V2_TILE_FRAGS + q * 64 + lane]
  bf16x8 a[V2_NFRAG];
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) a[q] = gp[q * 64 + lane];
  // fp32 bias rows behind the (ntiles + 1) tiles: [ntiles + 1][32]; this lane half's float4s are 2q' + hf
  const GBias gbias = (GBias)reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(G.wstream) + (size_t)(S.ntiles + 1) * V2_TILE_FRAGS);   // uniform
  f32x16 cb;
  {
    const GPtr<f32x4> gb4 = (GPtr<f32x4>)gbias;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const f32x4 b = gb4[hf + 2 * qq];
      cb[4 * qq + 0] = b.x; cb[4 * qq + 1] = b.y; cb[4 * qq + 2] = b.z; cb[4 * qq + 3] = b.w;
    }
  }

  // ---- gather both sub-tiles.  Lanes past the end of the group read the group's last edge (unconditional loads) and are dropped
  //      at the end through src = -1.
  Act6 Bx0, Bx1;
  float v0[3], v1[3];
  // Two rounds of memory latency instead of four: the edge indices of BOTH sub-tiles first, then every gather that depends on them in
  // one batch (attributes, the two 32-column node segments, sub-tile 0's full destination row: 136 registers in flight), sub-tile 1's
  // row while the first is transposed into LDS.  (In source order per sub-tile the compiler waited for each sub-tile's indices and
  // gathers in turn: 21.5 k cycles of a 113 k lifetime, in-kernel stamps.)
  int src_r[2], dstn[2], aidx[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const int e = e0 + 32 * sub + j;
    const bool valid = e < cnt;
    const int ec = valid ? e : cnt - 1;
    src_r[sub] = G.src[ec]; dstn[sub] = G.dst[ec]; aidx[sub] = G.attr_idx[ec];
    const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[ec];
    if (sub) { v1[0] = vv.x; v1[1] = vv.y; v1[2] = vv.z; } else { v0[0] = vv.x; v0[1] = vv.y; v0[2] = vv.z; }
    if (hf == 0) srcl[32 * sub + j] = valid ? src_r[sub] : -1;
  }
  __builtin_amdgcn_sched_barrier(0);
  f32x4 ta[2][4], ts[2][4], td[2][4], tr[10];
  const f32x4* pr[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx[sub] * 32 + 16 * hf);
    const f32x4* ps = reinterpret_cast<const f32x4*>(G.node_in + (size_t)src_r[sub] * NODE_STRIDE + 16 * hf);
    const f32x4* pd = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dstn[sub] * NODE_STRIDE + 16 * hf);
    // full destination row; lane half hf takes cols 40hf .. 40hf+39 (cols 76..79 are the row's padding: loaded, not stored)
    pr[sub] = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dstn[sub] * NODE_STRIDE + 40 * hf);
#pragma unroll
    for (int q = 0; q < 4; ++q) { ta[sub][q] = pa[q]; ts[sub][q] = ps[q]; td[sub][q] = pd[q]; }
  }
#pragma unroll
  for (int q = 0; q < 10; ++q) tr[q] = pr[0][q];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    Act6& Bx = sub ? Bx1 : Bx0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v2_set_in(Bx, 0, q, ta[sub][q]);
      v2_set_in(Bx, 1, q, ts[sub][q]);
      v2_set_in(Bx, 2, q, td[sub][q]);
    }
  }
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    float* xT = sub ? xT1 : xT0;
#pragma unroll
    for (int q = 0; q < 10; ++q) {   // transposed LDS copy xT[col][j]
      const f32x4 r = tr[q];
      if (sub == 0) tr[q] = pr[1][q];
      if (40 * hf + 4 * q < 76) {
        float* o = xT + (40 * hf + 4 * q) * 32 + j;
        o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
      }
    }
  }
  __syncthreads();   // single-wave workgroup: orders the LDS writes above before the reads below
  if constexpr (DIAG == 4) st_t1 = stamp();

  int T = 0;
  f32x16 acc0, acc1;
  Act6 h0, h1;
  const int i_lo = G.i0e_lo, i_hi = G.i0e_hi;
  const bool vec_on = G.vec_on != 0;
  const int T_vec = 3 + S.t0e;
  // The bias rows are requested TWO tiles ahead (every chain names the tile after its successor: NEXT2).  This wave's sequence: first
  // Linear 0..2, the 0e tiles [3 + i_lo, 3 + i_hi), the vector blocks if vec_on, then the zero tile S.ntiles.  All selects, no branches.
  const int past0e = vec_on ? T_vec : S.ntiles;                                  // first tile behind the 0e block
  const int first0e = i_lo < i_hi ? 3 + i_lo : past0e;                           // tile behind the first Linear
  const int past0e_1 = vec_on ? (T_vec + 1 < S.ntiles ? T_vec + 1 : S.ntiles) : S.ntiles;   // ... and the one behind `past0e`
  const int second0e = i_lo < i_hi ? (i_lo + 1 < i_hi ? 4 + i_lo : past0e) : past0e_1;     // tile behind `first0e`
  float raw_next = gbias[32 + (lane & 31)];   // bias of tile 1
#define V2_TILE(BA, BB, NEXT, NEXT2)                                               \
  {                                                                         \
    const int tn_ = (NEXT);                                                 \
    v2_gemm<DIAG>(a, cb, gp + ((DIAG & 1) ? (size_t)0 : (size_t)tn_ * V2_TILE_FRAGS), gbias + (size_t)(NEXT2) * 32, raw_next, lane, lane4hf, BA, BB, acc0, acc1); \
    T = tn_;                                                                \
  }
  // ---- first Linear (3 tiles): h = ReLU(W1 x + b1), kept in the C/D register layout = B operand of the second Linear
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    V2_TILE(Bx0, Bx1, m < 2 ? T + 1 : first0e, m == 0 ? 2 : (m == 1 ? first0e : second0e));
    if (m == 2) mfma_operand_guard();   // the first-Linear operands die here without a refill
    v2_set_hidden(h0, m, acc0);
    v2_set_hidden(h1, m, acc1);
    if constexpr (!(DIAG & 40)) bias_ready(cb);
  }

  if constexpr (DIAG == 4) st_t2 = stamp();
  const float* xc0 = xT0 + j;
  const float* xc1 = xT1 + j;
  // ---- block 0e: one tile per mid index, 32 output scalars
  // The 0e tiles carry NO bias: sum_i m_i (w_i + b_i) = sum_i m_i w_i + sum_i b_i m_i, and the second sum is ONE small matrix product
  // per block -- A = the block's bias rows as a [32 outputs x 48 mids] bf16 tile (3 fragments behind the bias table of the stream),
  // B = the edge's mids -- accumulated straight into the output registers, which have the accumulator layout.  38 of the 57 tiles
  // then need neither the 16 bias registers' re-load nor its wait (a per-tile bias costs 11 % of the kernel whichever way it is
  // brought in: four broadcast dwordx4 loads or one dword + 16 ds_bpermute; timing-only diagnostics).
  f32x16 o0e0 = {}, o0e1 = {};
  bf16x8 ab0e[3];
  if (i_lo < i_hi) {
    const GFrag gb0e = (GFrag)reinterpret_cast<const bf16x8*>(reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(G.wstream) + (size_t)(S.ntiles + 1) * V2_TILE_FRAGS) + (size_t)(S.ntiles + 1) * 32);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) ab0e[s3] = gb0e[s3 * 64 + lane];
    if (vec_on && !(DIAG & 8)) {   // the vector blocks behind this one still take their bias as the C operand: request it now
      const GPtr<f32x4> gb4 = (GPtr<f32x4>)(gbias + (size_t)T_vec * 32);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 b = gb4[hf + 2 * qq];
        cb[4 * qq + 0] = b.x; cb[4 * qq + 1] = b.y; cb[4 * qq + 2] = b.z; cb[4 * qq + 3] = b.w;
      }
      raw_next = gbias[(size_t)past0e_1 * 32 + (lane & 31)];
    }
  }
  // the mids are read from LDS BEFORE the MFMA chain (the fences inside v2_gemm keep the reads above it, their wait lands at the first
  // use below it): the LDS latency is covered by the MFMAs instead of being exposed after them.  Two loops, one per kind of mid (the
  // scalar features themselves, then the 1o . direction dot products), so that neither body branches on the mid index.
  // Software-pipelined (two accumulator sets A = acc0/acc1 and B = accB0/accB1; this block has ~70 registers of headroom below the
  // vector blocks' peak): the chain of tile i + 1 carries the epilogue of tile i between its MFMA pairs.
  f32x16 accB0, accB1;
  const int EPI_LO[V2_NFRAG + 1] = {0, 3, 6, 9, 12, 14, 16};
  auto next_of = [&](int i) { return i + 1 < i_hi ? 4 + i : past0e; };                              // stream index of the tile after 0e tile i
  // (macros, not lambdas taking the accumulators by reference: through reference parameters hipcc keeps all four accumulator tuples
  //  in scratch memory)
#define V2_EPI_ALL(Y0, Y1, M0, M1)                                                                                   \
  {                                                                                                                  \
    if constexpr (DIAG & 2) { o0e0[0] += Y0[0] + M0; o0e1[0] += Y1[0] + M1; } else {                                 \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { o0e0[r] = fmaf(M0, Y0[r], o0e0[r]); o0e1[r] = fmaf(M1, Y1[r], o0e1[r]); } \
    }                                                                                                                \
  }
#define V2_CHAIN_PLAIN(I, X0, X1)                                                                                    \
  {                                                                                                                  \
    const int tn_ = next_of(I);                                                                                      \
    v2_gemm_p<DIAG, false>(a, cb, gp + ((DIAG & 1) ? (size_t)0 : (size_t)tn_ * V2_TILE_FRAGS), gbias, raw_next, lane, lane4hf, h0, h1, \
                    X0, X1, [](int) {});                                                                             \
    T = tn_;                                                                                                         \
  }
#define V2_CHAIN_EPI(I, X0, X1, Y0, Y1, M0, M1)                                                                      \
  {                                                                                                                  \
    const int tn_ = next_of(I);                                                                                      \
    v2_gemm_p<DIAG, false>(a, cb, gp + ((DIAG & 1) ? (size_t)0 : (size_t)tn_ * V2_TILE_FRAGS), gbias, raw_next, lane, lane4hf, h0, h1, \
                    X0, X1, [&](int q) __attribute__((always_inline)) {                                              \
                      if constexpr (DIAG & 2) { if (q == 0) { o0e0[0] += Y0[0] + M0; o0e1[0] += Y1[0] + M1; } } else { \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                               \
                          if (r >= EPI_LO[q] && r < EPI_LO[q + 1]) { o0e0[r] = fmaf(M0, Y0[r], o0e0[r]); o0e1[r] = fmaf(M1, Y1[r], o0e1[r]); } \
                      }                                                                                              \
                    });                                                                                              \
    T = tn_;                                                                                                         \
  }
  // tiles [IB, IE) whose mids come from MID(i, m0, m1) (LDS reads, issued one chain before they are used)
#define V2_RUN0E(IB, IE, MID)                                                                                        \
  if ((IB) < (IE)) {                                                                                                 \
    float ma0, ma1, mb0 = 0.f, mb1 = 0.f;                                                                            \
    MID((IB), ma0, ma1);                                                                                             \
    V2_CHAIN_PLAIN((IB), acc0, acc1);                                                                                \
    const int pairs_ = ((IE) - (IB) - 1) >> 1;   /* double steps: set B then set A again */                          \
    int i = (IB);                                                                                                    \
    _Pragma("unroll 1") for (int p_ = 0; p_ < pairs_; ++p_) {                                                        \
      MID(i + 1, mb0, mb1);                                                                                          \
      V2_CHAIN_EPI(i + 1, accB0, accB1, acc0, acc1, ma0, ma1);                                                       \
      MID(i + 2, ma0, ma1);                                                                                          \
      V2_CHAIN_EPI(i + 2, acc0, acc1, accB0, accB1, mb0, mb1);                                                       \
      i += 2;                                                                                                        \
    }                                                                                                                \
    if (i + 1 < (IE)) {   /* one tile left: set B, then drain it */                                                  \
      MID(i + 1, mb0, mb1);                                                                                          \
      V2_CHAIN_EPI(i + 1, accB0, accB1, acc0, acc1, ma0, ma1);                                                       \
      V2_EPI_ALL(accB0, accB1, mb0, mb1);                                                                            \
    } else {                                                                                                         \
      V2_EPI_ALL(acc0, acc1, ma0, ma1);                                                                              \
    }                                                                                                                \
  }
#define V2_MID_SCALAR(I, M0, M1) { M0 = xc0[(I) * 32]; M1 = xc1[(I) * 32]; }
#define V2_MID_DOT(I, M0, M1)                                                                                        \
  {                                                                                                                  \
    const float* p0_ = xc0 + (COL_1O + 3 * ((I) - NS)) * 32;                                                          \
    const float* p1_ = xc1 + (COL_1O + 3 * ((I) - NS)) * 32;                                                          \
    M0 = p0_[0] * v0[0] + p0_[32] * v0[1] + p0_[64] * v0[2];                                                         \
    M1 = p1_[0] * v1[0] + p1_[32] * v1[1] + p1_[64] * v1[2];                                                         \
  }
  const int i_mid = i_hi < NS ? i_hi : NS;
  V2_RUN0E(i_lo, i_mid, V2_MID_SCALAR);
  if constexpr (IN >= 1) {
    const int i_b = i_lo > NS ? i_lo : NS;
    V2_RUN0E(i_b, i_hi, V2_MID_DOT);
  }
  if (i_lo < i_hi) {
    // sum_i b_i m_i for the mids [i_lo, i_hi) of this wave (a virtual slice of a group runs only part of the 0e tiles)
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      bf16x8 bm0, bm1;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int i = 16 * s3 + 8 * hf + jj;       // mid index of element jj of this lane half's k-slice (runtime through hf only)
        float m0 = 0.f, m1 = 0.f;
        if (s3 < 2) { m0 = xc0[i * 32]; m1 = xc1[i * 32]; }
        else if constexpr (IN >= 1) {
          if (jj < S.n1o) {                         // mids 32 .. 37 live in the lower lane half's slice (i = 32 + jj); the upper half's are padding
            const float* p0 = xc0 + (COL_1O + 3 * jj) * 32;
            const float* p1 = xc1 + (COL_1O + 3 * jj) * 32;
            m0 = p0[0] * v0[0] + p0[32] * v0[1] + p0[64] * v0[2];
            m1 = p1[0] * v1[0] + p1[32] * v1[1] + p1[64] * v1[2];
          }
          if (hf) { m0 = 0.f; m1 = 0.f; }
        }
        const bool in = i >= i_lo && i < i_hi;
        bm0[jj] = (__bf16)(in ? m0 : 0.f);
        bm1[jj] = (__bf16)(in ? m1 : 0.f);
      }
      o0e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab0e[s3], bm0, o0e0, 0, 0, 0);
      o0e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab0e[s3], bm1, o0e1, 0, 0, 0);
    }
  }
#undef V2_RUN0E
#undef V2_CHAIN_EPI
#undef V2_CHAIN_PLAIN
#undef V2_EPI_ALL
#undef V2_MID_SCALAR
#undef V2_MID_DOT

  if constexpr (DIAG == 4) st_0e = stamp();
  // ---- vector / pseudoscalar blocks: tile = 5 mid indices x 6 outputs; lane half hf owns outputs 3hf..3hf+2
  float k1o0[9], k1e0[9], k0o0[3], k1o1[9], k1e1[9], k0o1[3];
#pragma unroll
  for (int r = 0; r < 9; ++r) { k1o0[r] = 0.f; k1e0[r] = 0.f; k1o1[r] = 0.f; k1e1[r] = 0.f; }
  k0o0[0] = k0o0[1] = k0o0[2] = 0.f;
  k0o1[0] = k0o1[1] = k0o1[2] = 0.f;

  // The tile loops of the vector blocks are FULLY unrolled: the mid index is then a compile-time constant, the kind of every mid
  // (scalar x direction, copy, cross product, padding) is resolved by the compiler, and the LDS reads of a tile's ten mids are issued
  // together before the chain.  Rolled, every mid was a chain of scalar branches around LDS reads that were each waited for in turn:
  // 44 s_waitcnt and ~3.6 k cycles per vector tile against ~0.7 k for a scalar tile (in-kernel stamps, round 2).
  // Mids of the form (scalar feature) x (edge direction) -- the first NS mids of block 1o, the last n0o of block 1e -- are not
  // multiplied out: sum_i (x_i v_c) w_io = v_c sum_i x_i w_io, so they cost one FMA per output instead of three and the direction is
  // applied once per block (`is_scalar(i)` / `scalar_of(x, i)` describe them; padded slots i >= fan are skipped altogether).
  auto vec_block = [&](auto mid_fn, auto is_scalar, auto scalar_of, auto ntile_c, auto fan_c, float (&keep0)[9], float (&keep1)[9])
                       __attribute__((always_inline)) {
    constexpr int ntile = decltype(ntile_c)::value, fan = decltype(fan_c)::value;
    float s0[3] = {0.f, 0.f, 0.f}, s1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < ntile; ++t) {
      // vector-valued mids are evaluated before the chain (LDS latency and the cross products under the MFMAs); the scalar ones are
      // read behind it -- ten more live registers across the chain would spill
      float ma[VEC_TILE_I][3], mb[VEC_TILE_I][3];
      if constexpr (!(DIAG & 2)) {
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          const int i = VEC_TILE_I * t + q;
          if (i >= fan || is_scalar(i)) continue;
          mid_fn(xc0, i, v0, ma[q]); mid_fn(xc1, i, v1, mb[q]);
        }
      }
      V2_TILE(h0, h1, T + 1, T + 2 < S.ntiles ? T + 2 : S.ntiles);
      if constexpr (DIAG & 2) { keep0[0] += acc0[0]; keep1[0] += acc1[0]; if constexpr (!(DIAG & 40)) bias_ready(cb); continue; }
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const int i = VEC_TILE_I * t + q;
        if (i >= fan) continue;
        float xa = 0.f, xb = 0.f;
        if (is_scalar(i)) { xa = scalar_of(xc0, i); xb = scalar_of(xc1, i); }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const float wa = acc0[3 * q + o], wb = acc1[3 * q + o];
          if (is_scalar(i)) {
            s0[o] = fmaf(xa, wa, s0[o]);
            s1[o] = fmaf(xb, wb, s1[o]);
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              keep0[3 * o + c] = fmaf(ma[q][c], wa, keep0[3 * o + c]);
              keep1[3 * o + c] = fmaf(mb[q][c], wb, keep1[3 * o + c]);
            }
          }
        }
      }
      // the tile's FMAs have no side effect, so nothing ties them to this place in the fully unrolled code: without the pins hipcc
      // parks the accumulators of the scalar tiles in scratch and does their FMAs several tiles later (1 KB of spills per lane)
#pragma unroll
      for (int o = 0; o < 3; ++o) { pin(s0[o]); pin(s1[o]); }
      if constexpr (!(DIAG & 40)) bias_ready(cb);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        keep0[3 * o + c] = fmaf(v0[c], s0[o], keep0[3 * o + c]);
        keep1[3 * o + c] = fmaf(v1[c], s1[o], keep1[3 * o + c]);
      }
  };
  if (vec_on) {
    vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, vv, m); },
              [](int i) { return i < NS; }, [](const float* x, int i) { return x[i * 32]; },
              std::integral_constant<int, S.t1o>{}, std::integral_constant<int, S.fan1o>{}, k1o0, k1o1);
    // merged tails (ConvShape::vmerged): block 1e runs its first 5 (t1e - 1) mids; the others are guests of block 0o's last tile
    constexpr int OWN1E = S.vmerged ? VEC_TILE_I * (S.t1e - 1) : S.fan1e, GUESTS = S.fan1e - OWN1E;
    if constexpr (OUT >= 2)
      vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, vv, m); },
                [](int i) { return i >= S.n1o + S.n1e; }, [](const float* x, int i) { return x[(COL_0O + (i - S.n1o - S.n1e)) * 32]; },
                std::integral_constant<int, S.t1e - S.vmerged>{}, std::integral_constant<int, OWN1E>{}, k1e0, k1e1);
    if constexpr (OUT >= 3) {
#pragma unroll
      for (int t = 0; t < S.t0o; ++t) {
        float ma[VEC_TILE_I], mb[VEC_TILE_I];
        float ga[GUESTS > 0 ? GUESTS : 1][3], gb[GUESTS > 0 ? GUESTS : 1][3];     // block 1e's tail mids of the two sub-tiles
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          if (VEC_TILE_I * t + q >= S.fan0o) continue;
          ma[q] = mid0o<IN>(xc0, VEC_TILE_I * t + q, v0); mb[q] = mid0o<IN>(xc1, VEC_TILE_I * t + q, v1);
        }
        if (t == S.t0o - 1) {
#pragma unroll
          for (int g = 0; g < GUESTS; ++g) {
            const int i = OWN1E + g;
            if (i >= S.n1o + S.n1e) {
              const float xa = xc0[(COL_0O + (i - S.n1o - S.n1e)) * 32], xb = xc1[(COL_0O + (i - S.n1o - S.n1e)) * 32];
#pragma unroll
              for (int c = 0; c < 3; ++c) { ga[g][c] = xa * v0[c]; gb[g][c] = xb * v1[c]; }
            } else { mid1e<IN>(xc0, i, v0, ga[g]); mid1e<IN>(xc1, i, v1, gb[g]); }
          }
        }
        V2_TILE(h0, h1, T + 1, T + 2 < S.ntiles ? T + 2 : S.ntiles);
        if constexpr (DIAG & 2) { k0o0[0] += acc0[0]; k0o1[0] += acc1[0]; if constexpr (!(DIAG & 40)) bias_ready(cb); continue; }
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          if (VEC_TILE_I * t + q >= S.fan0o) continue;
#pragma unroll
          for (int o = 0; o < 3; ++o) { k0o0[o] = fmaf(ma[q], acc0[3 * q + o], k0o0[o]); k0o1[o] = fmaf(mb[q], acc1[3 * q + o], k0o1[o]); }
        }
        if (t == S.t0o - 1) {
          constexpr int SLOT0 = S.fan0o - VEC_TILE_I * (S.t0o - 1);
#pragma unroll
          for (int g = 0; g < GUESTS; ++g)
#pragma unroll
            for (int o = 0; o < 3; ++o) {
              const float wa = acc0[3 * (SLOT0 + g) + o], wb = acc1[3 * (SLOT0 + g) + o];
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                k1e0[3 * o + c] = fmaf(ga[g][c], wa, k1e0[3 * o + c]);
                k1e1[3 * o + c] = fmaf(gb[g][c], wb, k1e1[3 * o + c]);
              }
            }
        }
        if constexpr (!(DIAG & 40)) bias_ready(cb);
      }
    }
  }
#undef V2_TILE

  // ---- messages -> LDS (re-using the gathered-row tiles, stride 34), then run-length sums per aggregating node and sub-tile
  if constexpr (DIAG == 4) st_t3 = stamp();
  __syncthreads();   // every read of xT (mids) is complete before it is overwritten
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    float* xT = sub ? xT1 : xT0;
    const f32x16& o0e = sub ? o0e1 : o0e0;
    const float* k1o = sub ? k1o1 : k1o0;
    const float* k1e = sub ? k1e1 : k1e0;
    const float* k0o = sub ? k0o1 : k0o0;
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
  }
  __syncthreads();
  // Run-length sums per aggregating node and 32-edge reduction tile, exactly the pieces of tp_conv_kernel (reduce_runs, tp_conv_dev.h)
  // a virtual slice writes only the columns it produces (role split, engine.hip: a 0e-only slice the 32 scalar columns, the vector
  // slice the rest -- the two share the group's piece buffers; a second 0e slice has buffers of its own)
  const int col_lo = (vec_on && i_lo >= i_hi) ? NS : 0, col_hi = vec_on ? S.out_dim : NS;
#pragma unroll 1
  for (int sub = 0; sub < 2; ++sub)
    reduce_runs<NODE_STRIDE, OUT_STRIDE, S.out_dim, V2_SUB_FLOATS>(sub ? xT1 : xT0,      // (tile bases 0 and V2_SUB_FLOATS: both even)
                                                                  srcl + 32 * sub, lane, col_hi, G.first_sum + (size_t)(tile_local + sub) * NODE_STRIDE,
                G.last_sum + (size_t)(tile_local + sub) * NODE_STRIDE, G.run_acc, col_lo);
  if constexpr (DIAG == 4) {   // same record layout as tp_conv_kernel's CBD_CONV_VARIANT=8 stamps (tools/conv_clock.py)
    if (lane == 0 && args.stamps && blockIdx.x < 8192) {
      unsigned long long* o = args.stamps + (size_t)blockIdx.x * 8;
      o[0] = st_t0; o[1] = st_r0; o[2] = stamp(); o[3] = __builtin_amdgcn_s_memrealtime();
      o[4] = st_t1; o[5] = st_t2; o[6] = st_t3; o[7] = st_0e;   // slot 7: end of the 0e block (fp32 kernel: end of the first first-Linear tile)
    }
  }
}

template <int IN, int OUT>
static hipError_t launch_one64(const ConvArgs& a, int grid, hipStream_t s) {
  constexpr int lds_bytes = (2 * V2_SUB_FLOATS + 64) * 4;
#ifdef CBD_DIAG      // diagnostic library only (tools/diag_lib.py): timing-only variants with WRONG results, 4 = phase stamps
  static const int diag = getenv("CBD_BF16_DIAG") ? atoi(getenv("CBD_BF16_DIAG")) : 0;
  if (IN == 3 && diag == 1) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 1 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && diag == 2) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 2 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && diag == 4) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 4 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && diag == 3) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 3 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && diag == 8) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 8 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && diag == 9) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 9 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && diag == 24) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 24 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && diag == 32) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 32 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else if (IN == 3 && diag == 26) hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT, (IN == 3 ? 26 : 0)>), dim3(grid), dim3(64), lds_bytes, s, a);
  else
#endif
  hipLaunchKernelGGL((tp_conv64_kernel<IN, OUT>), dim3(grid), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}

// grid: number of 64-edge waves (sum over groups of ceil(cap / 64))
hipError_t launch_tp_conv_bf16(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s) {
  if (grid <= 0) return hipSuccess;
  if (in_level == 0 && out_level == 1) return launch_one64<0, 1>(a, grid, s);
  if (in_level == 1 && out_level == 2) return launch_one64<1, 2>(a, grid, s);
  if (in_level == 2 && out_level == 3) return launch_one64<2, 3>(a, grid, s);
  if (in_level == 3 && out_level == 3) return launch_one64<3, 3>(a, grid, s);
  return hipErrorInvalidValue;
}

}  // namespace cbd
