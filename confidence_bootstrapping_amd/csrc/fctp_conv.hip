// Fused message passing of the all-atom CONFIDENCE model for gfx950 (MI355X):
//   per edge:  h = ReLU(W1 [edge_attr | x_src[:24] | x_dst[:24]] + b1)            (FCBlock layer 1, 72 -> 72)
//              w = W2 h + b2                                                       (FCBlock layer 2, 720..1944 wide)
//              msg = FullyConnectedTensorProduct(x_dst, Y_{0,1,2}(edge_vec), w)    (e3nn 'uvw' paths, lmax = 2)
//   per node:  mean -> BatchNorm -> residual in fctp_finalize_kernel
// replacing reference models/tensor_layers.py:195-217 with `faster=False` (models/all_atom_score_model.py:170-187).
//
// Same mapping onto the matrix cores as tp_conv.hip (exact fp32 v_mfma_f32_32x32x2_f32, one wave = 32 edges, weights
// streamed L2 -> registers one tile ahead, the ReLU'd accumulator of the first Linear re-used in place as the B operand of
// the second), with K = 72: 36 k-steps, of which the last 4 come from the 8 live rows of the third hidden tile.
// Block structure (conf_common.h): scalar blocks (24 outputs) use tiles of 4 mid indices x 8 outputs, three tiles per
// group of 4 mids; vector blocks (6 outputs) use tiles of 5 mid indices x 6 outputs.  The Clebsch-Gordan contraction
// runs on the VALU with canonical intermediates x, x.n, x n, x cross n and (n n^T - I/3) x; the e3nn path weights,
// sqrt(2l+1) spherical-harmonic scales and Wigner-3j constants are folded into the packed weights.
#include <cstddef>
#include <type_traits>

#include "conf_common.h"
#include "reduce_runs.h"

namespace cbd {

__device__ __forceinline__ f32x16 cmfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// `next` is the wave-uniform base of the next tile (an SGPR pair; one pinned base per four 1-KB fragments because the immediate field
// holds < 4 KB), the lane adds its constant offset: no vector address arithmetic per tile (see tp_conv_dev.h)
// The tile's 32 bias floats (accumulator layout: this lane half's float4s 2q + hf of the stream's bias row) sit in `bq`, requested from
// global memory one tile ahead -- the previous tile's MFMA chain covers the latency -- and are replaced by the next tile's here (as in
// tp_train_bwd_kernel: no bias table in LDS, 11 KB instead of 19.8 KB per wave).
__device__ __forceinline__ void cgemm_tile(f32x4 (&a)[CKSTEPS / 4], GPtr<f32x4> next, int lane, f32x4 (&bq)[4], GPtr<f32x4> next_bias,
                                           const float (&B)[CKSTEPS], f32x16& acc, int hf) {
  GPtr<f32x4> p0 = next, p1 = next + 4 * 64, p2 = next + 8 * 64, pb = next_bias;
  pin_s(p0); pin_s(p1); pin_s(p2); pin_s(pb);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    acc[4 * q + 0] = bq[q].x; acc[4 * q + 1] = bq[q].y; acc[4 * q + 2] = bq[q].z; acc[4 * q + 3] = bq[q].w;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = pb[hf + 2 * q];
#pragma unroll
  for (int sg = 0; sg < CKSTEPS / 4; ++sg) {
    const f32x4 w = a[sg];
    acc = cmfma32(w.x, B[4 * sg + 0], acc);
    acc = cmfma32(w.y, B[4 * sg + 1], acc);
    acc = cmfma32(w.z, B[4 * sg + 2], acc);
    acc = cmfma32(w.w, B[4 * sg + 3], acc);
    a[sg] = (sg < 4 ? p0 : sg < 8 ? p1 : p2)[lane + (sg & 3) * 64];
    __builtin_amdgcn_sched_barrier(0);   // keep each refill right behind its last use (see tp_conv.hip)
  }
}

// ---- canonical CG intermediates of edge j (xc = &xT[0][j], column stride 32; n = unit edge vector)
__device__ __forceinline__ void ld3(const float* p, float (&x)[3]) { x[0] = p[0]; x[1] = p[32]; x[2] = p[64]; }
__device__ __forceinline__ void cross_n(const float (&x)[3], const float (&n)[3], float (&m)[3]) {
  m[0] = x[1] * n[2] - x[2] * n[1];
  m[1] = x[2] * n[0] - x[0] * n[2];
  m[2] = x[0] * n[1] - x[1] * n[0];
}
__device__ __forceinline__ void quad_n(const float (&x)[3], const float (&n)[3], float (&m)[3]) {   // (n n^T - I/3) x
  const float d = x[0] * n[0] + x[1] * n[1] + x[2] * n[2];
  m[0] = d * n[0] - x[0] * (1.0f / 3.0f);
  m[1] = d * n[1] - x[1] * (1.0f / 3.0f);
  m[2] = d * n[2] - x[2] * (1.0f / 3.0f);
}

template <int IN>
__device__ __forceinline__ float cmid0e(const float* xc, int i, const float (&n)[3]) {
  if (i < CNS) return xc[i * 32];
  if (IN >= 1 && i < CNS + CNV) {
    const float* p = xc + (CC_1O + 3 * (i - CNS)) * 32;
    return p[0] * n[0] + p[32] * n[1] + p[64] * n[2];
  }
  return 0.f;
}

template <int IN>
__device__ __forceinline__ void cmid1o(const float* xc, int i, const float (&n)[3], float (&m)[3]) {
  constexpr FctpShape S = fctp_shape(IN, 3);
  float x[3];
  if (i < CNS) {
    const float s = xc[i * 32];
    m[0] = s * n[0]; m[1] = s * n[1]; m[2] = s * n[2];
  } else if (i < CNS + S.n1o) {
    ld3(xc + (CC_1O + 3 * (i - CNS)) * 32, m);
  } else if (i < CNS + 2 * S.n1o) {
    ld3(xc + (CC_1O + 3 * (i - CNS - S.n1o)) * 32, x);
    quad_n(x, n, m);
  } else if (i < S.fan1o) {
    ld3(xc + (CC_1E + 3 * (i - CNS - 2 * S.n1o)) * 32, x);
    cross_n(x, n, m);
  } else {
    m[0] = m[1] = m[2] = 0.f;
  }
}

template <int IN>
__device__ __forceinline__ void cmid1e(const float* xc, int i, const float (&n)[3], float (&m)[3]) {
  constexpr FctpShape S = fctp_shape(IN, 3);
  float x[3];
  if (i < S.n1o) {
    ld3(xc + (CC_1O + 3 * i) * 32, x);
    cross_n(x, n, m);
  } else if (i < S.n1o + S.n1e) {
    ld3(xc + (CC_1E + 3 * (i - S.n1o)) * 32, m);
  } else if (i < S.n1o + 2 * S.n1e) {
    ld3(xc + (CC_1E + 3 * (i - S.n1o - S.n1e)) * 32, x);
    quad_n(x, n, m);
  } else if (i < S.fan1e) {
    const float s = xc[(CC_0O + (i - S.n1o - 2 * S.n1e)) * 32];
    m[0] = s * n[0]; m[1] = s * n[1]; m[2] = s * n[2];
  } else {
    m[0] = m[1] = m[2] = 0.f;
  }
}

template <int IN>
__device__ __forceinline__ float cmid0o(const float* xc, int i, const float (&n)[3]) {
  constexpr FctpShape S = fctp_shape(IN, 3);
  if (i < S.n1e) {
    const float* p = xc + (CC_1E + 3 * i) * 32;
    return p[0] * n[0] + p[32] * n[1] + p[64] * n[2];
  }
  if (i < S.fan0o) return xc[(CC_0O + (i - S.n1e)) * 32];
  return 0.f;
}

constexpr int C_OUT_STRIDE = 34;      // two edges per ds_read_b64 in reduce_runs (reduce_runs.h)
constexpr int CXT_FLOATS = CN_STRIDE * C_OUT_STRIDE;   // gathered rows [84][32], later the message tile [84][34]
__host__ __device__ constexpr int fctp_lds_floats(int ntiles) { return CXT_FLOATS + 32; }

template <int IN, int OUT>
__global__ __launch_bounds__(64, 2) void fctp_conv_kernel(CArgs args) {
  constexpr FctpShape S = fctp_shape(IN, OUT);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;
  int* srcl = reinterpret_cast<int*>(xT + CXT_FLOATS);
  const int lane = threadIdx.x;
  const int j = lane & 31, hf = lane >> 5;

  // which group / tile?  The edge counts live on the device: lane g reads group g's count (one vector load for all groups; a scalar
  // loop costs two dependent scalar loads per group), a wave prefix sum turns tile counts into ranges, a ballot finds the owner.
  int grp = -1, e0 = 0, cnt = 0, tile_local = 0;
  {
    static_assert(CONF_LAUNCH_GROUPS <= 64, "one lane per group");
    int c = 0;
    if (lane < args.n_groups) {
      const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
      const int* cp = *reinterpret_cast<const int* const*>(ka + offsetof(CArgs, g) + (size_t)lane * sizeof(CGroup) + offsetof(CGroup, count));
      c = *cp;
    }
    const int nt = (c + WAVE_EDGES - 1) / WAVE_EDGES;
    int incl = nt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if (lane >= d) incl += v;
    }
    const int t = blockIdx.x;
    const unsigned long long owner = __ballot(t >= incl - nt && t < incl);
    if (owner == 0) return;
    grp = __builtin_ctzll(owner);
    cnt = __builtin_amdgcn_readlane(c, grp);
    tile_local = t - (__builtin_amdgcn_readlane(incl, grp) - __builtin_amdgcn_readlane(nt, grp));
    e0 = tile_local * WAVE_EDGES;
  }
  const CGroup G = args.g[grp];

  const GPtr<f32x4> gp = (GPtr<f32x4>)reinterpret_cast<const f32x4*>(G.wstream);   // uniform; tile T fragment sg of this lane: gp[T*576 + sg*64 + lane]
  f32x4 a[CKSTEPS / 4];
#pragma unroll
  for (int sg = 0; sg < CKSTEPS / 4; ++sg) a[sg] = gp[sg * 64 + lane];
  const GPtr<f32x4> gbias = (GPtr<f32x4>)reinterpret_cast<const f32x4*>(G.wstream + (size_t)(S.ntiles + 1) * CTILE_W_FLOATS);   // [ntiles + 1][32]
  f32x4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = gbias[hf + 2 * q];

  const int e = e0 + j;
  const bool valid = e < cnt;
  const int ec = valid ? e : cnt - 1;
  const int src_r = G.src[ec], dst = G.dst[ec], aidx = G.attr_idx[ec];
  const int src = valid ? src_r : -1;
  const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[aidx];
  const float n[3] = {vv.x, vv.y, vv.z};
  if (hf == 0) srcl[j] = src;

  float Bx[CKSTEPS];   // [edge_attr(24) | x_src[:24] | x_dst[:24]]; lane half hf holds columns 12hf .. 12hf+11 of each part
  {
    const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx * CNS + 12 * hf);
    const f32x4* ps = reinterpret_cast<const f32x4*>(G.node_in + (size_t)src_r * CN_STRIDE + 12 * hf);
    const f32x4* pd = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dst * CN_STRIDE + 12 * hf);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const f32x4 aa = pa[q], s = ps[q], d = pd[q];
      Bx[4 * q + 0] = aa.x; Bx[4 * q + 1] = aa.y; Bx[4 * q + 2] = aa.z; Bx[4 * q + 3] = aa.w;
      Bx[12 + 4 * q + 0] = s.x; Bx[12 + 4 * q + 1] = s.y; Bx[12 + 4 * q + 2] = s.z; Bx[12 + 4 * q + 3] = s.w;
      Bx[24 + 4 * q + 0] = d.x; Bx[24 + 4 * q + 1] = d.y; Bx[24 + 4 * q + 2] = d.z; Bx[24 + 4 * q + 3] = d.w;
    }
    // full destination row (21 float4) -> transposed LDS copy; half 0 copies float4 0..10, half 1 copies 10..20
    const f32x4* pr = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dst * CN_STRIDE) + 10 * hf;
#pragma unroll
    for (int q = 0; q < 11; ++q) {
      const f32x4 r = pr[q];
      float* o = xT + (40 * hf + 4 * q) * 32 + j;
      o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
    }
  }
  __syncthreads();

  int T = 0;
  f32x16 acc;
  float h1[CKSTEPS];
#define CBD_CTILE(BOP)                                                                             \
  cgemm_tile(a, gp + (size_t)(T + 1) * (CTILE_W_FLOATS / 4), lane, bq, gbias + (size_t)(T + 1 < S.ntiles ? T + 1 : T) * 8, BOP, acc, hf); \
  ++T

  // ---- first Linear: 72 hidden units = two full tiles + 8 live rows (registers 0..3 of both halves) of a third
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    CBD_CTILE(Bx);
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[16 * m + r] = relu1(acc[r]);
  }
  CBD_CTILE(Bx);
#pragma unroll
  for (int r = 0; r < 4; ++r) h1[32 + r] = relu1(acc[r]);

  const float* xc = xT + j;
  // ---- scalar blocks: group of 4 mids, three tiles (8 outputs each); register 4i+c of lane half hf = (mid 4g+i, output 8q+c+4hf)
  float o0e[12], o0o[12];
#pragma unroll
  for (int r = 0; r < 12; ++r) { o0e[r] = 0.f; o0o[r] = 0.f; }
  // One group = 4 mids x 3 tiles.  Groups whose four mids are plain scalar features run in a rolled, branch-free loop; the few
  // groups with dot-product mids or padding are unrolled so that the kind of each mid is a compile-time fact (rolled, every mid was a
  // chain of scalar branches around LDS reads that were waited for one by one -- see tp_conv.hip).
  auto scalar_group = [&](const float (&m)[4], float (&out)[12]) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      CBD_CTILE(h1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[4 * q + c] = fmaf(m[i], acc[4 * i + c], out[4 * q + c]);
    }
  };
  // tail group of one or two mids in two denser tiles (conf_common.h::sc_tail_dense): tile A carries output octets 0 and 1 of both
  // mids (slot i = (mid i & 1, octet i >> 1)), tile B octet 2 (slots 0, 1)
  // With merged tails (FctpShape::merged) block 0e runs tile A only and keeps its two tail mids; block 0o's tile B then carries block
  // 0e's octet 2 in slots 2, 3.
  float m0e_tail[2] = {0.f, 0.f};
  auto scalar_tail = [&](const float (&m)[4], float (&out)[12], auto is_0e_c) __attribute__((always_inline)) {
    constexpr bool is_0e = decltype(is_0e_c)::value;
    CBD_CTILE(h1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c) out[4 * (i >> 1) + c] = fmaf(m[i & 1], acc[4 * i + c], out[4 * (i >> 1) + c]);
    if constexpr (S.merged && is_0e) {
      m0e_tail[0] = m[0]; m0e_tail[1] = m[1];
    } else {
      CBD_CTILE(h1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 4; ++c) out[8 + c] = fmaf(m[i], acc[4 * i + c], out[8 + c]);
      if constexpr (S.merged && !is_0e) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) o0e[8 + c] = fmaf(m0e_tail[i], acc[4 * (2 + i) + c], o0e[8 + c]);
      }
    }
  };
  constexpr int G0E_PLAIN = CNS / 4;   // groups 0 .. G0E_PLAIN-1 of block 0e read x0e[4g .. 4g+3]
#pragma unroll 1
  for (int g = 0; g < G0E_PLAIN; ++g) {
    const float m[4] = {xc[(4 * g + 0) * 32], xc[(4 * g + 1) * 32], xc[(4 * g + 2) * 32], xc[(4 * g + 3) * 32]};
    scalar_group(m, o0e);
  }
#pragma unroll
  for (int g = G0E_PLAIN; g < S.g0e; ++g) {
    float m[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = cmid0e<IN>(xc, 4 * g + i, n);
    if (sc_tail_dense(S.fan0e, g)) scalar_tail(m, o0e, std::true_type{});
    else scalar_group(m, o0e);
#pragma unroll
    for (int r = 0; r < 12; ++r) pin(o0e[r]);   // ties the FMAs of an unrolled group to its place (tp_conv.hip)
  }

  // ---- vector blocks: tile = 5 mids x 6 outputs; register reg < 15 of lane half hf = (mid 5t + reg/3, output 3hf + reg%3).
  //      Fully unrolled (compile-time mids, a tile's LDS reads issued together before the chain); mids of the form scalar x direction
  //      cost one FMA per output, the direction is applied once per block; padded slots are skipped (as in tp_conv.hip).
  float k1o[9], k1e[9];
#pragma unroll
  for (int r = 0; r < 9; ++r) { k1o[r] = 0.f; k1e[r] = 0.f; }
  // `guest`: with merged vector tails (FctpShape::vmerged) block 1o runs one tile less and its `guest_n` tail mids (never of the
  // scalar x direction kind) occupy slots guest_slot .. of block 1e's last tile, accumulating into block 1o's sums (`gkeep`).
  auto vec_block = [&](auto mid_fn, auto is_scalar, auto scalar_of, auto ntile_c, auto fan_c, float (&keep)[9], auto guest_n_c, auto guest_base_c,
                       auto guest_slot_c, auto guest_fn, float (&gkeep)[9]) __attribute__((always_inline)) {
    constexpr int ntile = decltype(ntile_c)::value, fan = decltype(fan_c)::value;
    constexpr int guest_n = decltype(guest_n_c)::value, guest_base = decltype(guest_base_c)::value, guest_slot = decltype(guest_slot_c)::value;
    static_assert(guest_n == 0 || guest_base >= CNS, "guest mids must not be of the scalar x direction kind");
    float sc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < ntile; ++t) {
      float m[C_VEC_TILE_I][3], xs[C_VEC_TILE_I];
      float gm[guest_n > 0 ? guest_n : 1][3];
#pragma unroll
      for (int q = 0; q < C_VEC_TILE_I; ++q) {
        const int i = C_VEC_TILE_I * t + q;
        if (i >= fan) continue;
        if (is_scalar(i)) xs[q] = scalar_of(xc, i);
        else mid_fn(xc, i, n, m[q]);
      }
      if (t == ntile - 1) {
#pragma unroll
        for (int g = 0; g < guest_n; ++g) guest_fn(xc, guest_base + g, n, gm[g]);
      }
      CBD_CTILE(h1);
#pragma unroll
      for (int q = 0; q < C_VEC_TILE_I; ++q) {
        const int i = C_VEC_TILE_I * t + q;
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
      if (t == ntile - 1) {
#pragma unroll
        for (int g = 0; g < guest_n; ++g)
#pragma unroll
          for (int o = 0; o < 3; ++o) {
            const float w = acc[3 * (guest_slot + g) + o];
            gkeep[3 * o + 0] = fmaf(gm[g][0], w, gkeep[3 * o + 0]);
            gkeep[3 * o + 1] = fmaf(gm[g][1], w, gkeep[3 * o + 1]);
            gkeep[3 * o + 2] = fmaf(gm[g][2], w, gkeep[3 * o + 2]);
          }
      }
#pragma unroll
      for (int o = 0; o < 3; ++o) pin(sc[o]);
#pragma unroll
      for (int r = 0; r < 9; ++r) pin(keep[r]);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int c = 0; c < 3; ++c) keep[3 * o + c] = fmaf(n[c], sc[o], keep[3 * o + c]);
  };
  auto fn1o = [](const float* x, int i, const float (&nn)[3], float (&m)[3]) __attribute__((always_inline)) { cmid1o<IN>(x, i, nn, m); };
  constexpr int R1O = S.fan1o % C_VEC_TILE_I, R1E = S.fan1e % C_VEC_TILE_I;
  using I0 = std::integral_constant<int, 0>;
  vec_block(fn1o, [](int i) { return i < CNS; }, [](const float* x, int i) { return x[i * 32]; },
            std::integral_constant<int, S.t1o - S.vmerged>{}, std::integral_constant<int, S.vmerged ? C_VEC_TILE_I * (S.t1o - 1) : S.fan1o>{}, k1o,
            I0{}, I0{}, I0{}, fn1o, k1o);
  if constexpr (OUT >= 2)
    vec_block([](const float* x, int i, const float (&nn)[3], float (&m)[3]) __attribute__((always_inline)) { cmid1e<IN>(x, i, nn, m); },
              [](int i) { return i >= S.n1o + 2 * S.n1e; }, [](const float* x, int i) { return x[(CC_0O + (i - S.n1o - 2 * S.n1e)) * 32]; },
              std::integral_constant<int, S.t1e>{}, std::integral_constant<int, S.fan1e>{}, k1e,
              std::integral_constant<int, S.vmerged ? R1O : 0>{}, std::integral_constant<int, C_VEC_TILE_I * (S.t1o - 1)>{},
              std::integral_constant<int, R1E>{}, fn1o, k1o);
  if constexpr (OUT >= 3) {
    // block 0o: mids 0 .. n1e-1 are 1e . direction dot products, the rest plain 0o features: unrolled head (and padded tail), rolled middle
    constexpr int G_HEAD = (S.n1e + 3) / 4, G_FULL = S.fan0o / 4;
#pragma unroll
    for (int g = 0; g < (G_HEAD < S.g0o ? G_HEAD : S.g0o); ++g) {
      float m[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = cmid0o<IN>(xc, 4 * g + i, n);
      if (sc_tail_dense(S.fan0o, g)) scalar_tail(m, o0o, std::false_type{});
      else scalar_group(m, o0o);
#pragma unroll
      for (int r = 0; r < 12; ++r) pin(o0o[r]);
    }
#pragma unroll 1
    for (int g = G_HEAD; g < G_FULL; ++g) {
      const float* p = xc + (CC_0O + 4 * g - S.n1e) * 32;
      const float m[4] = {p[0], p[32], p[64], p[96]};
      scalar_group(m, o0o);
    }
#pragma unroll
    for (int g = (G_FULL > G_HEAD ? G_FULL : G_HEAD); g < S.g0o; ++g) {
      float m[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) m[i] = cmid0o<IN>(xc, 4 * g + i, n);
      if (sc_tail_dense(S.fan0o, g)) scalar_tail(m, o0o, std::false_type{});
      else scalar_group(m, o0o);
#pragma unroll
      for (int r = 0; r < 12; ++r) pin(o0o[r]);
    }
  }
#undef CBD_CTILE

  // ---- messages -> LDS, run-length sums per aggregating node (deterministic, see tp_conv.hip)
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      xT[(8 * q + c + 4 * hf) * C_OUT_STRIDE + j] = o0e[4 * q + c];
      if constexpr (OUT >= 3) xT[(CC_0O + 8 * q + c + 4 * hf) * C_OUT_STRIDE + j] = o0o[4 * q + c];
    }
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xT[(CC_1O + 3 * (3 * hf + o) + c) * C_OUT_STRIDE + j] = k1o[3 * o + c];
      if constexpr (OUT >= 2) xT[(CC_1E + 3 * (3 * hf + o) + c) * C_OUT_STRIDE + j] = k1e[3 * o + c];
    }
  __syncthreads();
  reduce_runs<CN_STRIDE, C_OUT_STRIDE, S.out_dim, 0>(xT, srcl, lane, S.out_dim, G.first_sum + (size_t)tile_local * CN_STRIDE,
                                       G.last_sum + (size_t)tile_local * CN_STRIDE, G.run_acc);
}

// mean over all incoming edge types -> e3nn BatchNorm (eval) -> residual (reference tensor_layers.py:206-216)
__global__ void fctp_finalize_kernel(CFinArgs fa, const float* __restrict__ node_in, float* __restrict__ node_out,
                                     const float* __restrict__ bn_scale, const float* __restrict__ bn_mean,
                                     const float* __restrict__ bn_bias, int n_nodes, int in_dim, int out_dim, int node_off) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = idx / CN_STRIDE, c = idx % CN_STRIDE;
  if (i >= n_nodes) return;
  const size_t o = (size_t)(i + node_off) * CN_STRIDE + c;
  float r = 0.f;
  if (c < out_dim) {
    float sum = 0.f;
    int deg = 0;
    for (int g = 0; g < fa.n_groups; ++g) {
      const CFinGroup& G = fa.g[g];
      const int s = G.start[i], n = G.cnt[i];
      deg += n;
      if (n <= 0) continue;
      const int e = s + n, t0 = s / WAVE_EDGES, t1 = (e - 1) / WAVE_EDGES;
      const bool at_start = (s % WAVE_EDGES) == 0;
      if (t0 == t1) {
        if (at_start) sum += G.first_sum[(size_t)t0 * CN_STRIDE + c];
        else if ((e % WAVE_EDGES) == 0) sum += G.last_sum[(size_t)t0 * CN_STRIDE + c];
        else sum += G.run_acc[(size_t)(i + node_off) * CN_STRIDE + c];
      } else {
        sum += (at_start ? G.first_sum : G.last_sum)[(size_t)t0 * CN_STRIDE + c];
        for (int t = t0 + 1; t <= t1; ++t) sum += G.first_sum[(size_t)t * CN_STRIDE + c];
      }
    }
    float m = sum / (float)(deg > 1 ? deg : 1);
    m = (m - bn_mean[c]) * bn_scale[c] + bn_bias[c];
    r = m + (c < in_dim ? node_in[o] : 0.f);
  }
  node_out[o] = r;
}

template <int IN, int OUT>
static hipError_t launch_fctp_one(const CArgs& a, int grid, hipStream_t s) {
  constexpr int lds_bytes = fctp_lds_floats(fctp_shape(IN, OUT).ntiles) * 4;
  hipLaunchKernelGGL((fctp_conv_kernel<IN, OUT>), dim3(grid), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}

hipError_t launch_fctp_conv(int in_level, int out_level, const CArgs& a, int grid, hipStream_t s) {
  if (grid <= 0) return hipSuccess;
  if (in_level == 0 && out_level == 1) return launch_fctp_one<0, 1>(a, grid, s);
  if (in_level == 1 && out_level == 2) return launch_fctp_one<1, 2>(a, grid, s);
  if (in_level == 2 && out_level == 3) return launch_fctp_one<2, 3>(a, grid, s);
  if (in_level == 3 && out_level == 3) return launch_fctp_one<3, 3>(a, grid, s);
  return hipErrorInvalidValue;
}

hipError_t launch_fctp_finalize(const CFinArgs& fa, const float* node_in, float* node_out, const float* bn_scale,
                                const float* bn_mean, const float* bn_bias, int n_nodes, int in_dim, int out_dim,
                                int node_off, hipStream_t s) {
  if (n_nodes <= 0) return hipSuccess;
  const int total = n_nodes * CN_STRIDE;
  hipLaunchKernelGGL(fctp_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, fa, node_in, node_out, bn_scale,
                     bn_mean, bn_bias, n_nodes, in_dim, out_dim, node_off);
  return hipGetLastError();
}

}  // namespace cbd
