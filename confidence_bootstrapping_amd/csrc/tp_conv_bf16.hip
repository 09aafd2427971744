// bf16-operand variant of the fused tensor-product message passing (tp_conv.hip) for gfx950 (MI355X):
// identical wave/tile/epilogue/reduction structure, but both Linears of the FCBlock run on v_mfma_f32_32x32x16_bf16
// (bf16 A/B operands, fp32 accumulate): weights are packed as bf16 (6 KB per 32-row tile, half the L2 stream), the gathered
// first-Linear inputs and the ReLU'd hidden activations are converted to bf16 in registers (v_cvt_pk_bf16_f32).  Node
// features, spherical-harmonic contraction, messages and the segmented reduction stay fp32.  Selected by
// cbd_set_option("bf16", 1) (BASELINE.json configs[3]: large-pocket stress case in bf16); tolerance vs the fp32 path is
// stated in tests/test_gpu_bf16.py.
//
// Operand maps (cdna_hip_programming.md section 3): lane (r = lane&31, h = lane>>5) holds A[row r][k = 8h + j] and
// B[k = 8h + j][col r], j = 0..7, for each of the 6 k-steps of 16 of a K = 96 tile.  The accumulator of the first Linear is
// re-used in place as the B operand of the second: registers 8s..8s+7 of hidden tile m are the fragment of k-step 2m+s, i.e.
// element j of lane half h is hidden unit 32m + 16s + 8(j>>2) + 4h + (j&3); W2's k order is permuted to match at pack time.
#include <cstdlib>

#include "common.h"

namespace cbd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int BQ = KDIM / 16;                 // 6 MFMA k-steps of 16 per tile
constexpr int BTILE_FRAGS = BQ * 64;          // bf16x8 fragments (16 B) per tile: 6 KB

__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// One 32x32 tile: acc = bias + A_tile * B with 6 bf16 MFMAs; the A fragments of the current tile are in registers and
// each is refilled with the next tile's data right after its use (same streaming scheme as the fp32 kernel).
__device__ __forceinline__ void gemm_tile_b(bf16x8 (&a)[BQ], const bf16x8* __restrict__ next, const float* __restrict__ bias_l,
                                            const bf16x8 (&B)[BQ], f32x16& acc, int hf) {
  const f32x4* bp = reinterpret_cast<const f32x4*>(bias_l);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 b = bp[2 * q + hf];
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
#pragma unroll
  for (int q = 0; q < BQ; ++q) {
    acc = mfma_bf16(a[q], B[q], acc);
    a[q] = next[q * 64];
    __builtin_amdgcn_sched_barrier(0);
  }
}

// "mid" evaluators: value of the CG intermediate with index i for edge j (xc = &xT[0][j], column stride 32).
// Index spaces follow reference tensor_layers.py:72-85 (concatenation order of out_dict[...] lists).
template <int IN>
__device__ __forceinline__ float mid0e(const float* xc, int i, const float (&v)[3]) {
  if (i < NS) return xc[i * 32];
  if (IN >= 1) {
    const float* p = xc + (COL_1O + 3 * (i - NS)) * 32;
    return p[0] * v[0] + p[32] * v[1] + p[64] * v[2];
  }
  return 0.f;
}

__device__ __forceinline__ void cross3(const float* p, const float (&v)[3], float (&m)[3]) {
  const float a0 = p[0], a1 = p[32], a2 = p[64];
  m[0] = a1 * v[2] - a2 * v[1];
  m[1] = a2 * v[0] - a0 * v[2];
  m[2] = a0 * v[1] - a1 * v[0];
}

template <int IN>
__device__ __forceinline__ void mid1o(const float* xc, int i, const float (&v)[3], float (&m)[3]) {
  constexpr ConvShape S = conv_shape(IN, 3);
  if (i < NS) {
    const float s = xc[i * 32];
    m[0] = s * v[0]; m[1] = s * v[1]; m[2] = s * v[2];
  } else if (i < NS + S.n1o) {
    const float* p = xc + (COL_1O + 3 * (i - NS)) * 32;
    m[0] = p[0]; m[1] = p[32]; m[2] = p[64];
  } else if (i < S.fan1o) {
    cross3(xc + (COL_1E + 3 * (i - NS - S.n1o)) * 32, v, m);
  } else {
    m[0] = m[1] = m[2] = 0.f;
  }
}

template <int IN>
__device__ __forceinline__ void mid1e(const float* xc, int i, const float (&v)[3], float (&m)[3]) {
  constexpr ConvShape S = conv_shape(IN, 3);
  if (i < S.n1o) {
    cross3(xc + (COL_1O + 3 * i) * 32, v, m);
  } else if (i < S.n1o + S.n1e) {
    const float* p = xc + (COL_1E + 3 * (i - S.n1o)) * 32;
    m[0] = p[0]; m[1] = p[32]; m[2] = p[64];
  } else if (i < S.fan1e) {
    const float s = xc[(COL_0O + (i - S.n1o - S.n1e)) * 32];
    m[0] = s * v[0]; m[1] = s * v[1]; m[2] = s * v[2];
  } else {
    m[0] = m[1] = m[2] = 0.f;
  }
}

template <int IN>
__device__ __forceinline__ float mid0o(const float* xc, int i, const float (&v)[3]) {
  constexpr ConvShape S = conv_shape(IN, 3);
  if (i < S.n1e) {
    const float* p = xc + (COL_1E + 3 * i) * 32;
    return p[0] * v[0] + p[32] * v[1] + p[64] * v[2];
  }
  if (i < S.fan0o) return xc[(COL_0O + (i - S.n1e)) * 32];
  return 0.f;
}

constexpr int XT_FLOATS = NODE_STRIDE * 32;              // per-wave transposed copy of the gathered rows
constexpr int OUT_STRIDE = 33;                           // message tile stride (conflict-free column reads)
__host__ __device__ constexpr int conv_lds_floats(int ntiles) { return ntiles * 32 + XT_FLOATS + 32; }

template <int IN, int OUT>
__global__ __launch_bounds__(64, 2) void tp_conv_bf16_kernel(ConvArgs args) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bias_l = lds;                                 // [ntiles][32]
  float* xT = lds + S.ntiles * 32;                     // [80][32] gathered destination rows, transposed
  int* srcl = reinterpret_cast<int*>(xT + XT_FLOATS);  // [32]
  const int lane = threadIdx.x;
  const int j = lane & 31, hf = lane >> 5;

  // ---- which group / edge range does this wave own?  (edge counts live on the device)
  int grp = -1, e0 = 0, cnt = 0, tile_local = 0;
  {
    int t = blockIdx.x;
    for (int g = 0; g < args.n_groups; ++g) {
      const int c = *args.g[g].count;
      const int nt = (c + CONV_WG_EDGES - 1) / CONV_WG_EDGES;
      if (grp < 0) {
        if (t < nt) { grp = g; e0 = t * CONV_WG_EDGES; cnt = c; tile_local = t; }
        else t -= nt;
      }
    }
  }
  if (grp < 0) return;
  const ConvGroup G = args.g[grp];

  // ---- start the weight stream: tile 0 fragments + the bias table of the group
  const bf16x8* gp = reinterpret_cast<const bf16x8*>(G.wstream) + lane;   // tile T fragment q: gp[T*384 + q*64]
  bf16x8 a[BQ];
#pragma unroll
  for (int q = 0; q < BQ; ++q) a[q] = gp[q * 64];
  {  // bias table -> LDS: fixed number of unconditional, clamped loads (a counted loop compiles to a load/wait waterfall)
    const f32x4* gb = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(G.wstream) + (size_t)(S.ntiles + 1) * BTILE_FRAGS * 16);
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

  bf16x8 Bx[BQ];  // first-Linear input [edge_attr(32) | x_src[:32] | x_dst[:32]]: k-step 2*seg+sub = cols 16hf+8sub .. +7 of segment seg
  {
    const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx * 32 + 16 * hf);
    const f32x4* ps = reinterpret_cast<const f32x4*>(G.node_in + (size_t)src_r * NODE_STRIDE + 16 * hf);
    const f32x4* pd = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dst * NODE_STRIDE + 16 * hf);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 aa = pa[q], s = ps[q], d = pd[q];
      const int qq = q >> 1, o = 4 * (q & 1);
      Bx[0 + qq][o + 0] = (__bf16)aa.x; Bx[0 + qq][o + 1] = (__bf16)aa.y; Bx[0 + qq][o + 2] = (__bf16)aa.z; Bx[0 + qq][o + 3] = (__bf16)aa.w;
      Bx[2 + qq][o + 0] = (__bf16)s.x; Bx[2 + qq][o + 1] = (__bf16)s.y; Bx[2 + qq][o + 2] = (__bf16)s.z; Bx[2 + qq][o + 3] = (__bf16)s.w;
      Bx[4 + qq][o + 0] = (__bf16)d.x; Bx[4 + qq][o + 1] = (__bf16)d.y; Bx[4 + qq][o + 2] = (__bf16)d.z; Bx[4 + qq][o + 3] = (__bf16)d.w;
    }
    // full destination row -> transposed LDS copy xT[col][j]; lane half hf copies cols 40hf .. 40hf+39
    const f32x4* pr = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dst * NODE_STRIDE + 40 * hf);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const f32x4 r = pr[q];
      float* o = xT + (40 * hf + 4 * q) * 32 + j;
      o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
    }
  }
  __syncthreads();   // single-wave workgroup: orders the LDS writes above before the reads below

  int T = 0;
  f32x16 acc;
  bf16x8 h1[BQ];
  // the stream carries one zero tile after the last real one, so the prefetch of tile T+1 is always in bounds
#define CBD_TILE(BOP, NEXT)                                                           \
  {                                                                                   \
    const int tn_ = (NEXT);                                                           \
    gemm_tile_b(a, gp + (size_t)tn_ * BTILE_FRAGS, bias_l + T * 32, BOP, acc, hf);    \
    T = tn_;                                                                          \
  }

  // ---- first Linear (3 tiles): h1 = ReLU(W1 x + b1), kept in the C/D register layout
  // A group may be a VIRTUAL slice of an edge group (ConvGroup::i0e_lo/hi, vec_on): the same edges, but only the 0e tiles
  // [lo, hi) and/or the vector blocks -- several waves then share one 32-edge tile's weight-tile chain (short launches)
  const int i_lo = G.i0e_lo, i_hi = G.i0e_hi;
  const bool vec_on = G.vec_on != 0;
  const int T_vec = 3 + S.t0e;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    CBD_TILE(Bx, m < 2 ? T + 1 : (i_lo < i_hi ? 3 + i_lo : T_vec));
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[2 * m + (r >> 3)][r & 7] = (__bf16)fmaxf(acc[r], 0.f);
  }

  const float* xc = xT + j;
  // ---- block 0e: one tile per mid index, 32 output scalars
  float o0e[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) o0e[r] = 0.f;
#pragma unroll 1
  for (int i = i_lo; i < i_hi; ++i) {
    CBD_TILE(h1, i + 1 < i_hi ? T + 1 : (vec_on ? T_vec : S.ntiles));
    const float m = mid0e<IN>(xc, i, v);
#pragma unroll
    for (int r = 0; r < 16; ++r) o0e[r] = fmaf(m, acc[r], o0e[r]);
  }

  // ---- vector / pseudoscalar blocks: tile = 5 mid indices x 6 outputs; lane half hf owns outputs 3hf..3hf+2, register
  //      reg < 15 holds (i = 5t + reg/3, o = 3hf + reg%3) -- no cross-lane traffic, mids identical in both halves
  float k1o[9], k1e[9], k0o[3];
#pragma unroll
  for (int r = 0; r < 9; ++r) { k1o[r] = 0.f; k1e[r] = 0.f; }
  k0o[0] = k0o[1] = k0o[2] = 0.f;

  auto vec_block = [&](auto mid_fn, int ntile, float (&keep)[9]) __attribute__((always_inline)) {
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
      CBD_TILE(h1, T + 1);
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        float m[3];
        mid_fn(xc, VEC_TILE_I * t + q, v, m);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const float w = acc[3 * q + o];
          keep[3 * o + 0] = fmaf(m[0], w, keep[3 * o + 0]);
          keep[3 * o + 1] = fmaf(m[1], w, keep[3 * o + 1]);
          keep[3 * o + 2] = fmaf(m[2], w, keep[3 * o + 2]);
        }
      }
    }
  };

  if (vec_on) {
  vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, vv, m); }, S.t1o, k1o);
  if constexpr (OUT >= 2)
    vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, vv, m); }, S.t1e, k1e);
  if constexpr (OUT >= 3) {
#pragma unroll 1
    for (int t = 0; t < S.t0o; ++t) {
      CBD_TILE(h1, T + 1);
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const float m = mid0o<IN>(xc, VEC_TILE_I * t + q, v);
#pragma unroll
        for (int o = 0; o < 3; ++o) k0o[o] = fmaf(m, acc[3 * q + o], k0o[o]);
      }
    }
  }
  }

#undef CBD_TILE
  // ---- messages -> LDS (re-using the gathered-row tile, stride 33 so that the column reads below are conflict free),
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
  // Run-length sums without atomics (bitwise reproducible): a run that starts at the tile's first edge goes to
  // first_sum[tile], one that ends at edge 31 to last_sum[tile], any other run (strictly inside the tile) is the
  // node's only contribution from this group and is stored directly; conv_finalize_kernel adds the pieces in tile order.
  float* const fs = G.first_sum + (size_t)tile_local * NODE_STRIDE;
  float* const ls = G.last_sum + (size_t)tile_local * NODE_STRIDE;
  for (int col = lane; col < S.out_dim; col += 64) {
    const float* oc = xT + col * OUT_STRIDE;
    float sum = 0.f;
    int cur = srcl[0], a0 = 0;
    for (int jj = 0; jj < 32; ++jj) {
      const int sj = srcl[jj];
      if (sj != cur) {   // run [a0, jj-1] of node cur is complete (invalid lanes, src = -1, only follow valid ones)
        // (a run that ends at the last edge of the group's partial tile has no other tile either: stored as interior)
        float* dst = a0 == 0 ? fs : G.run_acc + (size_t)cur * NODE_STRIDE;
        dst[col] = sum;
        sum = 0.f;
        a0 = jj;
        cur = sj;
      }
      sum += oc[jj];
    }
    if (cur >= 0) (a0 == 0 ? fs : ls)[col] = sum;   // run that reaches edge 31 of a full tile
  }
}

// ---------------------------------------------------------------------------------------------- host launcher
template <int IN, int OUT>
static hipError_t launch_one_b(const ConvArgs& a, int grid, hipStream_t s) {
  constexpr int lds_bytes = conv_lds_floats(conv_shape(IN, OUT).ntiles) * 4;
  hipLaunchKernelGGL((tp_conv_bf16_kernel<IN, OUT>), dim3(grid), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}

hipError_t launch_tp_conv_bf16(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s) {
  if (grid <= 0) return hipSuccess;
  if (in_level == 0 && out_level == 1) return launch_one_b<0, 1>(a, grid, s);
  if (in_level == 1 && out_level == 2) return launch_one_b<1, 2>(a, grid, s);
  if (in_level == 2 && out_level == 3) return launch_one_b<2, 3>(a, grid, s);
  if (in_level == 3 && out_level == 3) return launch_one_b<3, 3>(a, grid, s);
  return hipErrorInvalidValue;
}

}  // namespace cbd
