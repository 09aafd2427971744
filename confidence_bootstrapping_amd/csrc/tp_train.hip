// Training-side tensor-product kernels for gfx950 (MI355X): forward and backward of
//   msg[e] = FasterTensorProduct(x[e], sh(vec[e]), w[e]),   w[e] = W2 h[e] + b2
// (reference models/tensor_layers.py:66-117 fed by the last Linear of the FCBlock, models/layers.py:8-15) for the
// confidence-bootstrapping fine-tuning step (reference finetune_train.py:330 -> utils/training.py:184-233), where autograd needs
// d msg / d{x, h, W2, b2}.  As in the inference kernel (tp_conv.hip) the [E, W] per-edge weight tensor never exists in the
// forward pass: one wave owns 32 edges, the hidden activations h are the B operand in registers, the packed W2 tiles stream from
// L2 into registers one tile ahead, and every finished 32x32 tile of w is consumed at once by the CG contraction on the VALU.
//
// What differs from inference:
//   * the first Linear + ReLU + Dropout of the FCBlock stay in PyTorch (rocBLAS GEMM + torch's own dropout RNG), so h[E, 96]
//     is an input and its gradient an output of the op -- dropout masks are then exactly torch's and the first Linear's
//     gradients are ordinary autograd;
//   * all tensors are per edge and dense (x rows already gathered, messages not yet reduced): gathers, the scatter-mean over
//     edge groups and BatchNorm in training mode are autograd-visible torch ops around the kernel;
//   * backward re-computes each w tile on the matrix cores (same 48 MFMAs as forward) instead of loading a saved [E, W]
//     tensor, forms   g_w[row] = mid . g_msg[out(row)]   and   g_mid = sum_out w[row] g_msg[out(row)]   on the VALU, folds
//     g_mid through the transposed CG "mid" maps into g_x (per-wave LDS tile), and writes g_w[E, Wp] (packed-row order, the
//     only [E, W]-sized tensor of the step: 6.6 KB/edge written once, read twice by the two rocBLAS GEMMs
//     g_h = g_w W2p and dW2p = g_w^T h that finish the FCBlock's backward).
//
// Weight stream: the inference layout (cbd_pack_conv_stream: 3 tiles of W1 that this kernel skips, then the W2 tiles
// regrouped per output irrep block with 1/sqrt(fan_in), sqrt(3), sqrt(1.5) folded, then the bias table); g_w columns are
// (tile - 3) * 32 + row of that stream.  The Python side maps them back to the reference parameter layout
// (confidence_bootstrapping_amd/train_ops.py).
#include <type_traits>

#include "host_util.h"
#include "kernels.h"
#include "tp_conv_dev.h"

namespace cbd {

constexpr int TRAIN_MAX_GROUPS = 4;
struct TrainTpArgs {
  const float* xrow;     // [E][NODE_STRIDE] features of the node every edge reads (node_attr[edge_dst]), zero padded
  const float* vec;      // [E][4] unit edge vector (xyz, 0)
  const float* h;        // [E][96] hidden activations of the radial MLP
  // Edge groups of one layer (reference edge_groups / fc[g], models/tensor_layers.py:190,201): group g owns edges
  // [e_begin[g], e_begin[g+1]) and its own FCBlock stream; one launch carries all groups, so that the small groups do not pay a
  // launch drain of their own.  Waves never straddle groups (the grid is the sum of the groups' tile counts).
  const float* wstream[TRAIN_MAX_GROUPS];
  int e_begin[TRAIN_MAX_GROUPS + 1];
  int n_groups;
  int E;
  float* msg;            // forward out  [E][NODE_STRIDE]
  const float* gmsg;     // backward in  [E][NODE_STRIDE]  d loss / d msg
  float* gx;             // backward out [E][NODE_STRIDE]  d loss / d xrow
  float* gw;             // backward out [E][Wp]           d loss / d (packed w)
  float* gh;             // g_h pass out [E][96]           d loss / d h   (tp_train_gh_kernel; wstream = TRANSPOSED streams)
};

constexpr int GX_STRIDE = 33;
// forward: bias table + transposed row tile.  Backward: NO bias table (the tile's bias rows come from global memory into registers, one tile
// ahead) and the g_x tile starts on the row tile's padding rows 76..79, which are written by the prologue and never read: 19.5 KB per wave
// = 8 waves per CU.  (With the bias table and an 80-column g_x tile it was 27.5 KB = 5 waves per CU against the 8 the registers allow.)
constexpr int XT_LIVE_ROWS = 76;
__host__ __device__ constexpr int train_lds_floats(int ntiles, bool bwd) {
  return bwd ? XT_LIVE_ROWS * 32 + NODE_DIM * GX_STRIDE : ntiles * 32 + XT_FLOATS;
}
static_assert(XT_LIVE_ROWS * 32 + NODE_DIM * GX_STRIDE >= XT_FLOATS, "the prologue writes all 80 rows of the row tile");

// which group / 32-edge tile does workgroup `block` own?
__device__ __forceinline__ void train_locate(const TrainTpArgs& A, int block, int& grp, int& e0, int& e_end) {
  int t = block;
  grp = 0;
#pragma unroll
  for (int g = 0; g < TRAIN_MAX_GROUPS; ++g) {
    if (g < A.n_groups) {
      const int nt = (A.e_begin[g + 1] - A.e_begin[g] + WAVE_EDGES - 1) / WAVE_EDGES;
      if (t >= 0 && t < nt) { grp = g; e0 = A.e_begin[g] + t * WAVE_EDGES; e_end = A.e_begin[g + 1]; t = -1; }
      else if (t >= 0) t -= nt;
    }
  }
}

// common prologue: weight stream start (tile 3), bias table, gathered row tile, hidden activations as the MFMA B operand
template <int IN, int OUT>
__device__ __forceinline__ void train_prologue(const TrainTpArgs& A, const float* wstream, float* bias_l, float* xT, int lane, int ec, GPtr<f32x4> gp,
                                               f32x4 (&a)[OpsF32::NFRAG], OpsF32::Act& h1) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  const int j = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int sg = 0; sg < OpsF32::NFRAG; ++sg) a[sg] = gp[(size_t)3 * OpsF32::TILE_FRAGS + sg * 64 + lane];
  if (bias_l) {
    const f32x4* gb = reinterpret_cast<const f32x4*>(wstream) + (size_t)(S.ntiles + 1) * OpsF32::TILE_FRAGS;
    constexpr int NB4 = S.ntiles * 8, NBI = (NB4 + 63) / 64;
    f32x4 bt[NBI];
#pragma unroll
    for (int i = 0; i < NBI; ++i) { const int k = lane + 64 * i; bt[i] = gb[k < NB4 ? k : NB4 - 1]; }
#pragma unroll
    for (int i = 0; i < NBI; ++i) { const int k = lane + 64 * i; reinterpret_cast<f32x4*>(bias_l)[k < NB4 ? k : NB4 - 1] = bt[i]; }
  }
  // k-step s = 16m + r of lane half hf is hidden unit 32m + (r&3) + 8(r>>2) + 4hf (the C/D layout the stream's W2 follows)
  const f32x4* ph = reinterpret_cast<const f32x4*>(A.h + (size_t)ec * KDIM + 4 * hf);
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 x = ph[8 * m + 2 * q];
      h1.v[16 * m + 4 * q + 0] = x.x; h1.v[16 * m + 4 * q + 1] = x.y; h1.v[16 * m + 4 * q + 2] = x.z; h1.v[16 * m + 4 * q + 3] = x.w;
    }
  const f32x4* pr = reinterpret_cast<const f32x4*>(A.xrow + (size_t)ec * NODE_STRIDE + 40 * hf);
#pragma unroll
  for (int q = 0; q < 10; ++q) {
    const f32x4 r = pr[q];
    float* o = xT + (40 * hf + 4 * q) * 32 + j;
    o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
  }
}

template <int IN, int OUT>
__global__ __launch_bounds__(64, 2) void tp_train_fwd_kernel(TrainTpArgs A) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* bias_l = lds;
  float* xT = lds + S.ntiles * 32;
  const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
  int grp = 0, e0 = 0, e_end = 0;
  train_locate(A, blockIdx.x, grp, e0, e_end);
  const int e = e0 + j;
  const int ec = e < e_end ? e : e_end - 1;
  const float* const wstream = A.wstream[grp];
  const GPtr<f32x4> gp = (GPtr<f32x4>)reinterpret_cast<const f32x4*>(wstream);   // wave-uniform stream base (tp_conv_dev.h: gemm_u)
  f32x4 a[OpsF32::NFRAG];
  OpsF32::Act h1;
  train_prologue<IN, OUT>(A, wstream, bias_l, xT, lane, ec, gp, a, h1);
  const f32x4 vv = reinterpret_cast<const f32x4*>(A.vec)[ec];
  const float v[3] = {vv.x, vv.y, vv.z};
  __syncthreads();

  int T = 3;
  f32x16 acc;
#define CBD_TT()                                                                                              \
  {                                                                                                           \
    gemm_tile_u<OpsF32>(a, gp + (size_t)(T + 1) * OpsF32::TILE_FRAGS, lane, bias_l + T * 32, h1, acc, hf);            \
    ++T;                                                                                                      \
  }
  const float* xc = xT + j;
  float o0e[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) o0e[r] = 0.f;
  // (the epilogue forms of the inference kernel, tp_conv.hip: the 0e mid is read from LDS BEFORE the tile's MFMA chain -- two loops, one per
  //  kind of mid, so that neither body branches on the mid index -- and the vector-block tile loops are fully unrolled: compile-time mid
  //  kinds, a tile's LDS reads issued together before the chain, scalar x direction mids factored out of the sums)
  auto tile0e = [&](float m) __attribute__((always_inline)) {
    CBD_TT();
#pragma unroll
    for (int r = 0; r < 16; ++r) o0e[r] = fmaf(m, acc[r], o0e[r]);
  };
#pragma unroll 1
  for (int i = 0; i < NS; ++i) tile0e(xc[i * 32]);
  if constexpr (IN >= 1) {
#pragma unroll 1
    for (int i = NS; i < S.t0e; ++i) {
      const float* p = xc + (COL_1O + 3 * (i - NS)) * 32;
      tile0e(p[0] * v[0] + p[32] * v[1] + p[64] * v[2]);
    }
  }
  float k1o[9], k1e[9], k0o[3];
#pragma unroll
  for (int r = 0; r < 9; ++r) { k1o[r] = 0.f; k1e[r] = 0.f; }
  k0o[0] = k0o[1] = k0o[2] = 0.f;
  auto vec_block = [&](auto mid_fn, auto is_scalar, auto scalar_of, auto ntile_c, auto fan_c, float (&keep)[9]) __attribute__((always_inline)) {
    constexpr int ntile = decltype(ntile_c)::value, fan = decltype(fan_c)::value;
    float sc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < ntile; ++t) {
      float m[VEC_TILE_I][3], xs[VEC_TILE_I];
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const int i = VEC_TILE_I * t + q;
        if (i >= fan) continue;
        if (is_scalar(i)) xs[q] = scalar_of(xc, i);
        else mid_fn(xc, i, v, m[q]);
      }
      CBD_TT();
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
#pragma unroll
      for (int o = 0; o < 3; ++o) pin(sc[o]);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int c = 0; c < 3; ++c) keep[3 * o + c] = fmaf(v[c], sc[o], keep[3 * o + c]);
  };
  vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, vv, m); },
            [](int i) { return i < NS; }, [](const float* x, int i) { return x[i * 32]; },
            std::integral_constant<int, S.t1o>{}, std::integral_constant<int, S.fan1o>{}, k1o);
  if constexpr (OUT >= 2)
    vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, vv, m); },
              [](int i) { return i >= S.n1o + S.n1e; }, [](const float* x, int i) { return x[(COL_0O + (i - S.n1o - S.n1e)) * 32]; },
              std::integral_constant<int, S.t1e>{}, std::integral_constant<int, S.fan1e>{}, k1e);
  if constexpr (OUT >= 3) {
#pragma unroll
    for (int t = 0; t < S.t0o; ++t) {
      float m[VEC_TILE_I];
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        if (VEC_TILE_I * t + q >= S.fan0o) continue;
        m[q] = mid0o<IN>(xc, VEC_TILE_I * t + q, v);
      }
      CBD_TT();
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        if (VEC_TILE_I * t + q >= S.fan0o) continue;
#pragma unroll
        for (int o = 0; o < 3; ++o) k0o[o] = fmaf(m[q], acc[3 * q + o], k0o[o]);
      }
#pragma unroll
      for (int o = 0; o < 3; ++o) pin(k0o[o]);
    }
  }
#undef CBD_TT
  // messages -> LDS tile -> coalesced rows of msg (columns >= out_dim are written as zeros)
  __syncthreads();
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
  const int nrow = e_end - e0 < WAVE_EDGES ? e_end - e0 : WAVE_EDGES;
  for (int col = lane; col < NODE_STRIDE; col += 64) {
    const bool live = col < S.out_dim;
    for (int jj = 0; jj < nrow; ++jj) A.msg[(size_t)(e0 + jj) * NODE_STRIDE + col] = live ? xT[col * OUT_STRIDE + jj] : 0.f;
  }
}

__device__ __forceinline__ float half_sum(float x) { return x + __shfl_xor(x, 32, 64); }

template <int IN, int OUT>
__global__ __launch_bounds__(64, 2) void tp_train_bwd_kernel(TrainTpArgs A) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  constexpr int WP = (S.ntiles - 3) * 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;
  float* gxT = xT + XT_LIVE_ROWS * 32;   // [NODE_DIM][GX_STRIDE] gradient wrt the gathered row, column-major per edge (see train_lds_floats)
  const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
  int grp = 0, e0 = 0, e_end = 0;
  train_locate(A, blockIdx.x, grp, e0, e_end);
  const int e = e0 + j;
  const bool valid = e < e_end;
  const int ec = valid ? e : e_end - 1;
  const float* const wstream = A.wstream[grp];
  const GPtr<f32x4> gp = (GPtr<f32x4>)reinterpret_cast<const f32x4*>(wstream);   // wave-uniform stream base (tp_conv_dev.h: gemm_u)
  f32x4 a[OpsF32::NFRAG];
  OpsF32::Act h1;
  train_prologue<IN, OUT>(A, wstream, nullptr, xT, lane, ec, gp, a, h1);
  const f32x4 vv = reinterpret_cast<const f32x4*>(A.vec)[ec];
  const float v[3] = {vv.x, vv.y, vv.z};
  // this lane's slice of d loss / d msg in the accumulator layout: 0e rows (r&3) + 8(r>>2) + 4hf; vector outputs 3hf..3hf+2
  const float* gm = A.gmsg + (size_t)ec * NODE_STRIDE;
  float g0e[16], g1o[9], g1e[9], g0o[3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 x = reinterpret_cast<const f32x4*>(gm + 4 * hf)[2 * q];
    g0e[4 * q + 0] = x.x; g0e[4 * q + 1] = x.y; g0e[4 * q + 2] = x.z; g0e[4 * q + 3] = x.w;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    g1o[k] = gm[COL_1O + 9 * hf + k];
    g1e[k] = OUT >= 2 ? gm[COL_1E + 9 * hf + k] : 0.f;
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) g0o[o] = OUT >= 3 ? gm[COL_0O + 3 * hf + o] : 0.f;
  for (int k = lane; k < NODE_DIM * GX_STRIDE; k += 64) gxT[k] = 0.f;     // (after the prologue's writes to the rows it overlaps)
  __syncthreads();

  int T = 3;
  f32x16 acc;
  // the tile's 32 bias floats in the accumulator layout: this lane half's float4s 2q + hf of row T of the stream's bias table, requested
  // one tile ahead with a wave-uniform base (the whole MFMA chain of the previous tile covers the latency)
  const GPtr<f32x4> gbias = (GPtr<f32x4>)(reinterpret_cast<const f32x4*>(wstream) + (size_t)(S.ntiles + 1) * OpsF32::TILE_FRAGS);
  f32x4 bq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) bq[q] = gbias[(size_t)T * 8 + hf + 2 * q];
#define CBD_TT()                                                                                              \
  {                                                                                                           \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                           \
      acc[4 * q + 0] = bq[q].x; acc[4 * q + 1] = bq[q].y; acc[4 * q + 2] = bq[q].z; acc[4 * q + 3] = bq[q].w; \
    }                                                                                                         \
    {                                                                                                         \
      GPtr<f32x4> pb = gbias + (size_t)(T + 1) * 8;                                                           \
      pin_s(pb);                                                                                              \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) bq[q] = pb[hf + 2 * q];                                   \
    }                                                                                                         \
    OpsF32::gemm_u(a, gp + (size_t)(T + 1) * OpsF32::TILE_FRAGS, lane, h1, acc);                              \
    ++T;                                                                                                      \
  }
  const float* xc = xT + j;
  float* gxc = gxT + j;                                   // gxc[col * GX_STRIDE]
  float* const gwrow = A.gw + (size_t)ec * WP + 4 * hf;   // + (tile - 3) * 32 + 8q: the 4 rows (r&3) of register quad q
  auto store_gw = [&](int tile, const float (&gw)[16]) __attribute__((always_inline)) {
    if (valid && A.gw) {         // gw == NULL: g_w is re-formed by the g_h / dW2p passes and never stored
      f32x4* o = reinterpret_cast<f32x4*>(gwrow + (size_t)(tile - 3) * 32);
#pragma unroll
      for (int q = 0; q < 4; ++q) o[2 * q] = f32x4{gw[4 * q], gw[4 * q + 1], gw[4 * q + 2], gw[4 * q + 3]};
    }
  };
  // transposed mid maps: add g (gradient wrt mid i of a block) into the gradient of the gathered row.  Only lane half 0
  // updates the LDS tile (both halves hold the same, already half-summed g).
  auto add_s = [&](int col, float g) __attribute__((always_inline)) { gxc[col * GX_STRIDE] += g; };
  auto add_v = [&](int col, float g0, float g1, float g2) __attribute__((always_inline)) {
    gxc[col * GX_STRIDE] += g0; gxc[(col + 1) * GX_STRIDE] += g1; gxc[(col + 2) * GX_STRIDE] += g2;
  };

  // ---- block 0e
#pragma unroll 1
  for (int i = 0; i < S.t0e; ++i) {
    const int Tc = T;
    CBD_TT();
    const float m = mid0e<IN>(xc, i, v);
    float gw[16], part = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { gw[r] = m * g0e[r]; part = fmaf(acc[r], g0e[r], part); }
    store_gw(Tc, gw);
    const float g = half_sum(part);
    if (hf == 0) {
      if (i < NS) add_s(i, g);
      else if (IN >= 1) add_v(COL_1O + 3 * (i - NS), g * v[0], g * v[1], g * v[2]);
    }
  }
  // ---- vector blocks
  auto vec_block = [&](auto mid_fn, auto back_fn, int ntile, const float (&gk)[9]) __attribute__((always_inline)) {
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
      const int Tc = T;
      CBD_TT();
      float gw[16];
      gw[15] = 0.f;
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        float m[3];
        mid_fn(xc, VEC_TILE_I * t + q, v, m);
        float gmid[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const float w = acc[3 * q + o];
          gw[3 * q + o] = m[0] * gk[3 * o + 0] + m[1] * gk[3 * o + 1] + m[2] * gk[3 * o + 2];
          gmid[0] = fmaf(w, gk[3 * o + 0], gmid[0]);
          gmid[1] = fmaf(w, gk[3 * o + 1], gmid[1]);
          gmid[2] = fmaf(w, gk[3 * o + 2], gmid[2]);
        }
        gmid[0] = half_sum(gmid[0]); gmid[1] = half_sum(gmid[1]); gmid[2] = half_sum(gmid[2]);
        if (hf == 0) back_fn(VEC_TILE_I * t + q, gmid);
      }
      store_gw(Tc, gw);
    }
  };
  // d/da of g . (a x v) is v x g
  auto vxg = [&](const float (&g)[3], float (&o)[3]) __attribute__((always_inline)) {
    o[0] = v[1] * g[2] - v[2] * g[1];
    o[1] = v[2] * g[0] - v[0] * g[2];
    o[2] = v[0] * g[1] - v[1] * g[0];
  };
  vec_block([](const float* x, int i, const float (&u)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, u, m); },
            [&](int i, const float (&g)[3]) __attribute__((always_inline)) {
              if (i < NS) add_s(i, g[0] * v[0] + g[1] * v[1] + g[2] * v[2]);
              else if (i < NS + S.n1o) add_v(COL_1O + 3 * (i - NS), g[0], g[1], g[2]);
              else if (i < S.fan1o) { float o[3]; vxg(g, o); add_v(COL_1E + 3 * (i - NS - S.n1o), o[0], o[1], o[2]); }
            },
            S.t1o, g1o);
  if constexpr (OUT >= 2)
    vec_block([](const float* x, int i, const float (&u)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, u, m); },
              [&](int i, const float (&g)[3]) __attribute__((always_inline)) {
                if (i < S.n1o) { float o[3]; vxg(g, o); add_v(COL_1O + 3 * i, o[0], o[1], o[2]); }
                else if (i < S.n1o + S.n1e) add_v(COL_1E + 3 * (i - S.n1o), g[0], g[1], g[2]);
                else if (i < S.fan1e) add_s(COL_0O + (i - S.n1o - S.n1e), g[0] * v[0] + g[1] * v[1] + g[2] * v[2]);
              },
              S.t1e, g1e);
  if constexpr (OUT >= 3) {
#pragma unroll 1
    for (int t = 0; t < S.t0o; ++t) {
      const int Tc = T;
      CBD_TT();
      float gw[16];
      gw[15] = 0.f;
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const int i = VEC_TILE_I * t + q;
        const float m = mid0o<IN>(xc, i, v);
        float part = 0.f;
#pragma unroll
        for (int o = 0; o < 3; ++o) { gw[3 * q + o] = m * g0o[o]; part = fmaf(acc[3 * q + o], g0o[o], part); }
        const float g = half_sum(part);
        if (hf == 0) {
          if (i < S.n1e) add_v(COL_1E + 3 * i, g * v[0], g * v[1], g * v[2]);
          else if (i < S.fan0o) add_s(COL_0O + (i - S.n1e), g);
        }
      }
      store_gw(Tc, gw);
    }
  }
#undef CBD_TT
  __syncthreads();
  const int nrow = e_end - e0 < WAVE_EDGES ? e_end - e0 : WAVE_EDGES;
  for (int col = lane; col < NODE_STRIDE; col += 64) {
    const bool live = col < NODE_DIM;
    for (int jj = 0; jj < nrow; ++jj) A.gx[(size_t)(e0 + jj) * NODE_STRIDE + col] = live ? gxT[col * GX_STRIDE + jj] : 0.f;
  }
}

template <int IN, int OUT>
static hipError_t launch_train(bool bwd, const TrainTpArgs& a, hipStream_t s) {
  int grid = 0;
  for (int g = 0; g < a.n_groups; ++g) grid += (a.e_begin[g + 1] - a.e_begin[g] + WAVE_EDGES - 1) / WAVE_EDGES;
  if (grid == 0) return hipSuccess;
  const int lds_bytes = train_lds_floats(conv_shape(IN, OUT).ntiles, bwd) * 4;
  if (bwd) hipLaunchKernelGGL((tp_train_bwd_kernel<IN, OUT>), dim3(grid), dim3(64), lds_bytes, s, a);
  else hipLaunchKernelGGL((tp_train_fwd_kernel<IN, OUT>), dim3(grid), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}

static hipError_t launch_train_any(int in_level, int out_level, bool bwd, const TrainTpArgs& a, hipStream_t s) {
  if (in_level == 0 && out_level == 1) return launch_train<0, 1>(bwd, a, s);
  if (in_level == 1 && out_level == 2) return launch_train<1, 2>(bwd, a, s);
  if (in_level == 2 && out_level == 3) return launch_train<2, 3>(bwd, a, s);
  if (in_level == 3 && out_level == 3) return launch_train<3, 3>(bwd, a, s);
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------------------
// g_h = g_w W2p on the matrix cores WITHOUT reading g_w from memory: the gradient of the hidden activations of the radial MLP,
// d loss / d h[e][k] = sum_w g_w[e][w] W2p[w][k]  (autograd of fc[3] in models/layers.py:8-15 under utils/training.py:205).
// One wave = 32 edges, like the other two training kernels.  Per weight tile it re-forms this wave's [32 rows x 32 edges] tile of
// g_w = g_msg (x) mid on the VALU -- exactly the registers tp_train_bwd_kernel stores -- and feeds it as the B operand of 3 x 16
// v_mfma_f32_32x32x2_f32 (the C/D layout of the 16 registers IS a valid k order: k-step s of lane half hf is tile row
// (s & 3) + 8 (s >> 2) + 4 hf), A = the TRANSPOSED weight tile streamed L2 -> registers one tile ahead:
//   fragment f = 16 kb + s of lane (i, hf') = W2p[32 T + (s & 3) + 8 (s >> 2) + 4 hf'][32 kb + i]      (train_ops.StreamHub packs it)
// accumulating the three 32-column blocks of g_h in 48 registers for the whole tile loop.
__device__ __forceinline__ void gemm_gh(f32x4 (&a)[OpsF32::NFRAG], GPtr<f32x4> next, int lane, const float (&gw)[16], f32x16& G0,
                                        f32x16& G1, f32x16& G2) {
  GPtr<f32x4> p0 = next, p1 = next + 4 * 64, p2 = next + 8 * 64;
  pin_s(p0); pin_s(p1); pin_s(p2);
#define CBD_GH_BLOCK(G, BASE, P)                                                             \
  _Pragma("unroll") for (int sg = 0; sg < 4; ++sg) {                                          \
    const f32x4 w = a[BASE + sg];                                                            \
    G = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, gw[4 * sg + 0], G, 0, 0, 0);               \
    G = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, gw[4 * sg + 1], G, 0, 0, 0);               \
    G = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, gw[4 * sg + 2], G, 0, 0, 0);               \
    G = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, gw[4 * sg + 3], G, 0, 0, 0);               \
    a[BASE + sg] = P[lane + sg * 64];                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
  }
  CBD_GH_BLOCK(G0, 0, p0)
  CBD_GH_BLOCK(G1, 4, p1)
  CBD_GH_BLOCK(G2, 8, p2)
#undef CBD_GH_BLOCK
}

template <int IN, int OUT>
__global__ __launch_bounds__(64, 2) void tp_train_gh_kernel(TrainTpArgs A) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xT = lds;
  const int lane = threadIdx.x, j = lane & 31, hf = lane >> 5;
  int grp = 0, e0 = 0, e_end = 0;
  train_locate(A, blockIdx.x, grp, e0, e_end);
  const int e = e0 + j;
  const bool valid = e < e_end;
  const int ec = valid ? e : e_end - 1;
  const GPtr<f32x4> gp = (GPtr<f32x4>)reinterpret_cast<const f32x4*>(A.wstream[grp]);   // transposed stream, tile 0 = second-Linear tile 3
  f32x4 a[OpsF32::NFRAG];
#pragma unroll
  for (int sg = 0; sg < OpsF32::NFRAG; ++sg) a[sg] = gp[sg * 64 + lane];
  {
    const f32x4* pr = reinterpret_cast<const f32x4*>(A.xrow + (size_t)ec * NODE_STRIDE + 40 * hf);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const f32x4 r = pr[q];
      float* o = xT + (40 * hf + 4 * q) * 32 + j;
      o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
    }
  }
  const f32x4 vv = reinterpret_cast<const f32x4*>(A.vec)[ec];
  const float v[3] = {vv.x, vv.y, vv.z};
  const float* gm = A.gmsg + (size_t)ec * NODE_STRIDE;
  float g0e[16], g1o[9], g1e[9], g0o[3];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 x = reinterpret_cast<const f32x4*>(gm + 4 * hf)[2 * q];
    g0e[4 * q + 0] = x.x; g0e[4 * q + 1] = x.y; g0e[4 * q + 2] = x.z; g0e[4 * q + 3] = x.w;
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    g1o[k] = gm[COL_1O + 9 * hf + k];
    g1e[k] = OUT >= 2 ? gm[COL_1E + 9 * hf + k] : 0.f;
  }
#pragma unroll
  for (int o = 0; o < 3; ++o) g0o[o] = OUT >= 3 ? gm[COL_0O + 3 * hf + o] : 0.f;
  __syncthreads();

  const float* xc = xT + j;
  f32x16 G0, G1, G2;
#pragma unroll
  for (int r = 0; r < 16; ++r) { G0[r] = 0.f; G1[r] = 0.f; G2[r] = 0.f; }
  int T = 0;
#define CBD_GT(GW)                                                                                   \
  {                                                                                                  \
    gemm_gh(a, gp + (size_t)(T + 1) * OpsF32::TILE_FRAGS, lane, GW, G0, G1, G2);                     \
    ++T;                                                                                             \
  }
#pragma unroll 1
  for (int i = 0; i < S.t0e; ++i) {
    const float m = mid0e<IN>(xc, i, v);
    float gw[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) gw[r] = m * g0e[r];
    CBD_GT(gw);
  }
  auto vec_block = [&](auto mid_fn, int ntile, const float (&gk)[9]) __attribute__((always_inline)) {
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
      float gw[16];
      gw[15] = 0.f;
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        float m[3];
        mid_fn(xc, VEC_TILE_I * t + q, v, m);
#pragma unroll
        for (int o = 0; o < 3; ++o) gw[3 * q + o] = m[0] * gk[3 * o + 0] + m[1] * gk[3 * o + 1] + m[2] * gk[3 * o + 2];
      }
      CBD_GT(gw);
    }
  };
  vec_block([](const float* x, int i, const float (&u)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, u, m); }, S.t1o, g1o);
  if constexpr (OUT >= 2)
    vec_block([](const float* x, int i, const float (&u)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, u, m); }, S.t1e, g1e);
  if constexpr (OUT >= 3) {
#pragma unroll 1
    for (int t = 0; t < S.t0o; ++t) {
      float gw[16];
      gw[15] = 0.f;
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const float m = mid0o<IN>(xc, VEC_TILE_I * t + q, v);
#pragma unroll
        for (int o = 0; o < 3; ++o) gw[3 * q + o] = m * g0o[o];
      }
      CBD_GT(gw);
    }
  }
#undef CBD_GT
  if (valid) {
    float* out = A.gh + (size_t)e * KDIM + 4 * hf;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      reinterpret_cast<f32x4*>(out)[2 * q] = f32x4{G0[4 * q], G0[4 * q + 1], G0[4 * q + 2], G0[4 * q + 3]};
      reinterpret_cast<f32x4*>(out + 32)[2 * q] = f32x4{G1[4 * q], G1[4 * q + 1], G1[4 * q + 2], G1[4 * q + 3]};
      reinterpret_cast<f32x4*>(out + 64)[2 * q] = f32x4{G2[4 * q], G2[4 * q + 1], G2[4 * q + 2], G2[4 * q + 3]};
    }
  }
}

template <int IN, int OUT>
static hipError_t launch_train_gh(const TrainTpArgs& a, hipStream_t s) {
  int grid = 0;
  for (int g = 0; g < a.n_groups; ++g) grid += (a.e_begin[g + 1] - a.e_begin[g] + WAVE_EDGES - 1) / WAVE_EDGES;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL((tp_train_gh_kernel<IN, OUT>), dim3(grid), dim3(64), XT_FLOATS * 4, s, a);
  return hipGetLastError();
}

static hipError_t launch_train_gh_any(int in_level, int out_level, const TrainTpArgs& a, hipStream_t s) {
  if (in_level == 0 && out_level == 1) return launch_train_gh<0, 1>(a, s);
  if (in_level == 1 && out_level == 2) return launch_train_gh<1, 2>(a, s);
  if (in_level == 2 && out_level == 3) return launch_train_gh<2, 3>(a, s);
  if (in_level == 3 && out_level == 3) return launch_train_gh<3, 3>(a, s);
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------------------
// dW2p = g_w^T h and db2p = column sums of g_w WITHOUT g_w in memory: the weight / bias gradient of the FCBlock's second Linear
// (autograd of fc[3], models/layers.py:8-15, under utils/training.py:205) with the EDGES as the MFMA k dimension.
// A workgroup of four waves owns four consecutive weight tiles (one per wave: 150 registers, three residents per SIMD; two tiles per
// wave share the B operand reads but leave two residents: 0.45 against 0.50 of peak) and one chunk of the group's edges.  Per 32-edge block
// the four waves stage h, g_msg, the gathered rows (transposed) and the edge vectors in LDS once; every wave then re-forms ITS tile of
// g_w on the VALU -- lane (rho, hf) needs g_w[edge 2s + hf][row rho] for k-step s: mid(edge) x g_msg(edge), the same products
// tp_train_bwd_kernel forms, here with the roles of lanes and registers exchanged -- as the A operand, takes h[edge][32 kb + n] as the
// B operand, and accumulates its [32 rows x 96] block of dW2p in 48 registers over the whole chunk; db2p is the running sum of the A
// operand.  One partial block per (chunk, tile); the caller adds the chunks (fixed order: bitwise repeatable).
struct TrainDwArgs {
  const float* xrow;     // [E][NODE_STRIDE]
  const float* vec;      // [E][4]
  const float* h;        // [E][96]
  const float* gmsg;     // [E][NODE_STRIDE]
  // edge groups of the layer: group g owns edges [g_lo[g], g_hi[g]) and the chunks [g_chunk0[g], g_chunk0[g + 1]) of the launch's
  // second grid dimension, g_bpc[g] 32-edge blocks per chunk (one launch for all groups of a layer: cbd_tp_backward_dw_groups)
  int n_groups;
  int g_lo[TRAIN_MAX_GROUPS], g_hi[TRAIN_MAX_GROUPS], g_bpc[TRAIN_MAX_GROUPS], g_chunk0[TRAIN_MAX_GROUPS + 1];
  float* partial;        // [total chunks][wp * 96 + wp]
};
constexpr int DW_GS = 96;                         // row stride of the staged g_msg rows: the two lane halves (edges e, e + 1) hit disjoint banks
constexpr int DW_TILES_PER_WG = 4;                // 4 waves x 1 tile (two tiles per wave: 218 registers, two residents per SIMD, 0.45 of peak)
constexpr int DW_MT = 16 * 32;                    // mid table of one tile slot: [15 (+1)][32 edges]
constexpr int DW_LDS_FLOATS = 32 * KDIM + 32 * DW_GS + NODE_STRIDE * 32 + 32 * 4 + DW_TILES_PER_WG * DW_MT;

// the (up to) 15 mid components a tile needs for edge column xc: a scalar-block tile has ONE scalar mid (rows 1, 2 stay zero), a
// vector-block tile five 3-vectors (0e / 0o block mids are scalars: component 0)
template <int IN, int OUT>
__device__ __forceinline__ void dw_tile_mid(int T, int q, const float* xc, const float (&v)[3], float (&m)[3]) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  m[0] = m[1] = m[2] = 0.f;
  if (T < S.t0e) {
    if (q == 0) m[0] = mid0e<IN>(xc, T, v);
  } else if (T < S.t0e + S.t1o) {
    mid1o<IN>(xc, VEC_TILE_I * (T - S.t0e) + q, v, m);
  } else if (OUT >= 2 && T < S.t0e + S.t1o + S.t1e) {
    mid1e<IN>(xc, VEC_TILE_I * (T - S.t0e - S.t1o) + q, v, m);
  } else {
    m[0] = mid0o<IN>(xc, VEC_TILE_I * (T - S.t0e - S.t1o - S.t1e) + q, v);
  }
}

// per-lane recipe of the A operand for tile T: a = sum_c M[mrow + c][e] * g[e][gcol + c]   (c < 3; unused terms meet zeros)
template <int IN, int OUT>
__device__ __forceinline__ void dw_recipe(int T, int rho, int& mrow, int& gcol, float& live) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  const int vr = 4 * (rho >> 3) + (rho & 3), vh = (rho >> 2) & 1, vq = vr / 3, vo = vr % 3;     // vector rows: r = 3 q + o of half vh
  live = 1.f;
  if (T < S.t0e) { mrow = 0; gcol = rho; return; }                                              // row rho = output channel rho
  live = vr < 15 ? 1.f : 0.f;
  mrow = 3 * vq;
  if (T < S.t0e + S.t1o) gcol = COL_1O + 9 * vh + 3 * vo;
  else if (OUT >= 2 && T < S.t0e + S.t1o + S.t1e) gcol = COL_1E + 9 * vh + 3 * vo;
  else { gcol = COL_0O + 3 * vh + vo; }                                                          // scalar x scalar: component 0 only
  if (vr >= 15) { mrow = 0; gcol = 0; }
}

template <int IN, int OUT>
__global__ __launch_bounds__(256) void tp_train_dw_kernel(TrainDwArgs A) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  constexpr int NTW = S.ntiles - 3, WP = NTW * 32;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* hS = lds;                              // [32][96]
  float* gS = hS + 32 * KDIM;                   // [32][DW_GS]  rows of invalid edges are zero => their g_w is zero
  float* xT = gS + 32 * DW_GS;                  // [80][32]
  float* vS = xT + NODE_STRIDE * 32;            // [32][4]
  float* mT = vS + 32 * 4;                      // [tile slots][16][32]  mids of the block's edges
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, rho = lane & 31, hf = lane >> 5;
  // XCD-aware mapping: the workgroups of one edge chunk (one per tile group) all read the same h / g_msg / row blocks.  Workgroups go
  // round-robin over the 8 XCDs by their linear id, each XCD with an L2 of its own: the tile groups of a chunk are therefore given linear
  // ids that are congruent mod 8, so that a chunk's edge data is fetched into ONE L2 instead of eight (grid.y is padded to a multiple
  // of 8 chunks; the padding workgroups exit).
  const int lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, slot = lin >> 3;
  const int bx = slot % (int)gridDim.x, by = (slot / (int)gridDim.x) * 8 + xcd;
  if (by >= A.g_chunk0[A.n_groups]) return;
  const int Tb = bx * DW_TILES_PER_WG;          // first tile of this workgroup
  const int T0 = __builtin_amdgcn_readfirstlane(Tb + wave);                  // this wave's second-Linear tile (wave-uniform)
  const bool live0 = T0 < NTW;
  int mrow0, gcol0;
  float lv0;
  dw_recipe<IN, OUT>(live0 ? T0 : NTW - 1, rho, mrow0, gcol0, lv0);
  if (!live0) lv0 = 0.f;
  // scalar-block tiles (0e, 0o) multiply a scalar mid with a scalar gradient: only component 0 of the generic three-term form
  const bool three0 = T0 >= S.t0e && T0 < S.t0e + S.t1o + S.t1e;   // vector-block tile
  const float* m0p = mT + wave * DW_MT + mrow0 * 32;
  f32x16 a00, a01, a02;
#pragma unroll
  for (int r = 0; r < 16; ++r) { a00[r] = 0.f; a01[r] = 0.f; a02[r] = 0.f; }
  float db0 = 0.f;
  int grp = 0;
#pragma unroll
  for (int g = 1; g < TRAIN_MAX_GROUPS; ++g)
    if (g < A.n_groups && by >= A.g_chunk0[g]) grp = g;
  const int e_lo = A.g_lo[grp], e_hi = A.g_hi[grp], bpc = A.g_bpc[grp];
  const int b_lo = (by - A.g_chunk0[grp]) * bpc;
  const int n_blk = min(bpc, (e_hi - e_lo + 31) / 32 - b_lo);
  // staging through registers: block b + 1 is in flight from global memory while block b is multiplied
  f32x4 hr[3], gr[3], xr[3], vr4;
  auto fetch = [&](int e0) {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int k = tid + 256 * it;                                   // h: 768 float4, coalesced rows
      const int e = min(e0 + k / (KDIM / 4), e_hi - 1);
      hr[it] = reinterpret_cast<const f32x4*>(A.h + (size_t)e * KDIM)[k % (KDIM / 4)];
      if (k < 32 * NODE_STRIDE / 4) {                                 // g_msg: 640 float4, coalesced rows
        const int row = k / (NODE_STRIDE / 4);
        const bool ok = e0 + row < e_hi;
        const f32x4 g = reinterpret_cast<const f32x4*>(A.gmsg + (size_t)(ok ? e0 + row : e_hi - 1) * NODE_STRIDE)[k % (NODE_STRIDE / 4)];
        gr[it] = ok ? g : f32x4{0.f, 0.f, 0.f, 0.f};
        const int xrow_i = k & 31, xc4 = k >> 5;                      // gathered rows: edge index fastest => conflict-free transposed stores
        xr[it] = reinterpret_cast<const f32x4*>(A.xrow + (size_t)min(e0 + xrow_i, e_hi - 1) * NODE_STRIDE)[xc4];
      }
    }
    if (tid < 32) vr4 = reinterpret_cast<const f32x4*>(A.vec)[min(e0 + tid, e_hi - 1)];
  };
  auto stage = [&]() {
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int k = tid + 256 * it;
      reinterpret_cast<f32x4*>(hS)[k] = hr[it];
      if (k < 32 * NODE_STRIDE / 4) {
        *reinterpret_cast<f32x4*>(gS + (k / (NODE_STRIDE / 4)) * DW_GS + 4 * (k % (NODE_STRIDE / 4))) = gr[it];
        float* o = xT + (4 * (k >> 5)) * 32 + (k & 31);
        o[0] = xr[it].x; o[32] = xr[it].y; o[64] = xr[it].z; o[96] = xr[it].w;
      }
    }
    if (tid < 32) reinterpret_cast<f32x4*>(vS)[tid] = vr4;
  };
  for (int k = tid; k < DW_TILES_PER_WG * DW_MT; k += 256) mT[k] = 0.f;      // rows a scalar-block tile never writes stay zero
  if (n_blk > 0) fetch(e_lo + b_lo * 32);
  for (int bk = 0; bk < n_blk; ++bk) {
    __syncthreads();                            // the previous block's readers are done
    stage();
    __syncthreads();
    if (bk + 1 < n_blk) fetch(e_lo + (b_lo + bk + 1) * 32);
    // mids of this block for the workgroup's tile slots: (slot, q, edge) -> 3 components; 4 x 5 x 32 items over 256 threads
#pragma unroll 1
    for (int it = tid; it < DW_TILES_PER_WG * VEC_TILE_I * 32; it += 256) {
      const int e = it & 31, q = (it >> 5) % VEC_TILE_I, slot = it / (VEC_TILE_I * 32);
      const int T = min(Tb + slot, NTW - 1);
      if (T < S.t0e && q > 0) continue;         // a 0e tile has one scalar mid
      const float v[3] = {vS[4 * e], vS[4 * e + 1], vS[4 * e + 2]};
      float m[3];
      dw_tile_mid<IN, OUT>(T, q, xT + e, v, m);
      float* o = mT + slot * DW_MT + (3 * q) * 32 + e;
      o[0] = m[0]; o[32] = m[1]; o[64] = m[2];
    }
    __syncthreads();
    // branch-free k-steps: a = sum_c M[mrow + c][e] g[e][gcol + c]; lanes of dead rows / dead tile slots multiply by zero
#pragma unroll 8
    for (int s = 0; s < 16; ++s) {
      const int e = 2 * s + hf;                 // this lane's edge of k-step s
      const float* ge = gS + e * DW_GS;
      const float* hb = hS + e * KDIM + rho;    // B operand: lane (n = rho, hf) supplies h[edge 2s + hf][32 kb + n]
      const float b0 = hb[0], b1 = hb[32], b2 = hb[64];
      float x0 = m0p[e] * ge[gcol0];
      if (three0) x0 += m0p[32 + e] * ge[gcol0 + 1] + m0p[64 + e] * ge[gcol0 + 2];
      x0 *= lv0;
      a00 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, b0, a00, 0, 0, 0);
      a01 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, b1, a01, 0, 0, 0);
      a02 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, b2, a02, 0, 0, 0);
      db0 += x0;
    }
  }
  float* out = A.partial + (size_t)by * (WP * KDIM + WP);
  // D layout: lane (n, hf) holds rows (r & 3) + 8 (r >> 2) + 4 hf of column n
  auto store = [&](int T, const f32x16& c0, const f32x16& c1, const f32x16& c2, float db) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = T * 32 + (r & 3) + 8 * (r >> 2) + 4 * hf;
      out[(size_t)row * KDIM + rho] = c0[r];
      out[(size_t)row * KDIM + 32 + rho] = c1[r];
      out[(size_t)row * KDIM + 64 + rho] = c2[r];
    }
    const float other = __shfl_xor(db, 32);
    if (hf == 0) out[(size_t)WP * KDIM + T * 32 + rho] = db + other;
  };
  if (live0) store(T0, a00, a01, a02, db0);
}

template <int IN, int OUT>
static hipError_t launch_train_dw(const TrainDwArgs& a, int n_chunks, hipStream_t s) {
  constexpr int NTW = conv_shape(IN, OUT).ntiles - 3;
  hipLaunchKernelGGL((tp_train_dw_kernel<IN, OUT>), dim3((NTW + DW_TILES_PER_WG - 1) / DW_TILES_PER_WG, (n_chunks + 7) / 8 * 8), dim3(256), DW_LDS_FLOATS * 4, s, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Weight / bias gradient of the FCBlock's FIRST Linear (96 -> 96):  dW[m][n] = sum_e G[e][m] X[e][n],  db[m] = sum_e G[e][m]
// (G = d loss / d pre-activation, X = [edge_attr | x_src[:32] | x_dst[:32]]).  A reduction over 10^5..10^6 edges into a 96 x 96
// output: library GEMMs run it at ~10 TFLOP/s (K huge, M = N = 96: 341 us per call in profiles/r01_h_train_b32_kernel_stats.csv),
// it is bound by reading the two [E, 96] operands once.  Here every wave owns a contiguous chunk of edges, feeds them as the K
// dimension of v_mfma_f32_32x32x2_f32 (two edges per k-step; lane (c = lane & 31, hf = lane >> 5) loads G[e + hf][32 i + c] and
// X[e + hf][32 j + c]: coalesced 128-byte rows) into nine 32 x 32 accumulators, and writes one partial [96 x 96 + 96] block; the
// caller adds the blocks (deterministic: no atomics).
constexpr int OUTER_DIM = KDIM;
constexpr int OUTER_PART_FLOATS = OUTER_DIM * OUTER_DIM + OUTER_DIM;
__global__ __launch_bounds__(64, 2) void outer_accum_kernel(const float* __restrict__ G, const float* __restrict__ X, int E, int chunk,
                                                             float* __restrict__ partial) {
  const int lane = threadIdx.x, c = lane & 31, hf = lane >> 5;
  const int e_lo = blockIdx.x * chunk, e_hi = min(E, e_lo + chunk);
  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float colsum[3] = {0.f, 0.f, 0.f};
  for (int e = e_lo; e < e_hi; e += 2) {
    const int row = e + hf;
    const bool ok = row < e_hi;
    const size_t off = (size_t)(ok ? row : e) * OUTER_DIM + c;
    float g[3], x[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const float gv = G[off + 32 * i], xv = X[off + 32 * i];
      g[i] = ok ? gv : 0.f;
      x[i] = ok ? xv : 0.f;
      colsum[i] += g[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[i], x[j], acc[i][j], 0, 0, 0);
  }
  float* out = partial + (size_t)blockIdx.x * OUTER_PART_FLOATS;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf;
        out[(size_t)m * OUTER_DIM + 32 * j + c] = acc[i][j][r];
      }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float t = colsum[i] + __shfl_xor(colsum[i], 32, 64);
    if (hf == 0) out[OUTER_DIM * OUTER_DIM + 32 * i + c] = t;
  }
}

}  // namespace cbd

namespace cbd {

// Deterministic segmented sum: out[n][c] = sum_{k in [rowptr[n], rowptr[n+1])} vals[perm[k]][c], added in index order (no atomics).
// Stands in for torch_scatter.scatter(..., reduce='sum') in TensorProductConvLayer.forward (reference models/tensor_layers.py:206;
// the mean divides afterwards) and for the backward of the node gathers `node_attr[edge_dst]` (an index_add in autograd) in the
// fine-tuning step: with every scatter in a fixed order the training step is bitwise repeatable.
// One wave per output row, lanes over columns (rows of <= 128 floats are read as coalesced segments), 4 gathered rows in flight.
template <bool MEAN>
__global__ __launch_bounds__(256) void segment_sum_kernel(int64_t n_rows, int width, const float* __restrict__ vals,
                                                          const int64_t* __restrict__ perm, const int64_t* __restrict__ rowptr,
                                                          float* __restrict__ out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int64_t lo = rowptr[row], hi = rowptr[row + 1];
  for (int c = lane; c < width; c += 64) {
    float acc = 0.f;
    int64_t k = lo;
    for (; k + 4 <= hi; k += 4) {
      const int64_t p0 = perm[k], p1 = perm[k + 1], p2 = perm[k + 2], p3 = perm[k + 3];
      const float v0 = vals[p0 * width + c], v1 = vals[p1 * width + c], v2 = vals[p2 * width + c], v3 = vals[p3 * width + c];
      acc = ((acc + v0) + v1) + v2;
      acc += v3;
    }
    for (; k < hi; ++k) acc += vals[perm[k] * width + c];
    if (MEAN) acc = acc / (float)(hi - lo > 1 ? hi - lo : 1);      // torch_scatter's mean: sum / clamp(count, min = 1)
    out[row * width + c] = acc;
  }
}

// The same for FEW, LONG rows (sums per graph: 8 rows of 9 216 edge rows each in the training step -- one wave per row left the GPU to
// eight waves for a millisecond): one workgroup of W waves per row, wave w sums the w-th contiguous part of the row's range in index
// order, the W partial sums are added in wave order.  Fixed association: bitwise repeatable (it differs from the one-wave kernel's).
template <bool MEAN, int W>
__global__ __launch_bounds__(64 * W) void segment_sum_long_kernel(int64_t n_rows, int width, const float* __restrict__ vals,
                                                                  const int64_t* __restrict__ perm, const int64_t* __restrict__ rowptr,
                                                                  float* __restrict__ out) {
  __shared__ float part[W][64];
  const int64_t row = blockIdx.x;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t lo = rowptr[row], hi = rowptr[row + 1], len = hi - lo;
  const int64_t a = lo + len * w / W, b = lo + len * (w + 1) / W;
  for (int c0 = 0; c0 < width; c0 += 64) {
    const int c = c0 + lane;
    float acc = 0.f;
    if (c < width) {
      int64_t k = a;
      for (; k + 4 <= b; k += 4) {
        const int64_t p0 = perm[k], p1 = perm[k + 1], p2 = perm[k + 2], p3 = perm[k + 3];
        const float v0 = vals[p0 * width + c], v1 = vals[p1 * width + c], v2 = vals[p2 * width + c], v3 = vals[p3 * width + c];
        acc = ((acc + v0) + v1) + v2;
        acc += v3;
      }
      for (; k < b; ++k) acc += vals[perm[k] * width + c];
    }
    part[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < width) {
      float s = part[0][lane];
#pragma unroll
      for (int i = 1; i < W; ++i) s += part[i][lane];
      if (MEAN) s = s / (float)(len > 1 ? len : 1);
      out[row * width + c] = s;
    }
    __syncthreads();
  }
}

template <bool MEAN>
static hipError_t launch_segment_sum(int64_t n_rows, int width, const float* vals, const int64_t* perm, const int64_t* rowptr, float* out,
                                     hipStream_t s) {
  // few rows = long rows in this code's uses (sums per graph, per ligand atom); the row count alone decides the kernel
  if (n_rows <= 64)
    hipLaunchKernelGGL((segment_sum_long_kernel<MEAN, 16>), dim3((unsigned)n_rows), dim3(1024), 0, s, n_rows, width, vals, perm, rowptr, out);
  else if (n_rows <= 1024)
    hipLaunchKernelGGL((segment_sum_long_kernel<MEAN, 4>), dim3((unsigned)n_rows), dim3(256), 0, s, n_rows, width, vals, perm, rowptr, out);
  else
    hipLaunchKernelGGL(segment_sum_kernel<MEAN>, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, s, n_rows, width, vals, perm, rowptr, out);
  return hipGetLastError();
}

}  // namespace cbd

extern "C" {

int64_t cbd_tp_packed_width(int32_t in_level, int32_t out_level) {
  if (in_level < 0 || in_level > 3 || out_level < 1 || out_level > 3) return -1;
  return (int64_t)(cbd::conv_shape(in_level, out_level).ntiles - 3) * 32;
}

static int fill_groups(cbd::TrainTpArgs& a, int32_t n_groups, const int64_t* group_edges, const float* const* wstreams) {
  if (n_groups < 1 || n_groups > cbd::TRAIN_MAX_GROUPS || !group_edges || !wstreams) return fail(CBD_ERR_ARG, "1..4 edge groups");
  int64_t e = 0;
  a.n_groups = n_groups;
  for (int g = 0; g < cbd::TRAIN_MAX_GROUPS + 1; ++g) a.e_begin[g] = 0;
  for (int g = 0; g < cbd::TRAIN_MAX_GROUPS; ++g) a.wstream[g] = nullptr;
  for (int g = 0; g < n_groups; ++g) {
    if (group_edges[g] < 0 || (group_edges[g] > 0 && !wstreams[g])) return fail(CBD_ERR_ARG, "bad edge group %d", g);
    a.e_begin[g] = (int)e;
    a.wstream[g] = wstreams[g];
    e += group_edges[g];
    if (e > (int64_t)1 << 30) return fail(CBD_ERR_ARG, "too many edges");
  }
  for (int g = n_groups; g <= cbd::TRAIN_MAX_GROUPS; ++g) a.e_begin[g] = (int)e;
  a.E = (int)e;
  return 0;
}

int cbd_tp_forward(int32_t in_level, int32_t out_level, int32_t n_groups, const int64_t* group_edges, const float* xrow_dev,
                   const float* vec4_dev, const float* h_dev, const float* const* wstreams_dev, float* msg_dev, void* stream) {
  cbd::TrainTpArgs a{};
  CHK(fill_groups(a, n_groups, group_edges, wstreams_dev));
  if (a.E == 0) return 0;
  if (!xrow_dev || !vec4_dev || !h_dev || !msg_dev) return fail(CBD_ERR_ARG, "null argument");
  a.xrow = xrow_dev; a.vec = vec4_dev; a.h = h_dev; a.msg = msg_dev;
  const hipError_t r = cbd::launch_train_any(in_level, out_level, false, a, reinterpret_cast<hipStream_t>(stream));
  if (r != hipSuccess) return fail(r == hipErrorInvalidValue ? CBD_ERR_ARG : CBD_ERR_HIP, "cbd_tp_forward: %s", hipGetErrorString(r));
  return 0;
}

int cbd_tp_backward(int32_t in_level, int32_t out_level, int32_t n_groups, const int64_t* group_edges, const float* xrow_dev,
                    const float* vec4_dev, const float* h_dev, const float* const* wstreams_dev, const float* gmsg_dev, float* gx_dev,
                    float* gw_dev, void* stream) {
  cbd::TrainTpArgs a{};
  CHK(fill_groups(a, n_groups, group_edges, wstreams_dev));
  if (a.E == 0) return 0;
  if (!xrow_dev || !vec4_dev || !h_dev || !gmsg_dev || !gx_dev) return fail(CBD_ERR_ARG, "null argument");
  a.xrow = xrow_dev; a.vec = vec4_dev; a.h = h_dev; a.gmsg = gmsg_dev; a.gx = gx_dev; a.gw = gw_dev;      /* gw_dev may be NULL */
  const hipError_t r = cbd::launch_train_any(in_level, out_level, true, a, reinterpret_cast<hipStream_t>(stream));
  if (r != hipSuccess) return fail(r == hipErrorInvalidValue ? CBD_ERR_ARG : CBD_ERR_HIP, "cbd_tp_backward: %s", hipGetErrorString(r));
  return 0;
}

int cbd_tp_backward_gh(int32_t in_level, int32_t out_level, int32_t n_groups, const int64_t* group_edges, const float* xrow_dev,
                       const float* vec4_dev, const float* const* wstreams_t_dev, const float* gmsg_dev, float* gh_dev, void* stream) {
  cbd::TrainTpArgs a{};
  CHK(fill_groups(a, n_groups, group_edges, wstreams_t_dev));
  if (a.E == 0) return 0;
  if (!xrow_dev || !vec4_dev || !gmsg_dev || !gh_dev) return fail(CBD_ERR_ARG, "null argument");
  a.xrow = xrow_dev; a.vec = vec4_dev; a.gmsg = gmsg_dev; a.gh = gh_dev;
  const hipError_t r = cbd::launch_train_gh_any(in_level, out_level, a, reinterpret_cast<hipStream_t>(stream));
  if (r != hipSuccess) return fail(r == hipErrorInvalidValue ? CBD_ERR_ARG : CBD_ERR_HIP, "cbd_tp_backward_gh: %s", hipGetErrorString(r));
  return 0;
}

static int launch_dw_any(int32_t in_level, int32_t out_level, const cbd::TrainDwArgs& a, int total_chunks, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipError_t r = hipErrorInvalidValue;
  if (in_level == 0 && out_level == 1) r = cbd::launch_train_dw<0, 1>(a, total_chunks, st);
  else if (in_level == 1 && out_level == 2) r = cbd::launch_train_dw<1, 2>(a, total_chunks, st);
  else if (in_level == 2 && out_level == 3) r = cbd::launch_train_dw<2, 3>(a, total_chunks, st);
  else if (in_level == 3 && out_level == 3) r = cbd::launch_train_dw<3, 3>(a, total_chunks, st);
  if (r != hipSuccess) return fail(r == hipErrorInvalidValue ? CBD_ERR_ARG : CBD_ERR_HIP, "cbd_tp_backward_dw: %s", hipGetErrorString(r));
  return 0;
}

int cbd_tp_backward_dw(int32_t in_level, int32_t out_level, int64_t e_lo, int64_t e_hi, const float* xrow_dev, const float* vec4_dev,
                       const float* h_dev, const float* gmsg_dev, int32_t n_chunks, float* partial_dev, void* stream) {
  if (e_lo < 0 || e_hi < e_lo || e_hi > ((int64_t)1 << 30) || n_chunks <= 0) return fail(CBD_ERR_ARG, "cbd_tp_backward_dw: bad argument");
  if (e_hi == e_lo) return 0;
  if (!xrow_dev || !vec4_dev || !h_dev || !gmsg_dev || !partial_dev) return fail(CBD_ERR_ARG, "null argument");
  cbd::TrainDwArgs a{};
  a.xrow = xrow_dev; a.vec = vec4_dev; a.h = h_dev; a.gmsg = gmsg_dev; a.partial = partial_dev;
  a.n_groups = 1; a.g_lo[0] = (int)e_lo; a.g_hi[0] = (int)e_hi; a.g_chunk0[0] = 0;
  for (int g = 1; g <= cbd::TRAIN_MAX_GROUPS; ++g) a.g_chunk0[g] = n_chunks;
  const int64_t blocks = (e_hi - e_lo + 31) / 32;
  a.g_bpc[0] = (int)((blocks + n_chunks - 1) / n_chunks);
  return launch_dw_any(in_level, out_level, a, n_chunks, stream);
}

int cbd_tp_backward_dw_groups(int32_t in_level, int32_t out_level, int32_t n_groups, const int64_t* group_edges, const int32_t* n_chunks,
                              const float* xrow_dev, const float* vec4_dev, const float* h_dev, const float* gmsg_dev, float* partial_dev,
                              void* stream) {
  if (n_groups < 1 || n_groups > cbd::TRAIN_MAX_GROUPS || !group_edges || !n_chunks) return fail(CBD_ERR_ARG, "cbd_tp_backward_dw_groups: 1..4 edge groups");
  if (!xrow_dev || !vec4_dev || !h_dev || !gmsg_dev || !partial_dev) return fail(CBD_ERR_ARG, "null argument");
  cbd::TrainDwArgs a{};
  a.xrow = xrow_dev; a.vec = vec4_dev; a.h = h_dev; a.gmsg = gmsg_dev; a.partial = partial_dev; a.n_groups = n_groups;
  int64_t e = 0;
  int chunks = 0;
  for (int g = 0; g < n_groups; ++g) {
    if (group_edges[g] <= 0 || n_chunks[g] <= 0) return fail(CBD_ERR_ARG, "cbd_tp_backward_dw_groups: bad group %d", g);
    a.g_lo[g] = (int)e; e += group_edges[g]; a.g_hi[g] = (int)e;
    if (e > (int64_t)1 << 30) return fail(CBD_ERR_ARG, "too many edges");
    a.g_chunk0[g] = chunks; chunks += n_chunks[g];
    const int64_t blocks = (group_edges[g] + 31) / 32;
    a.g_bpc[g] = (int)((blocks + n_chunks[g] - 1) / n_chunks[g]);
  }
  for (int g = n_groups; g <= cbd::TRAIN_MAX_GROUPS; ++g) a.g_chunk0[g] = chunks;
  return launch_dw_any(in_level, out_level, a, chunks, stream);
}

int64_t cbd_outer_accum_part_floats(void) { return cbd::OUTER_PART_FLOATS; }

int cbd_outer_accum(int64_t E, const float* g_dev, const float* x_dev, int32_t n_parts, float* partial_dev, void* stream) {
  if (E <= 0 || E > (int64_t)1 << 30 || n_parts <= 0 || !g_dev || !x_dev || !partial_dev) return fail(CBD_ERR_ARG, "bad argument");
  int chunk = (int)((E + n_parts - 1) / n_parts);
  chunk += chunk & 1;   // even: a k-step's two edges belong to one chunk
  hipLaunchKernelGGL(cbd::outer_accum_kernel, dim3(n_parts), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), g_dev, x_dev, (int)E,
                     chunk, partial_dev);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_outer_accum: %s", hipGetErrorString(r));
  return 0;
}

int cbd_segment_sum(int64_t n_rows, int32_t width, const float* vals_dev, const int64_t* perm_dev, const int64_t* rowptr_dev,
                    float* out_dev, void* stream) {
  if (n_rows < 0 || width <= 0 || !rowptr_dev || !out_dev) return fail(CBD_ERR_ARG, "bad argument");
  if (n_rows == 0) return 0;
  const hipError_t r = cbd::launch_segment_sum<false>(n_rows, (int)width, vals_dev, perm_dev, rowptr_dev, out_dev, reinterpret_cast<hipStream_t>(stream));
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_segment_sum: %s", hipGetErrorString(r));
  return 0;
}

int cbd_segment_mean(int64_t n_rows, int32_t width, const float* vals_dev, const int64_t* perm_dev, const int64_t* rowptr_dev,
                     float* out_dev, void* stream) {
  if (n_rows < 0 || width <= 0 || !rowptr_dev || !out_dev) return fail(CBD_ERR_ARG, "bad argument");
  if (n_rows == 0) return 0;
  const hipError_t r = cbd::launch_segment_sum<true>(n_rows, (int)width, vals_dev, perm_dev, rowptr_dev, out_dev, reinterpret_cast<hipStream_t>(stream));
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_segment_mean: %s", hipGetErrorString(r));
  return 0;
}

}  // extern "C"
