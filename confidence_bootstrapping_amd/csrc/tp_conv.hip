// Fused tensor-product message passing for gfx950 (MI355X):
//   per edge:  h = ReLU(W1 [edge_attr | x_src[:32] | x_dst[:32]] + b1)            (FCBlock layer 1)
//              w = W2 h + b2                                                       (FCBlock layer 2, 1216..1660 wide)
//              msg = FasterTensorProduct(x_dst, sh(edge_vec), w)                   (lmax = 1 CG paths)
//   per node:  acc[src] += msg                                                     (mean/BN/residual: finalize kernel)
// replacing reference models/tensor_layers.py:195-206 (FCBlock -> FasterTensorProduct.forward:66-117 ->
// torch_scatter.scatter) without ever materialising the [E, W] per-edge weight tensor in HBM.
//
// Mapping onto the matrix cores (exact fp32, v_mfma_f32_32x32x2_f32):
//   D[row = weight column, col = edge] = sum_k  A[row][k] * B[k][edge]
//   * B (activations) lives in registers for the whole tile: lane (j = lane&31, hf = lane>>5) holds
//     act[edge j][k(s, hf)] for the 48 k-steps s.  The accumulator of the first Linear (after bias+ReLU) IS the
//     B operand of the second Linear -- the k order of W2 is permuted at weight-packing time to the C/D layout
//     row(reg, hf) = (reg&3) + 8*(reg>>2) + 4*hf, so nothing moves between the two GEMMs.
//   * A (weights) is streamed as 32-row tiles [12 x 64 lanes x float4 | 32 bias] through a double-buffered LDS
//     stage shared by the 4 waves of the workgroup (each wave owns 32 edges -> 128 edges per 12 KB of weights).
//   * Each finished 32x32 tile of w is consumed immediately by the CG contraction on the VALU: lane (j, hf)
//     multiplies its 16 accumulator rows with the "mid" value(s) of edge j (scalar, dot, cross or outer
//     products of x_dst and the unit edge vector, read from a per-wave LDS copy of the gathered row) and adds
//     into per-lane output accumulators.  1/sqrt(fan_in), sqrt(3) (sh scale) and 1/sqrt(2) factors are folded
//     into the packed weights.
//   * Rows of a vector-block tile hold 5 mid indices x 6 outputs: accumulator register reg < 15 of lane half hf is
//     (i = 5*tile + reg/3, o = 3*hf + reg%3), so both the mid index and the output slot are compile-time functions of
//     the register index, each half owns three of the six outputs, and 30 of the 32 MFMA rows carry weights.
//   * Segmented reduction: edges are sorted by aggregating node; each wave run-length sums its 32 messages in
//     LDS and issues one 256-byte-contiguous fp32 atomic add per (node run, 64 columns).
#include "common.h"

namespace cbd {

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// D = bias + A_tile * B ; A tile and bias read from LDS.
__device__ __forceinline__ void gemm_tile(const float* __restrict__ tile, const float (&B)[KSTEPS], f32x16& acc, int lane) {
  const int hf = lane >> 5;
  const f32x4* bp = reinterpret_cast<const f32x4*>(tile + TILE_W_FLOATS);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 b = bp[2 * q + hf];
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
  const f32x4* ap = reinterpret_cast<const f32x4*>(tile) + lane;
#pragma unroll
  for (int sg = 0; sg < KSTEPS / 4; ++sg) {
    f32x4 a = ap[sg * 64];
    acc = mfma32(a.x, B[4 * sg + 0], acc);
    acc = mfma32(a.y, B[4 * sg + 1], acc);
    acc = mfma32(a.z, B[4 * sg + 2], acc);
    acc = mfma32(a.w, B[4 * sg + 3], acc);
  }
}

struct Stage {  // one thread's share of a weight tile in flight: 3 x 16 B of weights (+ 16 B of bias for tid < 8)
  f32x4 w0, w1, w2, bb;
};

__device__ __forceinline__ void stage_load(Stage& st, const float* __restrict__ gtile, int tid) {
  const f32x4* p = reinterpret_cast<const f32x4*>(gtile);
  st.w0 = p[tid];
  st.w1 = p[tid + 256];
  st.w2 = p[tid + 512];
  if (tid < 8) st.bb = p[768 + tid];
}

__device__ __forceinline__ void stage_store(const Stage& st, float* __restrict__ ltile, int tid) {
  f32x4* p = reinterpret_cast<f32x4*>(ltile);
  p[tid] = st.w0;
  p[tid + 256] = st.w1;
  p[tid + 512] = st.w2;
  if (tid < 8) p[768 + tid] = st.bb;
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
constexpr int WAVE_LDS_FLOATS = XT_FLOATS + 32;          // + 32 ints of src ids
constexpr int CONV_LDS_FLOATS = 2 * TILE_FLOATS + CONV_WAVES * WAVE_LDS_FLOATS;
constexpr int CONV_LDS_BYTES = CONV_LDS_FLOATS * 4;

template <int IN, int OUT>
__global__ __launch_bounds__(256, 2) void tp_conv_kernel(ConvArgs args) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* abuf = lds;                                   // 2 x TILE_FLOATS
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int j = lane & 31, hf = lane >> 5;
  float* xT = lds + 2 * TILE_FLOATS + wave * WAVE_LDS_FLOATS;
  int* srcl = reinterpret_cast<int*>(xT + XT_FLOATS);

  // ---- which group / edge range does this workgroup own?  (edge counts live on the device)
  int grp = -1, e0 = 0, cnt = 0;
  {
    int t = blockIdx.x;
    for (int g = 0; g < args.n_groups; ++g) {
      const int c = *args.g[g].count;
      const int nt = (c + CONV_WG_EDGES - 1) / CONV_WG_EDGES;
      if (grp < 0) {
        if (t < nt) { grp = g; e0 = t * CONV_WG_EDGES; cnt = c; }
        else t -= nt;
      }
    }
  }
  if (grp < 0) return;
  const ConvGroup G = args.g[grp];
  const float* wst = G.wstream;

  // ---- gather the edge's inputs
  const int e = e0 + wave * WAVE_EDGES + j;
  const bool valid = e < cnt;
  int src = -1, dst = 0, aidx = 0;
  float v[3] = {0.f, 0.f, 0.f};
  if (valid) {
    src = G.src[e]; dst = G.dst[e]; aidx = G.attr_idx[e];
    const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[e];
    v[0] = vv.x; v[1] = vv.y; v[2] = vv.z;
  }
  if (hf == 0) srcl[j] = src;

  float Bx[KSTEPS];  // first-Linear input: [edge_attr(32) | x_src[:32] | x_dst[:32]], lane half hf holds cols 16hf..16hf+15
  {
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx * 32 + 16 * hf);
    const f32x4* ps = reinterpret_cast<const f32x4*>(args.node_in + (size_t)(src < 0 ? 0 : src) * NODE_STRIDE + 16 * hf);
    const f32x4* pd = reinterpret_cast<const f32x4*>(args.node_in + (size_t)dst * NODE_STRIDE + 16 * hf);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 a = valid ? pa[q] : z, s = valid ? ps[q] : z, d = valid ? pd[q] : z;
      Bx[4 * q + 0] = a.x; Bx[4 * q + 1] = a.y; Bx[4 * q + 2] = a.z; Bx[4 * q + 3] = a.w;
      Bx[16 + 4 * q + 0] = s.x; Bx[16 + 4 * q + 1] = s.y; Bx[16 + 4 * q + 2] = s.z; Bx[16 + 4 * q + 3] = s.w;
      Bx[32 + 4 * q + 0] = d.x; Bx[32 + 4 * q + 1] = d.y; Bx[32 + 4 * q + 2] = d.z; Bx[32 + 4 * q + 3] = d.w;
    }
    // full destination row -> transposed LDS copy xT[col][j]; lane half hf copies cols 40hf .. 40hf+39
    const f32x4* pr = reinterpret_cast<const f32x4*>(args.node_in + (size_t)dst * NODE_STRIDE + 40 * hf);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      const f32x4 r = valid ? pr[q] : z;
      float* o = xT + (40 * hf + 4 * q) * 32 + j;
      o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
    }
  }

  // ---- weight-tile pipeline prologue
  Stage st;
  stage_load(st, wst, tid);
  stage_store(st, abuf, tid);
  __syncthreads();

  int T = 0;
  f32x16 acc;
  float h1[KSTEPS];
  auto advance = [&](bool more) __attribute__((always_inline)) {   // finish tile T: publish tile T+1 and flip
    if (more) stage_store(st, abuf + ((T + 1) & 1) * TILE_FLOATS, tid);
    __syncthreads();
    ++T;
  };

  // ---- first Linear (3 tiles): h1 = ReLU(W1 x + b1), kept in the C/D register layout
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    stage_load(st, wst + (size_t)(T + 1) * TILE_FLOATS, tid);
    gemm_tile(abuf + (T & 1) * TILE_FLOATS, Bx, acc, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) h1[16 * m + r] = fmaxf(acc[r], 0.f);
    advance(true);
  }

  const float* xc = xT + j;
  // ---- block 0e: one tile per mid index, 32 output scalars
  float o0e[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) o0e[r] = 0.f;
#pragma unroll 1
  for (int i = 0; i < S.t0e; ++i) {
    const bool more = (T + 1) < S.ntiles;
    if (more) stage_load(st, wst + (size_t)(T + 1) * TILE_FLOATS, tid);
    gemm_tile(abuf + (T & 1) * TILE_FLOATS, h1, acc, lane);
    const float m = mid0e<IN>(xc, i, v);
#pragma unroll
    for (int r = 0; r < 16; ++r) o0e[r] = fmaf(m, acc[r], o0e[r]);
    advance(more);
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
      const bool more = (T + 1) < S.ntiles;
      if (more) stage_load(st, wst + (size_t)(T + 1) * TILE_FLOATS, tid);
      gemm_tile(abuf + (T & 1) * TILE_FLOATS, h1, acc, lane);
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
      advance(more);
    }
  };

  vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, vv, m); }, S.t1o, k1o);
  if constexpr (OUT >= 2)
    vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, vv, m); }, S.t1e, k1e);
  if constexpr (OUT >= 3) {
#pragma unroll 1
    for (int t = 0; t < S.t0o; ++t) {
      const bool more = (T + 1) < S.ntiles;
      if (more) stage_load(st, wst + (size_t)(T + 1) * TILE_FLOATS, tid);
      gemm_tile(abuf + (T & 1) * TILE_FLOATS, h1, acc, lane);
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const float m = mid0o<IN>(xc, VEC_TILE_I * t + q, v);
#pragma unroll
        for (int o = 0; o < 3; ++o) k0o[o] = fmaf(m, acc[3 * q + o], k0o[o]);
      }
      advance(more);
    }
  }

  // ---- messages -> LDS (re-using the gathered-row tile), then run-length sum per aggregating node
  // (all reads of xT by this wave are complete: they feed the FMAs above, and xT is private to the wave)
#pragma unroll
  for (int r = 0; r < 16; ++r) xT[((r & 3) + 8 * (r >> 2) + 4 * hf) * 32 + j] = o0e[r];
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xT[(COL_1O + 3 * (3 * hf + o) + c) * 32 + j] = k1o[3 * o + c];
      if constexpr (OUT >= 2) xT[(COL_1E + 3 * (3 * hf + o) + c) * 32 + j] = k1e[3 * o + c];
    }
  if constexpr (OUT >= 3) {
#pragma unroll
    for (int o = 0; o < 3; ++o) xT[(COL_0O + 3 * hf + o) * 32 + j] = k0o[o];
  }
  __syncthreads();
  for (int col = lane; col < S.out_dim; col += 64) {
    const float* oc = xT + col * 32;
    float sum = 0.f;
    int cur = srcl[0];
    for (int jj = 0; jj < 32; ++jj) {
      const int s = srcl[jj];
      if (s != cur) {
        if (cur >= 0) atomicAdd(args.acc + (size_t)cur * NODE_STRIDE + col, sum);
        sum = 0.f;
        cur = s;
      }
      sum += oc[jj];
    }
    if (cur >= 0) atomicAdd(args.acc + (size_t)cur * NODE_STRIDE + col, sum);
  }
}

// ------------------------------------------------------------------------------------------------------------
// mean -> e3nn BatchNorm (eval) -> residual, reference tensor_layers.py:206-216.
//   out[n][c] = bn(acc[n][c] / max(deg[n],1)) + (c < in_dim ? node_in[n][c] : 0) ; acc is cleared for the next layer.
// bn_scale[c] = weight * rsqrt(running_var + eps) per column, bn_shift[c] = bias - running_mean*scale (0e columns
// only, 0 elsewhere), prepared on the host per column.
__global__ void conv_finalize_kernel(float* __restrict__ acc, const float* __restrict__ node_in, float* __restrict__ node_out,
                                     const int* __restrict__ deg, const float* __restrict__ bn_scale,
                                     const float* __restrict__ bn_mean, const float* __restrict__ bn_bias,
                                     int n_nodes, int in_dim, int out_dim, int node_off) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = idx / NODE_STRIDE, c = idx % NODE_STRIDE;
  if (n >= n_nodes) return;
  const size_t o = (size_t)(n + node_off) * NODE_STRIDE + c;
  float r = 0.f;
  if (c < out_dim) {
    const int d = deg[n + node_off];
    float m = acc[o] / (float)(d > 1 ? d : 1);
    m = (m - bn_mean[c]) * bn_scale[c] + bn_bias[c];
    r = m + (c < in_dim ? node_in[o] : 0.f);
  }
  node_out[o] = r;
  acc[o] = 0.f;
}

// ---------------------------------------------------------------------------------------------- host launchers
template <int IN, int OUT>
static hipError_t launch_one(const ConvArgs& a, int grid, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv_kernel<IN, OUT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((tp_conv_kernel<IN, OUT>), dim3(grid), dim3(256), CONV_LDS_BYTES, s, a);
  return hipGetLastError();
}

hipError_t launch_tp_conv(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s) {
  if (grid <= 0) return hipSuccess;
  if (in_level == 0 && out_level == 1) return launch_one<0, 1>(a, grid, s);
  if (in_level == 1 && out_level == 2) return launch_one<1, 2>(a, grid, s);
  if (in_level == 2 && out_level == 3) return launch_one<2, 3>(a, grid, s);
  if (in_level == 3 && out_level == 3) return launch_one<3, 3>(a, grid, s);
  return hipErrorInvalidValue;
}

hipError_t launch_conv_finalize(float* acc, const float* node_in, float* node_out, const int* deg, const float* bn_scale,
                                const float* bn_mean, const float* bn_bias, int n_nodes, int in_dim, int out_dim,
                                int node_off, hipStream_t s) {
  if (n_nodes <= 0) return hipSuccess;
  const int total = n_nodes * NODE_STRIDE;
  hipLaunchKernelGGL(conv_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, acc, node_in, node_out, deg,
                     bn_scale, bn_mean, bn_bias, n_nodes, in_dim, out_dim, node_off);
  return hipGetLastError();
}

}  // namespace cbd
