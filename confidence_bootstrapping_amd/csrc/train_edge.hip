// Edge-row assembly of the fine-tuning step, forward and backward as one launch each -- the torch-op forms cost
// 3 + ~12 and 2 + ~3 launches per layer, and the step is launch-bound at the reference's batch sizes.
//   edge_cat:    [edge_attr | node[src][:32] | node[dst][:32]]  -- the input of a layer's FCBlock, reference
//                models/score_model.py:319,327,367 (`torch.cat([edge_attr, node_attr[src, :ns], node_attr[dst, :ns]], -1)`)
//   gather_pad:  node_attr[edge_dst] widened to the kernels' 80-float rows (models/tensor_layers.py:203, `node_attr[edge_dst]`)
// Backward passes are fixed-order segmented sums over the edges grouped by node (cbd_csr_build): bitwise repeatable, no atomics.
// HBM-bound row copies; one thread per float4 / float2.
#include <hip/hip_runtime.h>

#include "host_util.h"
#include "../../include/cbdock.h"

namespace cbd {

__global__ __launch_bounds__(256) void edge_cat_kernel(long long E, const float* __restrict__ edge_attr, const float* __restrict__ node,
                                                       int ldn, const long long* __restrict__ src, const long long* __restrict__ dst,
                                                       float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // one float4 of the 24 per edge row
  if (i >= E * 24) return;
  const long long e = i / 24;
  const int q = (int)(i % 24), part = q >> 3, c = (q & 7) * 4;
  float4 v;
  if (part == 0) {
    v = *reinterpret_cast<const float4*>(edge_attr + e * 32 + c);
  } else {
    const float* p = node + (part == 1 ? src[e] : dst[e]) * ldn + c;   // node rows are only 8-byte aligned (74-float rows)
    const float2 a = *reinterpret_cast<const float2*>(p), b = *reinterpret_cast<const float2*>(p + 2);
    v = make_float4(a.x, a.y, b.x, b.y);
  }
  *reinterpret_cast<float4*>(out + e * 96 + q * 4) = v;
}

// g_node[n][c] = sum_{k in src rows of n} g[perm_s[k]][32 + c] + sum_{k in dst rows of n} g[perm_d[k]][64 + c]  (c < 32), 0 for c >= 32.
// One wave per node: lanes 0..31 run the src sum, lanes 32..63 the dst sum, each in index order; the two are added last.
__global__ __launch_bounds__(256) void edge_cat_bwd_kernel(long long N, int D, const float* __restrict__ g,
                                                           const long long* __restrict__ perm_s, const long long* __restrict__ rowptr_s,
                                                           const long long* __restrict__ perm_d, const long long* __restrict__ rowptr_d,
                                                           float* __restrict__ g_node) {
  const long long n = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const bool second = lane >= 32;
  const long long* perm = second ? perm_d : perm_s;
  const long long lo = second ? rowptr_d[n] : rowptr_s[n], hi = second ? rowptr_d[n + 1] : rowptr_s[n + 1];
  const int col = 32 + lane;                                            // 32..63 for the src part, 64..95 for the dst part
  float acc = 0.f;
  long long k = lo;
  for (; k + 4 <= hi; k += 4) {
    const long long p0 = perm[k], p1 = perm[k + 1], p2 = perm[k + 2], p3 = perm[k + 3];
    const float v0 = g[p0 * 96 + col], v1 = g[p1 * 96 + col], v2 = g[p2 * 96 + col], v3 = g[p3 * 96 + col];
    acc = ((acc + v0) + v1) + v2;
    acc += v3;
  }
  for (; k < hi; ++k) acc += g[perm[k] * 96 + col];
  const float other = __shfl_xor(acc, 32);
  if (!second) g_node[n * D + lane] = acc + other;
  for (int c = 32 + lane; c < D; c += 64) g_node[n * D + c] = 0.f;
}

__global__ __launch_bounds__(256) void gather_pad_kernel(long long E, int D, int ldo, const float* __restrict__ node,
                                                         const long long* __restrict__ idx, float* __restrict__ out) {
  const int per_row = ldo / 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // one float2 of a row
  if (i >= E * per_row) return;
  const long long e = i / per_row;
  const int c = (int)(i % per_row) * 2;
  float2 v = make_float2(0.f, 0.f);
  if (c < D) v = *reinterpret_cast<const float2*>(node + idx[e] * D + c);   // D is even for every irreps level (32, 50, 68, 74)
  *reinterpret_cast<float2*>(out + e * ldo + c) = v;
}

// cbd_segment_sum with a row stride: out[n][c] = sum_k vals[perm[k]][c], c < width <= ld
__global__ __launch_bounds__(256) void segment_sum_ld_kernel(long long n_rows, int width, int ld, const float* __restrict__ vals,
                                                             const long long* __restrict__ perm, const long long* __restrict__ rowptr,
                                                             float* __restrict__ out) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const long long lo = rowptr[row], hi = rowptr[row + 1];
  for (int c = lane; c < width; c += 64) {
    float acc = 0.f;
    long long k = lo;
    for (; k + 4 <= hi; k += 4) {
      const long long p0 = perm[k], p1 = perm[k + 1], p2 = perm[k + 2], p3 = perm[k + 3];
      const float v0 = vals[p0 * ld + c], v1 = vals[p1 * ld + c], v2 = vals[p2 * ld + c], v3 = vals[p3 * ld + c];
      acc = ((acc + v0) + v1) + v2;
      acc += v3;
    }
    for (; k < hi; ++k) acc += vals[perm[k] * ld + c];
    out[row * width + c] = acc;
  }
}

// backward of the segmented mean: out[e] = g[index[e]] / max(count(index[e]), 1), count from the row pointers
__global__ __launch_bounds__(256) void gather_mean_bwd_kernel(long long E, int width, const float* __restrict__ g,
                                                              const long long* __restrict__ index, const long long* __restrict__ rowptr,
                                                              float* __restrict__ out) {
  const int per_row = width / 4;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;      // one float4 of a row
  if (i >= E * per_row) return;
  const long long e = i / per_row, n = index[e];
  const int c = (int)(i % per_row) * 4;
  const long long cnt = rowptr[n + 1] - rowptr[n];
  const float sc = 1.0f / (float)(cnt > 1 ? cnt : 1);
  const float4 v = *reinterpret_cast<const float4*>(g + n * width + c);
  *reinterpret_cast<float4*>(out + e * width + c) = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
}

}  // namespace cbd

extern "C" {

int cbd_segment_mean_backward(int64_t n_edges, int32_t width, const float* g_dev, const int64_t* index_dev, const int64_t* rowptr_dev,
                              float* out_dev, void* stream) {
  if (n_edges < 0 || width <= 0 || (width & 3)) return fail(CBD_ERR_ARG, "cbd_segment_mean_backward: bad argument (width must be a multiple of 4)");
  if (n_edges == 0) return 0;
  if (!g_dev || !index_dev || !rowptr_dev || !out_dev) return fail(CBD_ERR_ARG, "cbd_segment_mean_backward: null pointer");
  hipLaunchKernelGGL(cbd::gather_mean_bwd_kernel, dim3((unsigned)((n_edges * (width / 4) + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (long long)n_edges, (int)width, g_dev, reinterpret_cast<const long long*>(index_dev),
                     reinterpret_cast<const long long*>(rowptr_dev), out_dev);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_segment_mean_backward: %s", hipGetErrorString(r));
  return 0;
}

int cbd_edge_cat(int64_t n_edges, const float* edge_attr_dev, const float* node_dev, int32_t node_ld, const int64_t* src_dev,
                 const int64_t* dst_dev, float* out_dev, void* stream) {
  if (n_edges < 0 || node_ld < 32 || (node_ld & 1)) return fail(CBD_ERR_ARG, "cbd_edge_cat: bad argument");
  if (n_edges == 0) return 0;
  if (!edge_attr_dev || !node_dev || !src_dev || !dst_dev || !out_dev) return fail(CBD_ERR_ARG, "cbd_edge_cat: null pointer");
  hipLaunchKernelGGL(cbd::edge_cat_kernel, dim3((unsigned)((n_edges * 24 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (long long)n_edges, edge_attr_dev, node_dev, (int)node_ld, reinterpret_cast<const long long*>(src_dev),
                     reinterpret_cast<const long long*>(dst_dev), out_dev);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_edge_cat: %s", hipGetErrorString(r));
  return 0;
}

int cbd_edge_cat_backward(int64_t n_nodes, int32_t node_dim, const float* g_dev, const int64_t* perm_src_dev, const int64_t* rowptr_src_dev,
                          const int64_t* perm_dst_dev, const int64_t* rowptr_dst_dev, float* g_node_dev, void* stream) {
  if (n_nodes < 0 || node_dim < 32) return fail(CBD_ERR_ARG, "cbd_edge_cat_backward: bad argument");
  if (n_nodes == 0) return 0;
  if (!rowptr_src_dev || !rowptr_dst_dev || !g_node_dev) return fail(CBD_ERR_ARG, "cbd_edge_cat_backward: null pointer");
  hipLaunchKernelGGL(cbd::edge_cat_bwd_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (long long)n_nodes, (int)node_dim, g_dev, reinterpret_cast<const long long*>(perm_src_dev),
                     reinterpret_cast<const long long*>(rowptr_src_dev), reinterpret_cast<const long long*>(perm_dst_dev),
                     reinterpret_cast<const long long*>(rowptr_dst_dev), g_node_dev);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_edge_cat_backward: %s", hipGetErrorString(r));
  return 0;
}

int cbd_gather_pad(int64_t n_edges, int32_t node_dim, int32_t out_ld, const float* node_dev, const int64_t* index_dev, float* out_dev,
                   void* stream) {
  if (n_edges < 0 || node_dim <= 0 || (node_dim & 1) || out_ld < node_dim || (out_ld & 1)) return fail(CBD_ERR_ARG, "cbd_gather_pad: bad argument");
  if (n_edges == 0) return 0;
  if (!node_dev || !index_dev || !out_dev) return fail(CBD_ERR_ARG, "cbd_gather_pad: null pointer");
  hipLaunchKernelGGL(cbd::gather_pad_kernel, dim3((unsigned)((n_edges * (out_ld / 2) + 255) / 256)), dim3(256), 0,
                     reinterpret_cast<hipStream_t>(stream), (long long)n_edges, (int)node_dim, (int)out_ld, node_dev,
                     reinterpret_cast<const long long*>(index_dev), out_dev);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_gather_pad: %s", hipGetErrorString(r));
  return 0;
}

int cbd_segment_sum_ld(int64_t n_rows, int32_t width, int32_t ld, const float* vals_dev, const int64_t* perm_dev, const int64_t* rowptr_dev,
                       float* out_dev, void* stream) {
  if (n_rows < 0 || width <= 0 || ld < width || !rowptr_dev || !out_dev) return fail(CBD_ERR_ARG, "cbd_segment_sum_ld: bad argument");
  if (n_rows == 0) return 0;
  hipLaunchKernelGGL(cbd::segment_sum_ld_kernel, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (long long)n_rows, (int)width, (int)ld, vals_dev, reinterpret_cast<const long long*>(perm_dev),
                     reinterpret_cast<const long long*>(rowptr_dev), out_dev);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_segment_sum_ld: %s", hipGetErrorString(r));
  return 0;
}

}  // extern "C"
