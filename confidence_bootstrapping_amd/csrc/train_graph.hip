// Radius graphs of the fine-tuning step: torch_cluster.radius / radius_graph as the reference's training forward calls
// them (models/score_model.py:498-503 ligand graph, :573-580 cross graph with the per-graph cutoff, :652-656 torsion graph) for BATCHED
// point sets -- for every query y the points x of the same graph with |x - y|^2 < r^2, the first `cap` in index order.  Two passes
// (count, fill) around one exclusive scan and one count read-back for all graphs of the step; the torch-op form builds four dense
// [Ny, Nx] tensors per graph (~15 launches each).  Latency-bound integer work: one wave per query, candidates through the lanes 64 at
// a time, ranks from ballots.  The distance arithmetic repeats the torch ops' roundings exactly (separate multiply and add, IEEE
// division), so the edge sets are identical to the mask formulation (train_forward.radius_mask) bit for bit.
#include <hip/hip_runtime.h>

#include "host_util.h"
#include "../../include/cbdock.h"

namespace cbd {

template <bool FILL>
__global__ __launch_bounds__(256) void radius_batched_kernel(long long ny, const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ cut, float r2, const long long* __restrict__ xptr,
                                                             const long long* __restrict__ ybatch, long long cap, int drop_self,
                                                             long long* __restrict__ counts, const long long* __restrict__ offsets,
                                                             long long* __restrict__ out_y, long long* __restrict__ out_x) {
  const long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (q >= ny) return;
  const long long b = ybatch[q], lo = xptr[b], hi = xptr[b + 1];
  const float c = cut ? cut[b] : 1.f;
  float yq[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) yq[k] = cut ? __fdiv_rn(y[q * 3 + k], c) : y[q * 3 + k];
  long long seen = 0, kept = 0;                       // in-radius points so far (self included), edges written so far (self excluded)
  const long long base = FILL ? offsets[q] : 0;
  for (long long j0 = lo; j0 < hi && seen < cap; j0 += 64) {
    const long long j = j0 + lane;
    bool in = false;
    if (j < hi) {
      float d2 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float xv = cut ? __fdiv_rn(x[j * 3 + k], c) : x[j * 3 + k];
        const float diff = __fsub_rn(xv, yq[k]);
        d2 = __fadd_rn(d2, __fmul_rn(diff, diff));
      }
      in = d2 < r2;
    }
    const unsigned long long m = __ballot(in);
    const unsigned long long below = m & ((1ull << lane) - 1ull);
    const long long rank = seen + __popcll(below) + 1;             // 1-based rank among the in-radius points, index order
    const bool keep = in && rank <= cap && !(drop_self && j == q);
    const unsigned long long km = __ballot(keep);
    if (FILL && keep) {
      const long long pos = base + kept + __popcll(km & ((1ull << lane) - 1ull));
      out_y[pos] = q;
      out_x[pos] = j;
    }
    seen += __popcll(m);
    kept += __popcll(km);
  }
  if (!FILL && lane == 0) counts[q] = kept;
}

// Edge geometry of the training forward in one launch: vec = pos_b[idx_b] - pos_a[idx_a] (idx == NULL: identity), its unit vector (the
// kernels' `vec` operand, F.normalize semantics: v / max(|v|, 1e-12)) and the Gaussian distance expansion exp(coeff (|v| - mu_k)^2)
// (models/score_model.py:667-677) -- as torch ops two gathers, a subtraction, two norms, a clamp, a division, a pad and the four ops
// of the expansion: 13 launches per edge set, five edge sets per step.
__global__ __launch_bounds__(256) void edge_geometry_kernel(long long E, const float* __restrict__ pos_a, const float* __restrict__ pos_b,
                                                            const long long* __restrict__ idx_a, const long long* __restrict__ idx_b,
                                                            int K, const float* __restrict__ mu, float coeff, float* __restrict__ raw4,
                                                            float* __restrict__ unit4, float* __restrict__ smear) {
  const long long e = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (e >= E) return;
  const long long a = idx_a ? idx_a[e] : e, b = idx_b ? idx_b[e] : e;
  const float x = pos_b[b * 3 + 0] - pos_a[a * 3 + 0], y = pos_b[b * 3 + 1] - pos_a[a * 3 + 1], z = pos_b[b * 3 + 2] - pos_a[a * 3 + 2];
  const float d = sqrtf(x * x + y * y + z * z);
  if (lane == 0) {
    if (raw4) *reinterpret_cast<float4*>(raw4 + e * 4) = make_float4(x, y, z, 0.f);
    if (unit4) {
      const float inv = 1.0f / fmaxf(d, 1e-12f);
      *reinterpret_cast<float4*>(unit4 + e * 4) = make_float4(x * inv, y * inv, z * inv, 0.f);
    }
  }
  if (smear)
    for (int k = lane; k < K; k += 64) {
      const float t = d - mu[k];
      smear[e * K + k] = expf(coeff * (t * t));
    }
}

}  // namespace cbd

extern "C" {

int cbd_edge_geometry(int64_t n_edges, const float* pos_a_dev, const float* pos_b_dev, const int64_t* idx_a_dev, const int64_t* idx_b_dev,
                      int32_t n_mu, const float* mu_dev, float coeff, float* raw4_dev, float* unit4_dev, float* smear_dev, void* stream) {
  if (n_edges < 0 || (smear_dev && (n_mu <= 0 || !mu_dev))) return fail(CBD_ERR_ARG, "cbd_edge_geometry: bad argument");
  if (n_edges == 0) return 0;
  if (!pos_a_dev || !pos_b_dev) return fail(CBD_ERR_ARG, "cbd_edge_geometry: null pointer");
  hipLaunchKernelGGL(cbd::edge_geometry_kernel, dim3((unsigned)((n_edges + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (long long)n_edges, pos_a_dev, pos_b_dev, reinterpret_cast<const long long*>(idx_a_dev),
                     reinterpret_cast<const long long*>(idx_b_dev), (int)n_mu, mu_dev, coeff, raw4_dev, unit4_dev, smear_dev);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_edge_geometry: %s", hipGetErrorString(r));
  return 0;
}

static int radius_args_ok(int64_t ny, const float* x, const float* y, const int64_t* xptr, const int64_t* ybatch, int64_t cap) {
  return ny >= 0 && cap > 0 && (ny == 0 || (x && y && xptr && ybatch));
}

int cbd_radius_count(int64_t n_query, const float* x_dev, const float* y_dev, const float* cutoff_dev, float r2, const int64_t* xptr_dev,
                     const int64_t* ybatch_dev, int64_t cap, int32_t drop_self, int64_t* counts_dev, void* stream) {
  if (!radius_args_ok(n_query, x_dev, y_dev, xptr_dev, ybatch_dev, cap) || (n_query > 0 && !counts_dev))
    return fail(CBD_ERR_ARG, "cbd_radius_count: bad argument");
  if (n_query == 0) return 0;
  hipLaunchKernelGGL(cbd::radius_batched_kernel<false>, dim3((unsigned)((n_query + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (long long)n_query, x_dev, y_dev, cutoff_dev, r2, reinterpret_cast<const long long*>(xptr_dev),
                     reinterpret_cast<const long long*>(ybatch_dev), (long long)cap, (int)drop_self, reinterpret_cast<long long*>(counts_dev),
                     (const long long*)nullptr, (long long*)nullptr, (long long*)nullptr);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_radius_count: %s", hipGetErrorString(r));
  return 0;
}

int cbd_radius_fill(int64_t n_query, const float* x_dev, const float* y_dev, const float* cutoff_dev, float r2, const int64_t* xptr_dev,
                    const int64_t* ybatch_dev, int64_t cap, int32_t drop_self, const int64_t* offsets_dev, int64_t* out_query_dev,
                    int64_t* out_point_dev, void* stream) {
  if (!radius_args_ok(n_query, x_dev, y_dev, xptr_dev, ybatch_dev, cap) || (n_query > 0 && (!offsets_dev || !out_query_dev || !out_point_dev)))
    return fail(CBD_ERR_ARG, "cbd_radius_fill: bad argument");
  if (n_query == 0) return 0;
  hipLaunchKernelGGL(cbd::radius_batched_kernel<true>, dim3((unsigned)((n_query + 3) / 4)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     (long long)n_query, x_dev, y_dev, cutoff_dev, r2, reinterpret_cast<const long long*>(xptr_dev),
                     reinterpret_cast<const long long*>(ybatch_dev), (long long)cap, (int)drop_self, (long long*)nullptr,
                     reinterpret_cast<const long long*>(offsets_dev), reinterpret_cast<long long*>(out_query_dev),
                     reinterpret_cast<long long*>(out_point_dev));
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_radius_fill: %s", hipGetErrorString(r));
  return 0;
}

}  // extern "C"
