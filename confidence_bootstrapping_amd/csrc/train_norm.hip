// Train-mode e3nn BatchNorm of the fine-tuning step: forward and backward as ONE launch each.  Reference:
// e3nn.nn.BatchNorm (0.5.0; affine, normalization='component', reduce='mean') as built by TensorProductConvLayer
// (models/tensor_layers.py:191-193) and applied in its forward (:208-209) under model.train() (utils/training.py:186).
//   0e fields:    y = (x - mean_n x) * w / sqrt(mean_n (x - mean)^2 + eps) + b
//   other fields: y = x * w / sqrt(mean_{n,k} x^2 + eps)                       (k = the 2l+1 components of the field)
// with the running statistics updated by `momentum`.  The layer's residual  out + pad(node_attr)  (tensor_layers.py:211-213) rides
// along.  As torch ops this was ~14 launches forward and ~25 backward per layer, and the step is host-bound at the reference's batch
// sizes.  Latency-bound: one workgroup of 1024 threads per field, partial sums in double in a fixed order (bitwise repeatable).
// Two row ranges [ex0_lo, ex0_hi), [ex1_lo, ex1_hi) can be EXCLUDED from the statistics (capacity padding of the hipGraph-captured
// training step, train_graph.py: the filler graph's ligand and receptor nodes): excluded rows count for nothing in the statistics,
// their output is ZERO (residual included) and so is their input gradient.
#include <hip/hip_runtime.h>

#include "host_util.h"
#include "../../include/cbdock.h"

namespace cbd {

constexpr int BN_THREADS = 1024;     // one workgroup per field: 1024 threads keep enough strided loads in flight for 10^4 rows

struct BnExclude { long long lo0, hi0, lo1, hi1; };
__device__ inline bool bn_live(long long n, const BnExclude& ex) { return !((n >= ex.lo0 && n < ex.hi0) || (n >= ex.lo1 && n < ex.hi1)); }
// each range inside [0, n], disjoint when both are non-empty, at least one live row
inline bool bn_ranges_ok(const BnExclude& ex, long long n) {
  if (ex.lo0 < 0 || ex.hi0 < ex.lo0 || ex.hi0 > n || ex.lo1 < 0 || ex.hi1 < ex.lo1 || ex.hi1 > n) return false;
  if (ex.hi0 > ex.lo0 && ex.hi1 > ex.lo1 && !(ex.hi0 <= ex.lo1 || ex.hi1 <= ex.lo0)) return false;
  return (ex.hi0 - ex.lo0) + (ex.hi1 - ex.lo1) < n || n <= 0;
}

// fixed-order block sum of one double per thread (BN_THREADS threads)
__device__ inline double block_sum(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
#pragma unroll
  for (int s = BN_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) sh[t] += sh[t + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// chan[3c] = first column, chan[3c+1] = components, chan[3c+2] = index among the 0e fields or -1
__global__ __launch_bounds__(BN_THREADS) void irreps_bn_fwd_kernel(long long N, int D, int ldx, const int* __restrict__ chan,
                                                            const float* __restrict__ x, const float* __restrict__ res, int res_dim, const float* __restrict__ weight,
                                                            const float* __restrict__ bias, float* __restrict__ running_mean,
                                                            float* __restrict__ running_var, float momentum, float eps,
                                                            float* __restrict__ out, float* __restrict__ save_mean,
                                                            float* __restrict__ save_inv, BnExclude ex) {
  __shared__ double sh[BN_THREADS];
  const int c = blockIdx.x, col = chan[3 * c], d = chan[3 * c + 1], i0 = chan[3 * c + 2];
  const long long cnt = N * d;
  const long long n_live = N - (ex.hi0 - ex.lo0) - (ex.hi1 - ex.lo1), cnt_live = n_live * d;
  float mean = 0.f;
  if (i0 >= 0) {
    double s = 0.0;
    for (long long n = threadIdx.x; n < N; n += BN_THREADS)
      if (bn_live(n, ex)) s += (double)x[n * ldx + col];
    mean = (float)(block_sum(s, sh) / (double)n_live);
  }
  double s2 = 0.0;
  for (long long i = threadIdx.x; i < cnt; i += BN_THREADS) {
    if (!bn_live(i / d, ex)) continue;
    const float v = x[(i / d) * ldx + col + (int)(i % d)] - mean;
    s2 += (double)v * (double)v;
  }
  const float var = (float)(block_sum(s2, sh) / (double)cnt_live);
  const float inv = 1.0f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    save_mean[c] = mean;
    save_inv[c] = inv;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * var;
    if (i0 >= 0) running_mean[i0] = (1.f - momentum) * running_mean[i0] + momentum * mean;
  }
  const float w = weight[c] * inv, b = i0 >= 0 ? bias[i0] : 0.f;
  for (long long i = threadIdx.x; i < cnt; i += BN_THREADS) {
    const long long n = i / d;
    const int cc = col + (int)(i % d);
    float y = (x[n * ldx + cc] - mean) * w + b;
    if (res && cc < res_dim) y += res[n * res_dim + cc];
    out[n * D + cc] = bn_live(n, ex) ? y : 0.f;     // excluded rows stay zero layer after layer (nothing normalises them: they would grow)
  }
}

__global__ __launch_bounds__(BN_THREADS) void irreps_bn_bwd_kernel(long long N, int D, int ldx, const int* __restrict__ chan,
                                                            const float* __restrict__ g, const float* __restrict__ x, const float* __restrict__ weight,
                                                            const float* __restrict__ save_mean, const float* __restrict__ save_inv,
                                                            float* __restrict__ gx, float* __restrict__ gw, float* __restrict__ gb, BnExclude ex) {
  __shared__ double sh[BN_THREADS];
  const int c = blockIdx.x, col = chan[3 * c], d = chan[3 * c + 1], i0 = chan[3 * c + 2];
  const long long cnt = N * d;
  const long long n_live = N - (ex.hi0 - ex.lo0) - (ex.hi1 - ex.lo1), cnt_live = n_live * d;
  const float mean = save_mean[c], inv = save_inv[c];
  double s1 = 0.0, s0 = 0.0;
  for (long long i = threadIdx.x; i < cnt; i += BN_THREADS) {
    const long long n = i / d;
    if (!bn_live(n, ex)) continue;
    const int cc = col + (int)(i % d);
    const float gv = g[n * D + cc];
    s1 += (double)gv * (double)((x[n * ldx + cc] - mean) * inv);
    s0 += (double)gv;
  }
  const double S1 = block_sum(s1, sh);
  const double S0 = i0 >= 0 ? block_sum(s0, sh) : 0.0;
  if (threadIdx.x == 0) {
    gw[c] = (float)S1;
    if (i0 >= 0) gb[i0] = (float)S0;
  }
  const float m1 = (float)(S1 / (double)cnt_live), m0 = (float)(S0 / (double)n_live), wi = weight[c] * inv;
  for (long long i = threadIdx.x; i < cnt; i += BN_THREADS) {
    const long long n = i / d;
    const int cc = col + (int)(i % d);
    gx[n * ldx + cc] = bn_live(n, ex) ? wi * (g[n * D + cc] - (x[n * ldx + cc] - mean) * inv * m1 - m0) : 0.f;
  }
  if (c == 0 && ldx > D)      // the padding columns of x carry no gradient
    for (long long i = threadIdx.x; i < N * (ldx - D); i += BN_THREADS) gx[(i / (ldx - D)) * ldx + D + (int)(i % (ldx - D))] = 0.f;
}

}  // namespace cbd

extern "C" {

int cbd_irreps_bn_forward(int64_t n, int32_t dim, int32_t ldx, int32_t n_fields, const int32_t* fields_dev, const float* x_dev, const float* res_dev,
                          int32_t res_dim, const float* weight_dev, const float* bias_dev, float* running_mean_dev, float* running_var_dev,
                          float momentum, float eps, float* out_dev, float* save_mean_dev, float* save_inv_dev, const int64_t* exclude4,
                          void* stream) {
  cbd::BnExclude ex = {0, 0, 0, 0};
  if (exclude4) ex = {exclude4[0], exclude4[1], exclude4[2], exclude4[3]};
  if (!cbd::bn_ranges_ok(ex, n)) return fail(CBD_ERR_ARG, "cbd_irreps_bn_forward: bad exclusion ranges");
  if (n <= 0 || dim <= 0 || ldx < dim || n_fields <= 0 || !fields_dev || !x_dev || !weight_dev || !running_var_dev || !out_dev || !save_mean_dev ||
      !save_inv_dev || (res_dev && (res_dim <= 0 || res_dim > dim)))
    return fail(CBD_ERR_ARG, "cbd_irreps_bn_forward: bad argument");
  hipLaunchKernelGGL(cbd::irreps_bn_fwd_kernel, dim3((unsigned)n_fields), dim3(cbd::BN_THREADS), 0, reinterpret_cast<hipStream_t>(stream), (long long)n,
                     (int)dim, (int)ldx, fields_dev, x_dev, res_dev, (int)res_dim, weight_dev, bias_dev, running_mean_dev, running_var_dev, momentum,
                     eps, out_dev, save_mean_dev, save_inv_dev, ex);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_irreps_bn_forward: %s", hipGetErrorString(r));
  return 0;
}

int cbd_irreps_bn_backward(int64_t n, int32_t dim, int32_t ldx, int32_t n_fields, const int32_t* fields_dev, const float* g_dev, const float* x_dev,
                           const float* weight_dev, const float* save_mean_dev, const float* save_inv_dev, float* gx_dev, float* gw_dev,
                           float* gb_dev, const int64_t* exclude4, void* stream) {
  cbd::BnExclude ex = {0, 0, 0, 0};
  if (exclude4) ex = {exclude4[0], exclude4[1], exclude4[2], exclude4[3]};
  if (!cbd::bn_ranges_ok(ex, n)) return fail(CBD_ERR_ARG, "cbd_irreps_bn_backward: bad exclusion ranges");
  if (n <= 0 || dim <= 0 || ldx < dim || n_fields <= 0 || !fields_dev || !g_dev || !x_dev || !weight_dev || !save_mean_dev || !save_inv_dev || !gx_dev ||
      !gw_dev)
    return fail(CBD_ERR_ARG, "cbd_irreps_bn_backward: bad argument");
  hipLaunchKernelGGL(cbd::irreps_bn_bwd_kernel, dim3((unsigned)n_fields), dim3(cbd::BN_THREADS), 0, reinterpret_cast<hipStream_t>(stream), (long long)n,
                     (int)dim, (int)ldx, fields_dev, g_dev, x_dev, weight_dev, save_mean_dev, save_inv_dev, gx_dev, gw_dev, gb_dev, ex);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_irreps_bn_backward: %s", hipGetErrorString(r));
  return 0;
}

}  // extern "C"
