// Denoising score-matching loss of the fine-tuning step (reference utils/training.py:17-126, the `apply_mean=True` form the training
// loop uses) and its gradient with respect to the three predictions, in ONE launch:
//   tr_loss  = mean_{b,c} (tr_pred - tr_score)^2 sigma_tr[b]^2          rot_loss = mean_{b,c} ((rot_pred - rot_score) / rot_norm[b])^2
//   tor_loss = mean_t (tor_pred - tor_score)^2 / tor_norm2[t]           loss = w_tr tr_loss + w_rot rot_loss + w_tor tor_loss
// plus the three "base" losses (the same with pred = 0).  As torch ops this was ~40 element-wise / reduction launches forward and ~60
// backward on tensors of 24 .. 200 floats.  One workgroup; sums in double in a fixed order (bitwise repeatable).
// out[11] = the reference's 11-tuple: loss, tr, rot, tor, 0 (backbone), 0 (side chain), tr_base, rot_base, tor_base, 0, 0.
#include <hip/hip_runtime.h>

#include <cmath>

#include "host_util.h"
#include "../../include/cbdock.h"

namespace cbd {

constexpr int LOSS_THREADS = 256;

__device__ inline double loss_block_sum(double v, double* sh) {
  const int t = threadIdx.x;
  sh[t] = v;
  __syncthreads();
#pragma unroll
  for (int s = LOSS_THREADS / 2; s > 0; s >>= 1) {
    if (t < s) sh[t] += sh[t + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

__global__ __launch_bounds__(LOSS_THREADS) void score_loss_kernel(int B, int T, int has_tor, const float* __restrict__ tr_pred,
                                                                  const float* __restrict__ tr_score, const float* __restrict__ tr_sigma,
                                                                  const float* __restrict__ rot_pred, const float* __restrict__ rot_score,
                                                                  const float* __restrict__ rot_norm, const float* __restrict__ tor_pred,
                                                                  const float* __restrict__ tor_score, const float* __restrict__ tor_norm2,
                                                                  float w_tr, float w_rot, float w_tor, float* __restrict__ out,
                                                                  float* __restrict__ g_tr, float* __restrict__ g_rot, float* __restrict__ g_tor) {
  __shared__ double sh[LOSS_THREADS];
  const int t = threadIdx.x;
  double s_tr = 0, s_trb = 0, s_rot = 0, s_rotb = 0, s_tor = 0, s_torb = 0;
  const float k_tr = 2.f * w_tr / (3.f * (float)B), k_rot = 2.f * w_rot / (3.f * (float)B);
  for (int i = t; i < 3 * B; i += LOSS_THREADS) {
    const int b = i / 3;
    const float sg = tr_sigma[b], s2 = sg * sg;
    const float d = tr_pred[i] - tr_score[i], sc = tr_score[i];
    s_tr += (double)(d * d * s2);
    s_trb += (double)(sc * sc * s2);
    g_tr[i] = k_tr * d * s2;
    const float n = rot_norm[b];
    const float dr = (rot_pred[i] - rot_score[i]) / n, rb = rot_score[i] / n;
    s_rot += (double)(dr * dr);
    s_rotb += (double)(rb * rb);
    g_rot[i] = k_rot * dr / n;
  }
  if (has_tor) {
    const float k_tor = 2.f * w_tor / (float)T;
    for (int i = t; i < T; i += LOSS_THREADS) {
      const float d = tor_pred[i] - tor_score[i], n2 = tor_norm2[i], sc = tor_score[i];
      s_tor += (double)(d * d / n2);
      s_torb += (double)(sc * sc / n2);
      g_tor[i] = k_tor * d / n2;
    }
  }
  const double S_tr = loss_block_sum(s_tr, sh), S_trb = loss_block_sum(s_trb, sh), S_rot = loss_block_sum(s_rot, sh),
               S_rotb = loss_block_sum(s_rotb, sh), S_tor = loss_block_sum(s_tor, sh), S_torb = loss_block_sum(s_torb, sh);
  if (t == 0) {
    const float nan = __builtin_nanf("");
    const float tr = (float)(S_tr / (3.0 * B)), rot = (float)(S_rot / (3.0 * B));
    // the mean of an empty tensor is NaN in torch (a batch without rotatable bonds: the reference's loss is NaN and the step is skipped)
    const float tor = has_tor ? (T > 0 ? (float)(S_tor / T) : nan) : 0.f;
    out[0] = tr * w_tr + rot * w_rot + tor * w_tor;
    out[1] = tr; out[2] = rot; out[3] = tor; out[4] = 0.f; out[5] = 0.f;
    out[6] = (float)(S_trb / (3.0 * B)); out[7] = (float)(S_rotb / (3.0 * B));
    out[8] = has_tor ? (T > 0 ? (float)(S_torb / T) : nan) : 0.f;
    out[9] = 0.f; out[10] = 0.f;
  }
}

}  // namespace cbd

extern "C" int cbd_score_loss(int32_t n_graphs, int32_t n_tor, int32_t has_tor, const float* tr_pred_dev, const float* tr_score_dev,
                              const float* tr_sigma_dev, const float* rot_pred_dev, const float* rot_score_dev, const float* rot_norm_dev,
                              const float* tor_pred_dev, const float* tor_score_dev, const float* tor_norm2_dev, float tr_weight,
                              float rot_weight, float tor_weight, float* out11_dev, float* g_tr_dev, float* g_rot_dev, float* g_tor_dev,
                              void* stream) {
  if (n_graphs <= 0 || n_tor < 0 || !tr_pred_dev || !tr_score_dev || !tr_sigma_dev || !rot_pred_dev || !rot_score_dev || !rot_norm_dev ||
      !out11_dev || !g_tr_dev || !g_rot_dev)
    return fail(CBD_ERR_ARG, "cbd_score_loss: bad argument");
  if (has_tor && n_tor > 0 && (!tor_pred_dev || !tor_score_dev || !tor_norm2_dev || !g_tor_dev))
    return fail(CBD_ERR_ARG, "cbd_score_loss: torsion tensors missing");
  hipLaunchKernelGGL(cbd::score_loss_kernel, dim3(1), dim3(cbd::LOSS_THREADS), 0, reinterpret_cast<hipStream_t>(stream), (int)n_graphs, (int)n_tor,
                     (int)has_tor, tr_pred_dev, tr_score_dev, tr_sigma_dev, rot_pred_dev, rot_score_dev, rot_norm_dev, tor_pred_dev,
                     tor_score_dev, tor_norm2_dev, tr_weight, rot_weight, tor_weight, out11_dev, g_tr_dev, g_rot_dev, g_tor_dev);
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_score_loss: %s", hipGetErrorString(r));
  return 0;
}
