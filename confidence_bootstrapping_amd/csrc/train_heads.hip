// The two e3nn heads of the score model in the fine-tuning step, forward and backward as ONE launch each:
//   * centre convolution (models/score_model.py:245-255, 393-404: final_conv.tp = o3.FullyConnectedTensorProduct(74-irreps, '1x0e+1x1o',
//     '2x1o+2x1e') with per-edge weights), closed forms of its six instructions
//         out1o[w] = ( sum_u wa[u,w] x0e[u] v + sum_u wb[u,w] x1o[u] + (1/sqrt2) sum_u we[u,w] (x1e[u] x v) ) / sqrt(44)
//         out1e[w] = ( (1/sqrt2) sum_u wc[u,w] (x1o[u] x v) + sum_u wd[u,w] x1e[u] + sum_u wf[u,w] x0o[u] v ) / sqrt(18),   v = sqrt3 * unit(vec)
//     weights instruction-major [0e*1o->1o (32x2) | 1o*0e->1o (6x2) | 1o*1o->1e (6x2) | 1e*0e->1e (6x2) | 1e*1o->1o (6x2) | 0o*1o->1e (6x2)];
//   * torsion head (models/score_model.py:257-274, 431-441: final_tp_tor = o3.FullTensorProduct('1x0e+1x1o', '2e') followed by
//     tor_bond_conv.tp with its two live paths):  T1 = (3/sqrt2)(b b^T - I/3) v,
//         out0e[c] = k sum_u w[u,c] (x1o[u] . T1),   out0o[c] = k sum_u w[192 + u*32 + c] (x1e[u] . T1),   k = 1/(sqrt6 sqrt3), rows [0o | 0e].
// As torch ops (train_forward.py, round 3) these were ~55 launches forward and ~100 backward per step on a few hundred rows.  The edge
// and bond directions come from the poses and carry no gradient.  Fixed-order sums: bitwise repeatable.
#include <hip/hip_runtime.h>

#include "host_util.h"
#include "../../include/cbdock.h"

namespace cbd {

constexpr int HX_1O = 32, HX_1E = 50, HX_0O = 68, HX_DIM = 74;
constexpr int CW = 124, CO = 12;         // centre head: weights and outputs per row
constexpr int BW = 384, BO = 64;         // torsion head

__device__ inline void unit3(const float* p, float s, float (&v)[3]) {
  const float n = sqrtf(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
  const float k = s / fmaxf(n, 1e-12f);                  // F.normalize: x / max(|x|, eps)
  v[0] = p[0] * k; v[1] = p[1] * k; v[2] = p[2] * k;
}
__device__ inline void cross(const float (&a)[3], const float (&b)[3], float (&c)[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// one thread per row
__global__ __launch_bounds__(64) void center_tp_fwd_kernel(long long n, const float* __restrict__ x, int ldx, const float* __restrict__ vec,
                                                           const float* __restrict__ w, float* __restrict__ out) {
  const long long r = (long long)blockIdx.x * 64 + threadIdx.x;
  if (r >= n) return;
  const float* xr = x + r * ldx;
  const float* wr = w + r * CW;
  float v[3];
  unit3(vec + 3 * r, 1.7320508075688772f, v);
  float o1o[2][3] = {{0, 0, 0}, {0, 0, 0}}, o1e[2][3] = {{0, 0, 0}, {0, 0, 0}};
  float sa[2] = {0, 0}, sf[2] = {0, 0};
  for (int u = 0; u < 32; ++u) { sa[0] = fmaf(wr[2 * u], xr[u], sa[0]); sa[1] = fmaf(wr[2 * u + 1], xr[u], sa[1]); }
  const float s2 = 0.70710678118654752f;
  for (int u = 0; u < 6; ++u) {
    const float a[3] = {xr[HX_1O + 3 * u], xr[HX_1O + 3 * u + 1], xr[HX_1O + 3 * u + 2]};
    const float e[3] = {xr[HX_1E + 3 * u], xr[HX_1E + 3 * u + 1], xr[HX_1E + 3 * u + 2]};
    float ca[3], ce[3];
    cross(a, v, ca);
    cross(e, v, ce);
    const float so = xr[HX_0O + u];
    for (int q = 0; q < 2; ++q) {
      const float wb = wr[64 + 2 * u + q], wc = wr[76 + 2 * u + q], wd = wr[88 + 2 * u + q], we = wr[100 + 2 * u + q], wf = wr[112 + 2 * u + q];
      for (int k = 0; k < 3; ++k) {
        o1o[q][k] += wb * a[k] + s2 * we * ce[k];
        o1e[q][k] += s2 * wc * ca[k] + wd * e[k];
      }
      sf[q] = fmaf(wf, so, sf[q]);
    }
  }
  const float k44 = 0.15075567228888181f, k18 = 0.23570226039551584f;     // 1/sqrt(44), 1/sqrt(18)
  float* o = out + r * CO;
  for (int q = 0; q < 2; ++q)
    for (int k = 0; k < 3; ++k) {
      o[3 * q + k] = (o1o[q][k] + sa[q] * v[k]) * k44;
      o[6 + 3 * q + k] = (o1e[q][k] + sf[q] * v[k]) * k18;
    }
}

__global__ __launch_bounds__(64) void center_tp_bwd_kernel(long long n, const float* __restrict__ x, int ldx, const float* __restrict__ vec,
                                                           const float* __restrict__ w, const float* __restrict__ gout, float* __restrict__ gx,
                                                           float* __restrict__ gw) {
  const long long r = (long long)blockIdx.x * 64 + threadIdx.x;
  if (r >= n) return;
  const float* xr = x + r * ldx;
  const float* wr = w + r * CW;
  float* gxr = gx + r * ldx;
  float* gwr = gw + r * CW;
  float v[3];
  unit3(vec + 3 * r, 1.7320508075688772f, v);
  const float k44 = 0.15075567228888181f, k18 = 0.23570226039551584f, s2 = 0.70710678118654752f;
  float g1o[2][3], g1e[2][3], vg1o[2], vg1e[2], xo[2][3], xe[2][3];       // scaled output gradients, v . g, v x g
  for (int q = 0; q < 2; ++q) {
    for (int k = 0; k < 3; ++k) { g1o[q][k] = gout[r * CO + 3 * q + k] * k44; g1e[q][k] = gout[r * CO + 6 + 3 * q + k] * k18; }
    vg1o[q] = v[0] * g1o[q][0] + v[1] * g1o[q][1] + v[2] * g1o[q][2];
    vg1e[q] = v[0] * g1e[q][0] + v[1] * g1e[q][1] + v[2] * g1e[q][2];
    cross(v, g1o[q], xo[q]);            // d/da of g . (a x v) = v x g
    cross(v, g1e[q], xe[q]);
  }
  for (int u = 0; u < 32; ++u) {
    gwr[2 * u] = xr[u] * vg1o[0];
    gwr[2 * u + 1] = xr[u] * vg1o[1];
    gxr[u] = wr[2 * u] * vg1o[0] + wr[2 * u + 1] * vg1o[1];
  }
  for (int u = 0; u < 6; ++u) {
    const float a[3] = {xr[HX_1O + 3 * u], xr[HX_1O + 3 * u + 1], xr[HX_1O + 3 * u + 2]};
    const float e[3] = {xr[HX_1E + 3 * u], xr[HX_1E + 3 * u + 1], xr[HX_1E + 3 * u + 2]};
    float ca[3], ce[3];
    cross(a, v, ca);
    cross(e, v, ce);
    const float so = xr[HX_0O + u];
    float ga[3] = {0, 0, 0}, ge[3] = {0, 0, 0}, gso = 0.f;
    for (int q = 0; q < 2; ++q) {
      const float wb = wr[64 + 2 * u + q], wc = wr[76 + 2 * u + q], wd = wr[88 + 2 * u + q], we = wr[100 + 2 * u + q], wf = wr[112 + 2 * u + q];
      gwr[64 + 2 * u + q] = a[0] * g1o[q][0] + a[1] * g1o[q][1] + a[2] * g1o[q][2];
      gwr[76 + 2 * u + q] = s2 * (ca[0] * g1e[q][0] + ca[1] * g1e[q][1] + ca[2] * g1e[q][2]);
      gwr[88 + 2 * u + q] = e[0] * g1e[q][0] + e[1] * g1e[q][1] + e[2] * g1e[q][2];
      gwr[100 + 2 * u + q] = s2 * (ce[0] * g1o[q][0] + ce[1] * g1o[q][1] + ce[2] * g1o[q][2]);
      gwr[112 + 2 * u + q] = so * vg1e[q];
      for (int k = 0; k < 3; ++k) {
        ga[k] += wb * g1o[q][k] + s2 * wc * xe[q][k];
        ge[k] += wd * g1e[q][k] + s2 * we * xo[q][k];
      }
      gso = fmaf(wf, vg1e[q], gso);
    }
    for (int k = 0; k < 3; ++k) { gxr[HX_1O + 3 * u + k] = ga[k]; gxr[HX_1E + 3 * u + k] = ge[k]; }
    gxr[HX_0O + u] = gso;
  }
  for (int c = HX_DIM; c < ldx; ++c) gxr[c] = 0.f;
}

// torsion head: 32 lanes per row (lane = output channel), two rows per wave
__device__ inline float row_sum32(float v) {
#pragma unroll
  for (int d = 16; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}
__device__ inline void bond_t1(const float* ev, const float* bv, float (&t1)[3]) {
  float v[3], b[3];
  unit3(ev, 1.7320508075688772f, v);
  unit3(bv, 1.f, b);
  const float bvd = b[0] * v[0] + b[1] * v[1] + b[2] * v[2];
  const float k = 2.1213203435596424f;     // 3 / sqrt(2)
  for (int c = 0; c < 3; ++c) t1[c] = k * (b[c] * bvd - v[c] / 3.0f);
}

__global__ __launch_bounds__(64) void bond_tp_fwd_kernel(long long n, const float* __restrict__ x, int ldx, const float* __restrict__ evec,
                                                         const float* __restrict__ bvec, const float* __restrict__ w, float* __restrict__ out) {
  const long long r = (long long)blockIdx.x * 2 + (threadIdx.x >> 5);
  const int c = threadIdx.x & 31;
  if (r >= n) return;
  const float* xr = x + r * ldx;
  const float* wr = w + r * BW;
  float t1[3];
  bond_t1(evec + 3 * r, bvec + 3 * r, t1);
  float oe = 0.f, oo = 0.f;
  for (int u = 0; u < 6; ++u) {
    const float so = xr[HX_1O + 3 * u] * t1[0] + xr[HX_1O + 3 * u + 1] * t1[1] + xr[HX_1O + 3 * u + 2] * t1[2];
    const float se = xr[HX_1E + 3 * u] * t1[0] + xr[HX_1E + 3 * u + 1] * t1[1] + xr[HX_1E + 3 * u + 2] * t1[2];
    oe = fmaf(wr[32 * u + c], so, oe);
    oo = fmaf(wr[192 + 32 * u + c], se, oo);
  }
  const float k = 0.23570226039551584f;     // 1 / (sqrt6 sqrt3) = 1 / sqrt(18)
  out[r * BO + c] = k * oo;                 // [0o | 0e]
  out[r * BO + 32 + c] = k * oe;
}

__global__ __launch_bounds__(64) void bond_tp_bwd_kernel(long long n, const float* __restrict__ x, int ldx, const float* __restrict__ evec,
                                                         const float* __restrict__ bvec, const float* __restrict__ w, const float* __restrict__ gout,
                                                         float* __restrict__ gx, float* __restrict__ gw) {
  const long long r0 = (long long)blockIdx.x * 2 + (threadIdx.x >> 5);
  const int c = threadIdx.x & 31;
  const bool live = r0 < n;
  const long long r = live ? r0 : n - 1;          // both halves of the wave take part in the shuffles
  const float* xr = x + r * ldx;
  const float* wr = w + r * BW;
  float t1[3];
  bond_t1(evec + 3 * r, bvec + 3 * r, t1);
  const float k = 0.23570226039551584f;
  const float go = k * gout[r * BO + c], ge = k * gout[r * BO + 32 + c];
  float gso[6], gse[6];
  for (int u = 0; u < 6; ++u) {
    const float so = xr[HX_1O + 3 * u] * t1[0] + xr[HX_1O + 3 * u + 1] * t1[1] + xr[HX_1O + 3 * u + 2] * t1[2];
    const float se = xr[HX_1E + 3 * u] * t1[0] + xr[HX_1E + 3 * u + 1] * t1[1] + xr[HX_1E + 3 * u + 2] * t1[2];
    if (live) { gw[r * BW + 32 * u + c] = so * ge; gw[r * BW + 192 + 32 * u + c] = se * go; }
    gso[u] = row_sum32(wr[32 * u + c] * ge);
    gse[u] = row_sum32(wr[192 + 32 * u + c] * go);
  }
  if (!live) return;
  float* gxr = gx + r * ldx;
  for (int col = c; col < ldx; col += 32) {
    float g = 0.f;
    if (col >= HX_1O && col < HX_1E) g = gso[(col - HX_1O) / 3] * t1[(col - HX_1O) % 3];
    else if (col >= HX_1E && col < HX_0O) g = gse[(col - HX_1E) / 3] * t1[(col - HX_1E) % 3];
    gxr[col] = g;
  }
}

}  // namespace cbd

extern "C" {

int cbd_center_tp_forward(int64_t n, const float* x_dev, int32_t ldx, const float* vec_dev, const float* w_dev, float* out_dev, void* stream) {
  if (n < 0 || ldx < cbd::HX_DIM || (n > 0 && (!x_dev || !vec_dev || !w_dev || !out_dev))) return fail(CBD_ERR_ARG, "cbd_center_tp_forward: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cbd::center_tp_fwd_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (long long)n, x_dev,
                     (int)ldx, vec_dev, w_dev, out_dev);
  const hipError_t r = hipGetLastError();
  return r == hipSuccess ? 0 : fail(CBD_ERR_HIP, "cbd_center_tp_forward: %s", hipGetErrorString(r));
}

int cbd_center_tp_backward(int64_t n, const float* x_dev, int32_t ldx, const float* vec_dev, const float* w_dev, const float* gout_dev,
                           float* gx_dev, float* gw_dev, void* stream) {
  if (n < 0 || ldx < cbd::HX_DIM || (n > 0 && (!x_dev || !vec_dev || !w_dev || !gout_dev || !gx_dev || !gw_dev)))
    return fail(CBD_ERR_ARG, "cbd_center_tp_backward: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cbd::center_tp_bwd_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (long long)n, x_dev,
                     (int)ldx, vec_dev, w_dev, gout_dev, gx_dev, gw_dev);
  const hipError_t r = hipGetLastError();
  return r == hipSuccess ? 0 : fail(CBD_ERR_HIP, "cbd_center_tp_backward: %s", hipGetErrorString(r));
}

int cbd_bond_tp_forward(int64_t n, const float* x_dev, int32_t ldx, const float* edge_vec_dev, const float* bond_vec_dev, const float* w_dev,
                        float* out_dev, void* stream) {
  if (n < 0 || ldx < cbd::HX_DIM || (n > 0 && (!x_dev || !edge_vec_dev || !bond_vec_dev || !w_dev || !out_dev)))
    return fail(CBD_ERR_ARG, "cbd_bond_tp_forward: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cbd::bond_tp_fwd_kernel, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (long long)n, x_dev,
                     (int)ldx, edge_vec_dev, bond_vec_dev, w_dev, out_dev);
  const hipError_t r = hipGetLastError();
  return r == hipSuccess ? 0 : fail(CBD_ERR_HIP, "cbd_bond_tp_forward: %s", hipGetErrorString(r));
}

int cbd_bond_tp_backward(int64_t n, const float* x_dev, int32_t ldx, const float* edge_vec_dev, const float* bond_vec_dev, const float* w_dev,
                         const float* gout_dev, float* gx_dev, float* gw_dev, void* stream) {
  if (n < 0 || ldx < cbd::HX_DIM || (n > 0 && (!x_dev || !edge_vec_dev || !bond_vec_dev || !w_dev || !gout_dev || !gx_dev || !gw_dev)))
    return fail(CBD_ERR_ARG, "cbd_bond_tp_backward: bad argument");
  if (n == 0) return 0;
  hipLaunchKernelGGL(cbd::bond_tp_bwd_kernel, dim3((unsigned)((n + 1) / 2)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), (long long)n, x_dev,
                     (int)ldx, edge_vec_dev, bond_vec_dev, w_dev, gout_dev, gx_dev, gw_dev);
  const hipError_t r = hipGetLastError();
  return r == hipSuccess ? 0 : fail(CBD_ERR_HIP, "cbd_bond_tp_backward: %s", hipGetErrorString(r));
}

}  // extern "C"
