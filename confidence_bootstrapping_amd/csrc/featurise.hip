// Neighbour graphs of the receptor featurisation: the C-alpha kNN-24 graph, the heavy-atom kNN-8 graph and the
// cutoff graphs of the non-kNN branch -- reference datasets/process_mols.py:456-479,491-513 (`knn_graph` of torch_cluster /
// cdist + argsort loops).  HBM/latency-bound integer work: one wave per centre, the candidates stream through the lanes 64 at a time
// with coalesced float reads (positions are re-packed to SoA by the caller-side wrapper below), selection by repeated wave minima.
// No atomics, deterministic; ties in distance go to the lower index.
#include <hip/hip_runtime.h>

#include "device_util.h"
#include "host_util.h"

namespace cbd {

// (distance, index) packed so that an unsigned 64-bit minimum orders by distance, then index; distances are >= 0, so the fp32 bit
// pattern is monotone
CBD_DEV unsigned long long pack_key(float d2, int j) { return ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned)j; }

CBD_DEV unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long w = __shfl_xor(v, o);
    v = w < v ? w : v;
  }
  return v;
}

// out[i][r] = index of the r-th nearest node of centre i (r < k), self excluded.  x / y / z: SoA coordinates [n].
__global__ __launch_bounds__(64) void knn_kernel(int n, int k, const float* __restrict__ x, const float* __restrict__ y,
                                                 const float* __restrict__ z, int* __restrict__ out) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const float cx = x[i], cy = y[i], cz = z[i];
  unsigned long long last = 0;   // key of the neighbour selected in the previous round (keys are unique: the index is part of them)
  bool first = true;
  for (int r = 0; r < k; ++r) {
    unsigned long long best = ~0ull;
    for (int j = lane; j < n; j += 64) {
      if (j == i) continue;
      const unsigned long long key = pack_key(dist2_nofma(x[j], y[j], z[j], cx, cy, cz), j);
      if ((first || key > last) && key < best) best = key;
    }
    best = wave_min_u64(best);
    if (lane == 0) out[(size_t)i * k + r] = (int)(best & 0xffffffffu);
    last = best;
    first = false;
  }
}

// Cutoff branch: cnt[i] neighbours of centre i in idx[i][0..cnt): all nodes closer than `cutoff` (strict, fp32) in index order when
// there are at most `cap` of them, else the `cap` nearest by increasing distance; a centre without any gets its nearest node.
__global__ __launch_bounds__(64) void radius_neighbors_kernel(int n, float cutoff, int cap, const float* __restrict__ x,
                                                              const float* __restrict__ y, const float* __restrict__ z,
                                                              int* __restrict__ idx, int* __restrict__ cnt) {
  const int i = blockIdx.x, lane = threadIdx.x;
  const float cx = x[i], cy = y[i], cz = z[i];
  // torch.cdist + `<`: the comparison is on the DISTANCE, not its square (process_mols.py:461,465)
  int total = 0;
  for (int j0 = 0; j0 < n; j0 += 64) {
    const int j = j0 + lane;
    const bool in = j < n && j != i && sqrtf(dist2_nofma(x[j], y[j], z[j], cx, cy, cz)) < cutoff;
    const unsigned long long m = __ballot(in);
    if (in) {
      const int slot = total + popc_below(m, lane);
      if (slot < cap) idx[(size_t)i * cap + slot] = j;
    }
    total += __popcll(m);
  }
  if (total > 0 && total <= cap) {
    if (lane == 0) cnt[i] = total;
    return;
  }
  // too many (the nearest `cap`, by distance) or none (the single nearest)
  const int want = total == 0 ? 1 : cap;
  unsigned long long last = 0;
  bool first = true;
  for (int r = 0; r < want; ++r) {
    unsigned long long best = ~0ull;
    for (int j = lane; j < n; j += 64) {
      if (j == i) continue;
      const unsigned long long key = pack_key(dist2_nofma(x[j], y[j], z[j], cx, cy, cz), j);
      if ((first || key > last) && key < best) best = key;
    }
    best = wave_min_u64(best);
    if (lane == 0) idx[(size_t)i * cap + r] = (int)(best & 0xffffffffu);
    last = best;
    first = false;
  }
  if (lane == 0) cnt[i] = want;
}

__global__ void aos_to_soa_kernel(int n, const float* __restrict__ pos, float* __restrict__ soa) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    soa[i] = pos[3 * i];
    soa[n + i] = pos[3 * i + 1];
    soa[2 * (size_t)n + i] = pos[3 * i + 2];
  }
}

}  // namespace cbd

using namespace cbd;

static int soa_of(int32_t n, const float* pos_dev, float** soa, hipStream_t s) {
  HIPCHK(hipMallocAsync(reinterpret_cast<void**>(soa), (size_t)3 * n * sizeof(float), s));
  hipLaunchKernelGGL(aos_to_soa_kernel, dim3((n + 255) / 256), dim3(256), 0, s, n, pos_dev, *soa);
  if (hipError_t e = hipGetLastError(); e != hipSuccess) {     // no early return that leaks the buffer
    (void)hipFreeAsync(*soa, s);
    *soa = nullptr;
    HIPCHK(e);
  }
  return 0;
}

int cbd_knn_graph(int32_t n, int32_t k, const float* pos_dev, int32_t* nbr_out_dev, void* stream) {
  if (n < 0 || k < 0 || (n > 0 && (!pos_dev || !nbr_out_dev))) return fail(CBD_ERR_ARG, "bad argument");
  if (k > n - 1 && n > 0) return fail(CBD_ERR_ARG, "k = %d exceeds n - 1 = %d", k, n - 1);
  if (n == 0 || k == 0) return 0;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* soa = nullptr;
  CHK(soa_of(n, pos_dev, &soa, s));
  hipLaunchKernelGGL(knn_kernel, dim3(n), dim3(64), 0, s, n, k, soa, soa + n, soa + 2 * (size_t)n, nbr_out_dev);
  const hipError_t launched = hipGetLastError();
  const hipError_t freed = hipFreeAsync(soa, s);              // freed on the error path as well
  HIPCHK(launched);
  HIPCHK(freed);
  return 0;
}

int cbd_radius_neighbors(int32_t n, float cutoff, int32_t cap, const float* pos_dev, int32_t* idx_out_dev, int32_t* cnt_out_dev,
                         void* stream) {
  if (n < 0 || cap < 1 || (n > 0 && (!pos_dev || !idx_out_dev || !cnt_out_dev))) return fail(CBD_ERR_ARG, "bad argument");
  if (n > 1 && cap > n - 1) return fail(CBD_ERR_ARG, "cap = %d exceeds n - 1 = %d", cap, n - 1);
  if (n == 0) return 0;
  if (n == 1) { HIPCHK(hipMemsetAsync(cnt_out_dev, 0, sizeof(int32_t), reinterpret_cast<hipStream_t>(stream))); return 0; }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* soa = nullptr;
  CHK(soa_of(n, pos_dev, &soa, s));
  hipLaunchKernelGGL(radius_neighbors_kernel, dim3(n), dim3(64), 0, s, n, cutoff, cap, soa, soa + n, soa + 2 * (size_t)n, idx_out_dev,
                     cnt_out_dev);
  const hipError_t launched = hipGetLastError();
  const hipError_t freed = hipFreeAsync(soa, s);
  HIPCHK(launched);
  HIPCHK(freed);
  return 0;
}
