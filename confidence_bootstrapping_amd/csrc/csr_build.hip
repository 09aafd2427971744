// Edge grouping of the fine-tuning step: edges grouped by target row for the fixed-order segmented sums that replace
// the atomic scatters of the reference's training graph (torch_scatter.scatter in models/tensor_layers.py:206, autograd's index_add for
// node_attr[edge_index]).  perm = STABLE argsort of the scatter index, rowptr = first position of every row in the sorted order.
// HBM/latency-bound integer work: an LSD radix sort (rocPRIM, stable) over exactly the bits the row count needs, enqueued on the
// caller's stream with caller-provided scratch -- no host synchronisation (torch.sort's stable path synchronises the stream, 33 times
// per training step in the profile of round 3, and that is what kept the host from running ahead of the GPU).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "host_util.h"
#include "../../include/cbdock.h"

namespace cbd {

__global__ __launch_bounds__(256) void csr_keys_kernel(long long n, const long long* __restrict__ index, unsigned* __restrict__ keys) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) keys[i] = (unsigned)index[i];
}

// rowptr[r] = number of sorted keys < r  (lower bound), r in [0, n_rows]
__global__ __launch_bounds__(256) void csr_rowptr_kernel(long long n, long long n_rows, const unsigned* __restrict__ sorted,
                                                         long long* __restrict__ rowptr) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r > n_rows) return;
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if ((long long)sorted[mid] < r) lo = mid + 1; else hi = mid;
  }
  rowptr[r] = lo;
}

static size_t align256(size_t v) { return (v + 255) & ~size_t(255); }

// ---- several groupings in ONE sort: segment s contributes keys (s << bits) | index_s[k]; a stable sort of the combined keys is the
//      concatenation of the segments' stable sorts.  18 index tensors per training step = 108 launches one by one, 7 this way.
constexpr int CSR_MAX_SEG = 32;
struct CsrBatch {
  int n_seg, bits;
  long long off[CSR_MAX_SEG + 1];       // positions of the segments in the combined key array
  long long roff[CSR_MAX_SEG + 1];      // positions of the segments' row pointers ((rows_s + 1) each) in the combined rowptr array
  const long long* index[CSR_MAX_SEG];
};

__device__ inline int seg_of(const long long* off, int n_seg, long long i) {   // largest s with off[s] <= i
  int lo = 0, hi = n_seg - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= i) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__global__ __launch_bounds__(256) void csr_batch_keys_kernel(CsrBatch b, unsigned* __restrict__ keys) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= b.off[b.n_seg]) return;
  const int s = seg_of(b.off, b.n_seg, i);
  keys[i] = ((unsigned)s << b.bits) | (unsigned)b.index[s][i - b.off[s]];
}

__global__ __launch_bounds__(256) void csr_batch_perm_kernel(CsrBatch b, const long long* __restrict__ sorted_pos, long long* __restrict__ perm) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= b.off[b.n_seg]) return;
  perm[i] = sorted_pos[i] - b.off[seg_of(b.off, b.n_seg, i)];
}

__global__ __launch_bounds__(256) void csr_batch_rowptr_kernel(CsrBatch b, const unsigned* __restrict__ sorted, long long* __restrict__ rowptr) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= b.roff[b.n_seg]) return;
  const int s = seg_of(b.roff, b.n_seg, i);
  const unsigned key = ((unsigned)s << b.bits) | (unsigned)(i - b.roff[s]);      // row r of segment s (r = rows_s: one past the last row)
  long long lo = b.off[s], hi = b.off[s + 1];
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    if (sorted[mid] < key) lo = mid + 1; else hi = mid;
  }
  rowptr[i] = lo - b.off[s];
}

}  // namespace cbd

extern "C" {

int cbd_csr_build(int64_t n, int64_t n_rows, const int64_t* index_dev, int64_t* perm_dev, int64_t* rowptr_dev, void* scratch_dev,
                  size_t scratch_bytes, size_t* scratch_needed, void* stream) {
  using namespace cbd;
  if (n < 0 || n_rows < 0 || n > 0x7fffffffLL || n_rows > 0x7fffffffLL) return fail(CBD_ERR_ARG, "cbd_csr_build: bad sizes");
  int bits = 1;
  while (bits < 32 && (1LL << bits) < n_rows) ++bits;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  size_t sort_bytes = 0;
  if (n > 0) {
    const hipError_t r = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const unsigned*)nullptr, (unsigned*)nullptr,
                                                   rocprim::counting_iterator<long long>(0), (long long*)nullptr, (size_t)n, 0u, (unsigned)bits, st);
    if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_csr_build: radix sort sizing: %s", hipGetErrorString(r));
  }
  const size_t key_bytes = align256((size_t)n * sizeof(unsigned));
  const size_t need = 2 * key_bytes + align256(sort_bytes) + 256;
  if (scratch_needed) *scratch_needed = need;
  if (!scratch_dev) return 0;                       /* sizing call */
  if (scratch_bytes < need) return fail(CBD_ERR_ARG, "cbd_csr_build: scratch of %zu bytes, %zu needed", scratch_bytes, need);
  if (!rowptr_dev || (n > 0 && (!index_dev || !perm_dev))) return fail(CBD_ERR_ARG, "cbd_csr_build: null pointer");
  char* base = reinterpret_cast<char*>((reinterpret_cast<size_t>(scratch_dev) + 255) & ~size_t(255));
  unsigned* keys_in = reinterpret_cast<unsigned*>(base);
  unsigned* keys_out = reinterpret_cast<unsigned*>(base + key_bytes);
  void* tmp = base + 2 * key_bytes;
  if (n > 0) {
    hipLaunchKernelGGL(csr_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (long long)n,
                       reinterpret_cast<const long long*>(index_dev), keys_in);
    const hipError_t r = rocprim::radix_sort_pairs(tmp, sort_bytes, (const unsigned*)keys_in, keys_out, rocprim::counting_iterator<long long>(0),
                                                   reinterpret_cast<long long*>(perm_dev), (size_t)n, 0u, (unsigned)bits, st);
    if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_csr_build: radix sort: %s", hipGetErrorString(r));
  }
  hipLaunchKernelGGL(csr_rowptr_kernel, dim3((unsigned)((n_rows + 1 + 255) / 256)), dim3(256), 0, st, (long long)n, (long long)n_rows,
                     (const unsigned*)keys_out, reinterpret_cast<long long*>(rowptr_dev));
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_csr_build: %s", hipGetErrorString(r));
  return 0;
}

int cbd_csr_build_batched(int32_t n_seg, const int64_t* const* index_dev, const int64_t* seg_n, const int64_t* seg_rows, int64_t* perm_dev,
                          int64_t* rowptr_dev, void* scratch_dev, size_t scratch_bytes, size_t* scratch_needed, void* stream) {
  using namespace cbd;
  if (n_seg <= 0 || n_seg > CSR_MAX_SEG || !seg_n || !seg_rows) return fail(CBD_ERR_ARG, "cbd_csr_build_batched: 1..%d segments", CSR_MAX_SEG);
  CsrBatch b{};
  b.n_seg = n_seg;
  int bits = 1, seg_bits = 1;
  while ((1 << seg_bits) < n_seg) ++seg_bits;
  for (int s = 0; s < n_seg; ++s) {
    if (seg_n[s] < 0 || seg_rows[s] < 0) return fail(CBD_ERR_ARG, "cbd_csr_build_batched: negative size");
    while (bits < 31 && (1LL << bits) < seg_rows[s] + 1) ++bits;         // + 1: the rowptr search uses the key `rows_s`
    b.off[s + 1] = b.off[s] + seg_n[s];
    b.roff[s + 1] = b.roff[s] + seg_rows[s] + 1;
    b.index[s] = index_dev ? reinterpret_cast<const long long*>(index_dev[s]) : nullptr;
  }
  b.bits = bits;
  const long long n = b.off[n_seg], nr = b.roff[n_seg];
  if (bits + seg_bits > 32 || n > 0x7fffffffLL) return fail(CBD_ERR_ARG, "cbd_csr_build_batched: keys do not fit 32 bits");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  size_t sort_bytes = 0;
  if (n > 0) {
    const hipError_t r = rocprim::radix_sort_pairs(nullptr, sort_bytes, (const unsigned*)nullptr, (unsigned*)nullptr, rocprim::counting_iterator<long long>(0),
                                                   (long long*)nullptr, (size_t)n, 0u, (unsigned)(bits + seg_bits), st);
    if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_csr_build_batched: radix sort sizing: %s", hipGetErrorString(r));
  }
  const size_t key_bytes = align256((size_t)n * sizeof(unsigned)), pos_bytes = align256((size_t)n * sizeof(long long));
  const size_t need = 2 * key_bytes + pos_bytes + align256(sort_bytes) + 256;
  if (scratch_needed) *scratch_needed = need;
  if (!scratch_dev) return 0;                       /* sizing call */
  if (scratch_bytes < need) return fail(CBD_ERR_ARG, "cbd_csr_build_batched: scratch of %zu bytes, %zu needed", scratch_bytes, need);
  if (!rowptr_dev || !index_dev || (n > 0 && !perm_dev)) return fail(CBD_ERR_ARG, "cbd_csr_build_batched: null pointer");
  for (int s = 0; s < n_seg; ++s)
    if (seg_n[s] > 0 && !index_dev[s]) return fail(CBD_ERR_ARG, "cbd_csr_build_batched: null index pointer");
  char* base = reinterpret_cast<char*>((reinterpret_cast<size_t>(scratch_dev) + 255) & ~size_t(255));
  unsigned* keys_in = reinterpret_cast<unsigned*>(base);
  unsigned* keys_out = reinterpret_cast<unsigned*>(base + key_bytes);
  long long* pos_out = reinterpret_cast<long long*>(base + 2 * key_bytes);
  void* tmp = base + 2 * key_bytes + pos_bytes;
  if (n > 0) {
    hipLaunchKernelGGL(csr_batch_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b, keys_in);
    const hipError_t r = rocprim::radix_sort_pairs(tmp, sort_bytes, (const unsigned*)keys_in, keys_out, rocprim::counting_iterator<long long>(0),
                                                   pos_out, (size_t)n, 0u, (unsigned)(bits + seg_bits), st);
    if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_csr_build_batched: radix sort: %s", hipGetErrorString(r));
    hipLaunchKernelGGL(csr_batch_perm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b, (const long long*)pos_out,
                       reinterpret_cast<long long*>(perm_dev));
  }
  hipLaunchKernelGGL(csr_batch_rowptr_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, b, (const unsigned*)keys_out,
                     reinterpret_cast<long long*>(rowptr_dev));
  const hipError_t r = hipGetLastError();
  if (r != hipSuccess) return fail(CBD_ERR_HIP, "cbd_csr_build_batched: %s", hipGetErrorString(r));
  return 0;
}

}  // extern "C"
