// Host-side helpers shared by the engines behind include/cbdock.h: error reporting, tracked device allocations.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/cbdock.h"

// records the calling thread's last-error string (returned by cbd_last_error) and returns `code`
int cbd_fail(int code, const char* fmt, ...);
#define fail cbd_fail

#define HIPCHK(x)                                                                                              \
  do {                                                                                                         \
    hipError_t _e = (x);                                                                                       \
    if (_e != hipSuccess) return fail(CBD_ERR_HIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define CHK(x)            \
  do {                    \
    int _r = (x);         \
    if (_r != 0) return _r; \
  } while (0)

namespace cbd {

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
};

// Device arena: bump allocation out of large hipMalloc'd chunks.  reset() rewinds without freeing, so the per-complex and
// per-batch workspaces of successive complexes re-use the same memory (a hipMalloc/hipFree pair per buffer costs more than
// the kernels of a small complex's set-up); release() returns everything to the runtime.
struct DevPool {
  struct Chunk { char* base; size_t size, used; };
  std::vector<Chunk> chunks;
  size_t cur = 0;
  hipStream_t stream = nullptr;   // uploads: nullptr = synchronous copies on the default stream; else asynchronous on this stream + wait (the host vector may go away)
  static constexpr size_t kChunk = size_t(64) << 20, kAlign = 256;

  hipError_t raw(void** out, size_t bytes) {
    bytes = (std::max<size_t>(bytes, 1) + kAlign - 1) / kAlign * kAlign;
    for (; cur < chunks.size(); ++cur) {
      Chunk& c = chunks[cur];
      if (c.size - c.used >= bytes) {
        *out = c.base + c.used;
        c.used += bytes;
        return hipSuccess;
      }
    }
    void* q = nullptr;
    const size_t sz = std::max(bytes, kChunk);
    hipError_t e = hipMalloc(&q, sz);
    if (e != hipSuccess) return e;
    chunks.push_back(Chunk{static_cast<char*>(q), sz, bytes});
    cur = chunks.size() - 1;
    *out = q;
    return hipSuccess;
  }
  template <typename T>
  hipError_t alloc(T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = raw(&q, n * sizeof(T));
    if (e == hipSuccess) *p = reinterpret_cast<T*>(q);
    return e;
  }
  template <typename T>
  hipError_t upload(T** p, const std::vector<T>& h) {
    hipError_t e = alloc(p, h.size());
    if (e != hipSuccess) return e;
    if (h.empty()) return e;
    if (!stream) return hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    e = hipMemcpyAsync(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream);
    return e != hipSuccess ? e : hipStreamSynchronize(stream);
  }
  void reset() {
    for (Chunk& c : chunks) c.used = 0;
    cur = 0;
  }
  void release() {
    for (Chunk& c : chunks) (void)hipFree(c.base);
    chunks.clear();
    cur = 0;
  }
};

}  // namespace cbd
