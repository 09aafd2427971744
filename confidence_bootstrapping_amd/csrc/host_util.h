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

// tracked device allocation helper
struct DevPool {
  std::vector<void*> ptrs;
  template <typename T>
  hipError_t alloc(T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    ptrs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return hipSuccess;
  }
  template <typename T>
  hipError_t upload(T** p, const std::vector<T>& h) {
    hipError_t e = alloc(p, h.size());
    if (e != hipSuccess) return e;
    if (!h.empty()) e = hipMemcpy(*p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
  }
  void release() {
    for (void* p : ptrs) (void)hipFree(p);
    ptrs.clear();
  }
};

}  // namespace cbd
