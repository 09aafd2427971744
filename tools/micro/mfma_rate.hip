// Microbenchmark: what fraction of the fp32 MFMA peak do two resident waves per SIMD reach when each runs dependent 48-MFMA chains
// (v_mfma_f32_32x32x2_f32, the tp_conv tile) separated by a VALU epilogue of E instructions that reads the accumulator?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int EPI, int LDS_READS, int CH, int W>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(W, W))) void chains(float* out, int tiles, float seed) {
  extern __shared__ float lds[];
  f32x16 acc;
  float keep[16];
  for (int r = 0; r < 16; ++r) keep[r] = 0.f;
  float a = seed + threadIdx.x, b = seed * 0.5f;
  if (LDS_READS) lds[threadIdx.x] = seed;
  for (int t = 0; t < tiles; ++t) {
    for (int r = 0; r < 16; ++r) acc[r] = LDS_READS ? lds[(threadIdx.x + r + t) & 63] : b;
    if (CH == 1) {
#pragma unroll
      for (int k = 0; k < 48; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    } else {   // the same 48 k-steps as two independent chains of 24, summed afterwards
      f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 24; ++k) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc2, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < EPI; ++e) keep[e & 15] = fmaf(acc[e & 15], a, keep[e & 15]);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += keep[r] + acc[r];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int EPI, int LDS_READS, int CH, int W>
static void run(const char* name) {
  const int waves_per_simd = W;
  const int tiles = 400, grid = 256 * 4 * waves_per_simd * 4;   // 4 generations of waves
  float* out;
  hipMalloc(&out, (size_t)grid * 64 * 4);
  // occupancy is set by the register budget (amdgpu_waves_per_eu pads the VGPR count), so the waves spread evenly over the SIMDs
  const int lds = 1024;
  hipFuncSetAttribute((const void*)chains<EPI, LDS_READS, CH, W>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((chains<EPI, LDS_READS, CH, W>), dim3(grid), dim3(64), lds, 0, out, tiles, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * tiles * 48 * 32 * 32 * 2 * 2;
    if (rep == 2) printf("%-44s waves/SIMD %d: %.3f ms, %.1f TFLOP/s = %.3f of 157.3\n", name, waves_per_simd, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
  }
  hipFree(out);
}

#define ROW(E, L, C, NAME) run<E, L, C, 1>(NAME); run<E, L, C, 2>(NAME); run<E, L, C, 3>(NAME); run<E, L, C, 4>(NAME);
int main() {
  ROW(0, 0, 1, "pure dependent chains")
  ROW(0, 0, 2, "two chains of 24, pure")
  ROW(48, 1, 1, "chain + 48 VALU epilogue + LDS acc init")
  ROW(128, 1, 1, "chain + 128 VALU epilogue + LDS acc init")
  ROW(128, 1, 2, "two chains of 24 + 128 VALU + LDS acc init")
  return 0;
}
