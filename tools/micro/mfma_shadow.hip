// Microbenchmark: do VALU instructions execute in the shadow of a v_mfma_f32_32x32x2_f32 (16 passes = 64 cycles) on gfx950?
// One wave per SIMD runs a dependent 48-MFMA chain per tile with N independent v_fma_f32 placed behind every MFMA.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_shadow.hip -o /tmp/mfma_shadow && /tmp/mfma_shadow
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, int W, int PK, int MF>
__global__ __attribute__((amdgpu_flat_work_group_size(64, 64), amdgpu_waves_per_eu(W, W))) void shadow(float* out, int tiles, float seed) {
  __shared__ float lds[256];
  lds[threadIdx.x] = seed; lds[threadIdx.x + 64] = seed;
  bf16x8 ah, bh;
  for (int r = 0; r < 8; ++r) { ah[r] = (__bf16)(seed + r); bh[r] = (__bf16)(seed - r); }
  int li = threadIdx.x;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = seed;
  float a = seed + threadIdx.x, b = seed * 0.5f;
  float k[8];
  f32x2 k2[8];
  for (int r = 0; r < 8; ++r) { k[r] = seed + r; k2[r] = f32x2{seed + r, seed - r}; }
  const f32x2 a2 = {a, b};
  for (int t = 0; t < tiles; ++t) {
#pragma unroll
    for (int s = 0; s < 48; ++s) {
      if (MF) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
      else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
      for (int e = 0; e < N; ++e) {
        if (PK == 2) { k[e & 7] = lds[(li + e) & 127]; }
        else if (PK) k2[e & 7] = __builtin_elementwise_fma(k2[e & 7], a2, a2);
        else k[e & 7] = fmaf(k[e & 7], a, b);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += acc[r];
  for (int r = 0; r < 8; ++r) s += k[r] + k2[r].x + k2[r].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int N, int W, int PK, int MF = 0>
static void run(const char* name) {
  const int tiles = 400, grid = 256 * 4 * W * 4;
  float* out;
  hipMalloc(&out, (size_t)grid * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((shadow<N, W, PK, MF>), dim3(grid), dim3(64), 0, 0, out, tiles, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double flops = (double)grid * tiles * 48 * 32 * 32 * 2 * (MF ? 16 : 2), peak = MF ? 2500.0 : 157.3;
  printf("%-28s %-9s %2d behind every MFMA, waves/SIMD %d: %.3f ms, MFMA %.3f of peak\n", name, MF ? "bf16 x16" : "f32 x2", N, W, ms, flops / ms / 1e9 / peak);
  hipFree(out);
}

int main() {
  run<0, 1, 0>("v_fma_f32");  run<1, 1, 0>("v_fma_f32");  run<2, 1, 0>("v_fma_f32");  run<4, 1, 0>("v_fma_f32");  run<8, 1, 0>("v_fma_f32");  run<12, 1, 0>("v_fma_f32");
  run<4, 2, 0>("v_fma_f32");  run<8, 2, 0>("v_fma_f32");
  run<4, 1, 1>("v_pk_fma_f32"); run<8, 1, 1>("v_pk_fma_f32");
  run<2, 1, 2>("ds_read_b32"); run<4, 1, 2>("ds_read_b32"); run<4, 2, 2>("ds_read_b32");
  run<0, 1, 0, 1>("v_fma_f32"); run<0, 2, 0, 1>("v_fma_f32"); run<2, 2, 0, 1>("v_fma_f32"); run<4, 2, 0, 1>("v_fma_f32"); run<6, 2, 0, 1>("v_fma_f32"); run<8, 2, 0, 1>("v_fma_f32");
  run<4, 2, 1, 1>("v_pk_fma_f32"); run<4, 2, 2, 1>("ds_read_b32");
  return 0;
}
