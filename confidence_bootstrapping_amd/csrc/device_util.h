// Small device helpers shared by the graph / featurisation kernels (gfx950, 64-wide wavefronts).
#pragma once
#include <hip/hip_runtime.h>

namespace cbd {

#define CBD_DEV __device__ __forceinline__

// Squared distance accumulated like torch_cluster's scalar loop (no FMA contraction, so that the
// in/out decision of a pair is the same arithmetic in every kernel that evaluates it).
CBD_DEV float dist2_nofma(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = __fsub_rn(ax, bx), dy = __fsub_rn(ay, by), dz = __fsub_rn(az, bz);
  return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// radius(rec_pos / c, lig_pos / c, 1): models/score_model.py:568-570
CBD_DEV bool cross_pair_in(const float* lp, const float* rp, float c) {
  const float d2 = dist2_nofma(__fdiv_rn(rp[0], c), __fdiv_rn(rp[1], c), __fdiv_rn(rp[2], c),
                               __fdiv_rn(lp[0], c), __fdiv_rn(lp[1], c), __fdiv_rn(lp[2], c));
  return d2 < 1.0f;
}

// unit vector with F.normalize semantics (x / max(|x|, 1e-12)) and the norm
CBD_DEV void unit_vec(float x, float y, float z, float& ux, float& uy, float& uz, float& n) {
  n = sqrtf(x * x + y * y + z * z);
  const float inv = 1.0f / fmaxf(n, 1e-12f);
  ux = x * inv; uy = y * inv; uz = z * inv;
}

CBD_DEV int lane_id() { return threadIdx.x & 63; }
CBD_DEV int popc_below(unsigned long long m, int lane) { return __popcll(m & ((1ull << lane) - 1ull)); }

}  // namespace cbd
