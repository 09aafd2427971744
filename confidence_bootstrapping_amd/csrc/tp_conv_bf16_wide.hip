// bf16 tensor-product kernel, wide variant: ONE wave per SIMD owning 128 edges (four 32-edge sub-tiles per weight fragment).
//
// Why: tp_conv64_kernel (tp_conv_bf16.hip) is bound by the L2 -> CU weight stream (~36 TB/s at 0.32 of the bf16 peak).  Four
// sub-tiles per fragment halve the stream per FLOP again (7 KB per 128 edges and tile); the price is 1 wave per SIMD (~450 VGPRs), so
// nothing hides this wave's VALU epilogue unless the instruction stream itself overlaps it with the MFMAs: the loops are software
// pipelined -- the MFMAs of tile T+1 (into a second accumulator set) and the CG epilogue of tile T (on the first) sit in the same
// basic block without scheduling fences, so the compiler interleaves VALU work between the 28 MFMAs of a tile.
// Same weight stream, same reduction pieces and the same results (bitwise) as tp_conv64_kernel: per-edge arithmetic is unchanged.
#include <cstdlib>

#include "kernels.h"
#include "tp_conv_dev.h"

namespace cbd {

constexpr int W_NFRAG = 7;
constexpr int W_TILE_FRAGS = W_NFRAG * 64;
constexpr int W_SUB_FLOATS = NODE_DIM * OUT_STRIDE;
constexpr int W_NSUB = 4;
constexpr int W_SUB_WORDS = W_SUB_FLOATS + 32;       // row / message tile + 32 aggregating-node ids

struct ActW { bf16x8 v[W_NFRAG]; };

__device__ __forceinline__ void w_set_in(ActW& B, int seg, int q, f32x4 x) {
  const int k = 2 * seg + (q >> 1), o = 4 * (q & 1);
  B.v[k][o + 0] = (__bf16)x.x; B.v[k][o + 1] = (__bf16)x.y; B.v[k][o + 2] = (__bf16)x.z; B.v[k][o + 3] = (__bf16)x.w;
}
__device__ __forceinline__ void w_set_hidden(ActW& h, int m, const f32x16& acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) h.v[2 * m + (r >> 3)][r & 7] = (__bf16)fmaxf(acc[r], 0.f);
}

// acc[s] = A_tile * B[s] for the four sub-tiles; every A fragment is refilled in place with the next tile's data after its four uses
__device__ __forceinline__ void w_gemm(bf16x8 (&a)[W_NFRAG], const bf16x8* __restrict__ next, const ActW (&B)[W_NSUB], f32x16 (&acc)[W_NSUB]) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < W_NFRAG; ++q) {
#pragma unroll
    for (int s = 0; s < W_NSUB; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B[s].v[q], q == 0 ? zero : acc[s], 0, 0, 0);
    a[q] = next[q * 64];
  }
}

template <int IN, int OUT>
__global__ __launch_bounds__(64, 1) void tp_conv128_kernel(ConvArgs args) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x;
  const int j = lane & 31, hf = lane >> 5;

  int grp = -1, e0 = 0, cnt = 0, tile_local = 0;
  {
    int t = blockIdx.x;
    for (int g = 0; g < args.n_groups; ++g) {
      const int c = *args.g[g].count;
      const int nt = (c + 32 * W_NSUB - 1) / (32 * W_NSUB);
      if (grp < 0) {
        if (t < nt) { grp = g; e0 = t * 32 * W_NSUB; cnt = c; tile_local = W_NSUB * t; }
        else t -= nt;
      }
    }
  }
  if (grp < 0) return;
  const ConvGroup G = args.g[grp];

  const bf16x8* gp = reinterpret_cast<const bf16x8*>(G.wstream) + lane;
  bf16x8 a[W_NFRAG];
#pragma unroll
  for (int q = 0; q < W_NFRAG; ++q) a[q] = gp[q * 64];

  ActW Bx[W_NSUB];
  float v[W_NSUB][3];
  bf16x8 one = {0, 0, 0, 0, 0, 0, 0, 0};
  one[0] = hf == 0 ? (__bf16)1.0f : (__bf16)0.0f;      // activation fragment of the bias step: unit vector e_96
#pragma unroll
  for (int s = 0; s < W_NSUB; ++s) {
    float* xT = lds + s * W_SUB_WORDS;
    int* srcl = reinterpret_cast<int*>(xT + W_SUB_FLOATS);
    Bx[s].v[6] = one;
    const int e = e0 + 32 * s + j;
    const bool valid = e < cnt;
    const int ec = valid ? e : cnt - 1;
    const int src_r = G.src[ec], dst = G.dst[ec], aidx = G.attr_idx[ec];
    const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[ec];
    v[s][0] = vv.x; v[s][1] = vv.y; v[s][2] = vv.z;
    if (hf == 0) srcl[j] = valid ? src_r : -1;
    const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx * 32 + 16 * hf);
    const f32x4* ps = reinterpret_cast<const f32x4*>(G.node_in + (size_t)src_r * NODE_STRIDE + 16 * hf);
    const f32x4* pd = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dst * NODE_STRIDE + 16 * hf);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      w_set_in(Bx[s], 0, q, pa[q]);
      w_set_in(Bx[s], 1, q, ps[q]);
      w_set_in(Bx[s], 2, q, pd[q]);
    }
    const f32x4* pr = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dst * NODE_STRIDE + 40 * hf);
#pragma unroll
    for (int q = 0; q < 10; ++q) {
      if (40 * hf + 4 * q < 76) {
        const f32x4 r = pr[q];
        float* o = xT + (40 * hf + 4 * q) * 32 + j;
        o[0] = r.x; o[32] = r.y; o[64] = r.z; o[96] = r.w;
      }
    }
  }
  __syncthreads();

  const int i_lo = G.i0e_lo, i_hi = G.i0e_hi;
  const bool vec_on = G.vec_on != 0;
  const int T_vec = 3 + S.t0e;
  f32x16 accA[W_NSUB], accB[W_NSUB];
  ActW h[W_NSUB];
#pragma unroll
  for (int s = 0; s < W_NSUB; ++s) h[s].v[6] = one;
  // ---- first Linear (3 tiles)
  int T = 0;
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    const int tn = m < 2 ? T + 1 : (i_lo < i_hi ? 3 + i_lo : T_vec);
    w_gemm(a, gp + (size_t)tn * W_TILE_FRAGS, Bx, accA);
    T = tn;
    if (m == 2) mfma_operand_guard();
#pragma unroll
    for (int s = 0; s < W_NSUB; ++s) w_set_hidden(h[s], m, accA[s]);
  }

  const float* xc[W_NSUB];
#pragma unroll
  for (int s = 0; s < W_NSUB; ++s) xc[s] = lds + s * W_SUB_WORDS + j;
  float o0e[W_NSUB][16], k1o[W_NSUB][9], k1e[W_NSUB][9], k0o[W_NSUB][3];
#pragma unroll
  for (int s = 0; s < W_NSUB; ++s) {
#pragma unroll
    for (int r = 0; r < 16; ++r) o0e[s][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 9; ++r) { k1o[s][r] = 0.f; k1e[s][r] = 0.f; }
    k0o[s][0] = k0o[s][1] = k0o[s][2] = 0.f;
  }

  // ---- block 0e, software pipelined: MFMAs of tile i+1 and the epilogue of tile i in one basic block
  auto epi0e = [&](int i, const f32x16 (&acc)[W_NSUB]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < W_NSUB; ++s) {
      const float m = mid0e<IN>(xc[s], i, v[s]);
#pragma unroll
      for (int r = 0; r < 16; ++r) o0e[s][r] = fmaf(m, acc[s][r], o0e[s][r]);
    }
  };
  auto next_of = [&](int i) { return i + 1 < i_hi ? 3 + i + 1 : (vec_on ? T_vec : S.ntiles); };   // tile after 0e tile i
  if (i_lo < i_hi) {
    // `a` holds 0e tile i_lo; its product goes to accA while `a` is refilled with the tile after it
    w_gemm(a, gp + (size_t)next_of(i_lo) * W_TILE_FRAGS, h, accA);
    int i = i_lo;
#pragma unroll 1
    for (; i + 2 < i_hi; i += 2) {
      w_gemm(a, gp + (size_t)next_of(i + 1) * W_TILE_FRAGS, h, accB);   // tile i+1
      epi0e(i, accA);
      w_gemm(a, gp + (size_t)next_of(i + 2) * W_TILE_FRAGS, h, accA);   // tile i+2
      epi0e(i + 1, accB);
    }
    if (i + 1 < i_hi) {
      w_gemm(a, gp + (size_t)next_of(i + 1) * W_TILE_FRAGS, h, accB);
      epi0e(i, accA);
      epi0e(i + 1, accB);
    } else {
      epi0e(i, accA);
    }
    T = vec_on ? T_vec : S.ntiles;
  }

  // ---- vector / pseudoscalar blocks (not pipelined)
  auto vec_block = [&](auto mid_fn, int ntile, float (&keep)[W_NSUB][9]) __attribute__((always_inline)) {
#pragma unroll 1
    for (int t = 0; t < ntile; ++t) {
      w_gemm(a, gp + (size_t)(T + 1) * W_TILE_FRAGS, h, accA);
      ++T;
#pragma unroll
      for (int s = 0; s < W_NSUB; ++s)
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          float m[3];
          mid_fn(xc[s], VEC_TILE_I * t + q, v[s], m);
#pragma unroll
          for (int o = 0; o < 3; ++o) {
            const float w = accA[s][3 * q + o];
#pragma unroll
            for (int c = 0; c < 3; ++c) keep[s][3 * o + c] = fmaf(m[c], w, keep[s][3 * o + c]);
          }
        }
    }
  };
  if (vec_on) {
    vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, vv, m); }, S.t1o, k1o);
    if constexpr (OUT >= 2)
      vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, vv, m); }, S.t1e, k1e);
    if constexpr (OUT >= 3) {
#pragma unroll 1
      for (int t = 0; t < S.t0o; ++t) {
        w_gemm(a, gp + (size_t)(T + 1) * W_TILE_FRAGS, h, accA);
        ++T;
#pragma unroll
        for (int s = 0; s < W_NSUB; ++s)
#pragma unroll
          for (int q = 0; q < VEC_TILE_I; ++q) {
            const float m = mid0o<IN>(xc[s], VEC_TILE_I * t + q, v[s]);
#pragma unroll
            for (int o = 0; o < 3; ++o) k0o[s][o] = fmaf(m, accA[s][3 * q + o], k0o[s][o]);
          }
      }
    }
  }

  // ---- messages -> LDS, run-length sums per aggregating node and sub-tile (the pieces of tp_conv_kernel)
  __syncthreads();
#pragma unroll
  for (int s = 0; s < W_NSUB; ++s) {
    float* xT = lds + s * W_SUB_WORDS;
#pragma unroll
    for (int r = 0; r < 16; ++r) xT[((r & 3) + 8 * (r >> 2) + 4 * hf) * OUT_STRIDE + j] = o0e[s][r];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        xT[(COL_1O + 3 * (3 * hf + o) + c) * OUT_STRIDE + j] = k1o[s][3 * o + c];
        if constexpr (OUT >= 2) xT[(COL_1E + 3 * (3 * hf + o) + c) * OUT_STRIDE + j] = k1e[s][3 * o + c];
      }
    if constexpr (OUT >= 3) {
#pragma unroll
      for (int o = 0; o < 3; ++o) xT[(COL_0O + 3 * hf + o) * OUT_STRIDE + j] = k0o[s][o];
    }
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < W_NSUB; ++s) {
    const float* xT = lds + s * W_SUB_WORDS;
    const int* sl = reinterpret_cast<const int*>(xT + W_SUB_FLOATS);
    float* const fs = G.first_sum + (size_t)(tile_local + s) * NODE_STRIDE;
    float* const ls = G.last_sum + (size_t)(tile_local + s) * NODE_STRIDE;
    for (int col = lane; col < S.out_dim; col += 64) {
      const float* oc = xT + col * OUT_STRIDE;
      float sum = 0.f;
      int cur = sl[0], a0 = 0;
      for (int jj = 0; jj < 32; ++jj) {
        const int sj = sl[jj];
        if (sj != cur) {
          float* dst = a0 == 0 ? fs : G.run_acc + (size_t)cur * NODE_STRIDE;
          dst[col] = sum;
          sum = 0.f;
          a0 = jj;
          cur = sj;
        }
        sum += oc[jj];
      }
      if (cur >= 0) (a0 == 0 ? fs : ls)[col] = sum;
    }
  }
}

template <int IN, int OUT>
static hipError_t launch_one128(const ConvArgs& a, int grid, hipStream_t s) {
  constexpr int lds_bytes = W_NSUB * W_SUB_WORDS * 4;
  hipLaunchKernelGGL((tp_conv128_kernel<IN, OUT>), dim3(grid), dim3(64), lds_bytes, s, a);
  return hipGetLastError();
}

// grid: number of 128-edge waves
hipError_t launch_tp_conv_bf16_wide(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s) {
  if (grid <= 0) return hipSuccess;
  if (in_level == 0 && out_level == 1) return launch_one128<0, 1>(a, grid, s);
  if (in_level == 1 && out_level == 2) return launch_one128<1, 2>(a, grid, s);
  if (in_level == 2 && out_level == 3) return launch_one128<2, 3>(a, grid, s);
  if (in_level == 3 && out_level == 3) return launch_one128<3, 3>(a, grid, s);
  return hipErrorInvalidValue;
}

}  // namespace cbd
