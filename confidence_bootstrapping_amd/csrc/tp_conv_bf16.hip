// bf16-operand tensor-product message passing for gfx950 (BASELINE.json configs[3]; cbd_set_option("bf16", 1)).
//
// Same math as tp_conv.hip (FCBlock -> FasterTensorProduct -> segmented sum, reference models/tensor_layers.py:195-206,66-117),
// re-tiled for the bf16 matrix pipe.  One 32x32x16 bf16 MFMA is 8 passes (32 cycles), so a 32-edge tile of per-edge weights costs
// 192 cycles of matrix pipe against 6 KB of weight stream: a wave-private stream (the fp32 kernel's layout) needs 128 B/clk/CU at
// the full matrix rate, the L2 delivers ~56 (MI355X_MICROARCH.md "L2": 34.5 TB/s chip-wide).  Measured on the way here (C4, PMC
// passes of round 2, profiles/r02_c_pmc_bf16_c4_tp_conv64_summary.txt): with every wave streaming its own tiles -- even with two
// 32-edge sub-tiles per weight fragment -- the kernel sat at 0.27-0.32 of the bf16 peak while moving ~36 TB/s out of the L2:
// bandwidth bound, not latency bound (turning the FLAT loads into global_load changed nothing).
//
// Layout of this kernel:
//   * a workgroup of NW waves (4 or 8) owns NW x 32 consecutive edges of ONE edge group and walks the group's weight-tile sequence
//     in lockstep; every 7 KB tile is fetched from L2 ONCE per workgroup (each wave loads one or two of its seven 1 KB fragments
//     two tiles ahead, parks them in registers for one iteration, then stores them into a 3-stage LDS ring) and read from LDS by
//     all waves: L2 weight traffic per (edge, tile) drops by NW against the wave-private stream;
//   * one barrier per tile keeps the ring consistent; to keep the matrix pipe and the VALU busy at the same time in spite of the
//     lockstep, the two halves of the workgroup run half a tile out of phase: waves [0, NW/2) issue the tile's MFMAs first and
//     their CG epilogue afterwards, waves [NW/2, NW) first finish the PREVIOUS tile's epilogue (its accumulator is still in
//     registers) and then issue this tile's MFMAs -- on every SIMD one wave's MFMA chain runs under the other's VALU work;
//   * the bias enters through the matrix core: K is extended from 96 to 112 (a 7th k-step whose activation fragment is the unit
//     vector e_96 and whose weight fragment carries the bias row): the accumulator starts from the inline constant 0, no bias
//     table in LDS;
//   * fp32 everywhere outside the two Linears (gathered rows, CG contraction, messages, reduction).
// Weight stream (pack_conv_stream_bf16, engine.hip): (ntiles + 1) tiles of [7 k-steps][64 lanes][8 bf16] = 7 KB.
// Reduction pieces (first_sum / last_sum / run_acc per 32-edge tile) are exactly those of tp_conv_kernel: conv_finalize is unchanged.
#include <cstdlib>

#include "kernels.h"
#include "tp_conv_dev.h"

namespace cbd {

constexpr int V2_NFRAG = 7;                      // 6 k-steps of the 96 inputs + 1 bias step
constexpr int V2_TILE_FRAGS = V2_NFRAG * 64;     // 16-byte fragments per tile
constexpr int V2_SUB_FLOATS = NODE_DIM * OUT_STRIDE;   // 74 x 33 floats per 32-edge tile (>= 76 x 32 of the gather image)
static_assert(V2_SUB_FLOATS >= 76 * 32, "gather image must fit the message tile");
constexpr int V2_STAGES = 4;
constexpr int V2_STAGE_BYTES = V2_NFRAG * 1024;
constexpr int V2_WAVE_BYTES = V2_SUB_FLOATS * 4 + 128;   // row / message tile + the 32 aggregating-node ids

struct Act7 { bf16x8 v[V2_NFRAG]; };

__device__ __forceinline__ void v2_set_in(Act7& B, int seg, int q, f32x4 x) {
  const int k = 2 * seg + (q >> 1), o = 4 * (q & 1);
  B.v[k][o + 0] = (__bf16)x.x; B.v[k][o + 1] = (__bf16)x.y; B.v[k][o + 2] = (__bf16)x.z; B.v[k][o + 3] = (__bf16)x.w;
}
__device__ __forceinline__ void v2_set_hidden(Act7& h, int m, const f32x16& acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) h.v[2 * m + (r >> 3)][r & 7] = (__bf16)fmaxf(acc[r], 0.f);
}

// acc = A_tile * B on seven A fragments held in registers
__device__ __forceinline__ void v2_gemm_regs(const bf16x8 (&a)[V2_NFRAG], const Act7& B, f32x16& acc) {
  const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B.v[q], q == 0 ? zero : acc, 0, 0, 0);
}
// the seven A fragments of an LDS stage -> registers (one conflict-free ds_read_b128 each, all issued back to back)
__device__ __forceinline__ void v2_read_stage(const bf16x8* __restrict__ stage, bf16x8 (&a)[V2_NFRAG]) {
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) a[q] = stage[q * 64];
}

template <int IN, int OUT, int NW>
__global__ __launch_bounds__(64 * NW) void tp_conv_wg_kernel(ConvArgs args) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  constexpr int WG_EDGES = 32 * NW;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = lane & 31, hf = lane >> 5;
  bf16x8* const ring = reinterpret_cast<bf16x8*>(lds_raw);                                          // [3][7][64] fragments
  float* const xT = reinterpret_cast<float*>(lds_raw + V2_STAGES * V2_STAGE_BYTES + wave * V2_WAVE_BYTES);   // this wave's tile
  int* const srcl = reinterpret_cast<int*>(xT + V2_SUB_FLOATS);

  // ---- which group / edge range does this workgroup own?  (edge counts live on the device)
  int grp = -1, e0 = 0, cnt = 0, tile0 = 0;
  {
    int t = blockIdx.x;
    for (int g = 0; g < args.n_groups; ++g) {
      const int c = *args.g[g].count;
      const int nt = (c + WG_EDGES - 1) / WG_EDGES;
      if (grp < 0) {
        if (t < nt) { grp = g; e0 = t * WG_EDGES; cnt = c; tile0 = t * NW; }
        else t -= nt;
      }
    }
  }
  if (grp < 0) return;       // uniform over the workgroup
  const ConvGroup G = args.g[grp];

  // ---- the group's weight-tile sequence: 3 tiles of the first Linear, the 0e tiles [i_lo, i_hi), then the vector / pseudoscalar
  //      blocks (virtual slices of an edge group run only part of it, ConvGroup::i0e_lo/hi, vec_on)
  const int i_lo = G.i0e_lo, n0e = G.i0e_hi - G.i0e_lo;
  const int nvec = G.vec_on ? S.t1o + S.t1e + S.t0o : 0;
  const int n_seq = 3 + n0e + nvec;
  auto tile_of = [&](int k) { return k < 3 ? k : (k < 3 + n0e ? 3 + i_lo + (k - 3) : (k < n_seq ? 3 + S.t0e + (k - 3 - n0e) : S.ntiles)); };

  // ---- weight ring: fragment f of a tile is fetched from L2 by wave f % NW.  Tile k+3 is loaded (to registers) during iteration k,
  //      stored to stage (k+3) % 4 during iteration k+1, read from LDS into the waves' spare fragment registers during iteration k+2 and
  //      multiplied in iteration k+3: no load, LDS or global, is ever waited for inside an iteration.  One barrier per iteration.
  const bf16x8* const gw = reinterpret_cast<const bf16x8*>(G.wstream) + lane;
  constexpr int MYF = (V2_NFRAG + NW - 1) / NW;     // fragments per wave (some waves have one fewer)
  bf16x8 greg[MYF];
  auto fetch = [&](int k) {
    const bf16x8* p = gw + (size_t)tile_of(k) * V2_TILE_FRAGS;
#pragma unroll
    for (int m = 0; m < MYF; ++m) {
      const int f = wave + m * NW;
      if (f < V2_NFRAG) greg[m] = p[f * 64];
    }
  };
  auto park = [&](int k) {
    bf16x8* st = ring + (k % V2_STAGES) * V2_TILE_FRAGS + lane;
#pragma unroll
    for (int m = 0; m < MYF; ++m) {
      const int f = wave + m * NW;
      if (f < V2_NFRAG) st[f * 64] = greg[m];
    }
  };
  fetch(0);

  // ---- gather this wave's 32 edges.  Lanes past the end of the group read the group's last edge (unconditional loads) and are
  //      dropped at the end through src = -1.
  Act7 Bx;
  float v[3];
  {
    bf16x8 one = {0, 0, 0, 0, 0, 0, 0, 0};
    one[0] = hf == 0 ? (__bf16)1.0f : (__bf16)0.0f;      // activation fragment of the bias step: unit vector e_96
    Bx.v[6] = one;
    const int e = e0 + 32 * wave + j;
    const bool valid = e < cnt;
    const int ec = valid ? e : cnt - 1;
    const int src_r = G.src[ec], dst = G.dst[ec], aidx = G.attr_idx[ec];
    const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[ec];
    v[0] = vv.x; v[1] = vv.y; v[2] = vv.z;
    if (hf == 0) srcl[j] = valid ? src_r : -1;
    const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx * 32 + 16 * hf);
    const f32x4* ps = reinterpret_cast<const f32x4*>(G.node_in + (size_t)src_r * NODE_STRIDE + 16 * hf);
    const f32x4* pd = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dst * NODE_STRIDE + 16 * hf);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v2_set_in(Bx, 0, q, pa[q]);
      v2_set_in(Bx, 1, q, ps[q]);
      v2_set_in(Bx, 2, q, pd[q]);
    }
    // full destination row -> transposed LDS copy xT[col][j]; lane half hf copies cols 40hf .. 40hf+39 (cols >= 76 are padding)
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
  park(0);
  fetch(1);
  park(1);
  fetch(2);
  __syncthreads();           // tiles 0 and 1 in stages 0 and 1, the gathered rows in LDS

  f32x16 acc;
  Act7 h;
  h.v[6] = Bx.v[6];
  bf16x8 aA[V2_NFRAG], aB[V2_NFRAG];     // A fragments of the tile being multiplied / of the next one (roles alternate)
  v2_read_stage(ring + lane, aA);
  // one ring step of iteration k: the next tile's fragments LDS -> registers, tile k+2 registers -> LDS, tile k+3 L2 -> registers
  auto advance = [&](int k, bf16x8 (&nxt)[V2_NFRAG]) __attribute__((always_inline)) {
    v2_read_stage(ring + ((k + 1) % V2_STAGES) * V2_TILE_FRAGS + lane, nxt);
    park(k + 2);
    fetch(k + 3);
  };
  // ---- first Linear (3 tiles, everybody in the same order): h = ReLU(W1 x + b1) in the C/D register layout = B operand of Linear 2
  v2_gemm_regs(aA, Bx, acc); advance(0, aB); v2_set_hidden(h, 0, acc); __syncthreads();
  v2_gemm_regs(aB, Bx, acc); advance(1, aA); v2_set_hidden(h, 1, acc); __syncthreads();
  v2_gemm_regs(aA, Bx, acc); advance(2, aB); v2_set_hidden(h, 2, acc); __syncthreads();

  const float* xc = xT + j;
  float o0e[16], k1o[9], k1e[9], k0o[3];
#pragma unroll
  for (int r = 0; r < 16; ++r) o0e[r] = 0.f;
#pragma unroll
  for (int r = 0; r < 9; ++r) { k1o[r] = 0.f; k1e[r] = 0.f; }
  k0o[0] = k0o[1] = k0o[2] = 0.f;

  // CG epilogue of sequence position k (k >= 3) on the accumulator of that tile
  auto epilogue = [&](int k) __attribute__((always_inline)) {
    if (k < 3 + n0e) {                                     // block 0e: one tile per mid index, 32 output scalars
      const float m = mid0e<IN>(xc, i_lo + (k - 3), v);
#pragma unroll
      for (int r = 0; r < 16; ++r) o0e[r] = fmaf(m, acc[r], o0e[r]);
      return;
    }
    int t = k - 3 - n0e;                                   // vector / pseudoscalar blocks: tile = 5 mid indices x 6 outputs,
    if (t < S.t1o) {                                       // lane half hf owns outputs 3hf..3hf+2
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        float m[3];
        mid1o<IN>(xc, VEC_TILE_I * t + q, v, m);
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const float w = acc[3 * q + o];
#pragma unroll
          for (int c = 0; c < 3; ++c) k1o[3 * o + c] = fmaf(m[c], w, k1o[3 * o + c]);
        }
      }
      return;
    }
    t -= S.t1o;
    if constexpr (OUT >= 2) {
      if (t < S.t1e) {
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          float m[3];
          mid1e<IN>(xc, VEC_TILE_I * t + q, v, m);
#pragma unroll
          for (int o = 0; o < 3; ++o) {
            const float w = acc[3 * q + o];
#pragma unroll
            for (int c = 0; c < 3; ++c) k1e[3 * o + c] = fmaf(m[c], w, k1e[3 * o + c]);
          }
        }
        return;
      }
      t -= S.t1e;
    }
    if constexpr (OUT >= 3) {
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const float m = mid0o<IN>(xc, VEC_TILE_I * t + q, v);
#pragma unroll
        for (int o = 0; o < 3; ++o) k0o[o] = fmaf(m, acc[3 * q + o], k0o[o]);
      }
    }
  };

  // ---- second Linear + CG contraction, one barrier per tile.  The upper half of the workgroup runs its epilogue one iteration
  //      late (before the next tile's MFMAs), so that on every SIMD one wave's MFMA chain overlaps the other wave's VALU epilogue.
  const bool late = wave >= NW / 2;
  auto step = [&](int k, const bf16x8 (&cur)[V2_NFRAG], bf16x8 (&nxt)[V2_NFRAG]) __attribute__((always_inline)) {
    if (late && k > 3) epilogue(k - 1);
    // fence: `nxt` is re-loaded from LDS below and was the A operand of this wave's PREVIOUS MFMA chain; a late wave issued that chain
    // just before the barrier, so it must have drained first -- the epilogue above waits on its accumulator, nothing may be hoisted
    // over it (in-flight MFMA operand hazard, tp_conv_dev.h)
    __builtin_amdgcn_sched_barrier(0);
    v2_gemm_regs(cur, h, acc);
    advance(k, nxt);
    if (!late) epilogue(k);
    __syncthreads();
  };
#pragma unroll 1
  for (int k = 3; k < n_seq; k += 2) {
    step(k, aB, aA);
    if (k + 1 < n_seq) step(k + 1, aA, aB);
  }
  if (late && n_seq > 3) epilogue(n_seq - 1);

  // ---- messages -> LDS (re-using the gathered-row tile, stride 33), then run-length sums per aggregating node (no atomics, bitwise
  //      reproducible): the run that starts at the tile's first edge -> first_sum[tile], the run that reaches edge 31 -> last_sum[tile],
  //      any other run is the node's only contribution from this group and is stored directly (conv_finalize adds the pieces)
  __syncthreads();   // every read of xT (mids) is complete before it is overwritten
#pragma unroll
  for (int r = 0; r < 16; ++r) xT[((r & 3) + 8 * (r >> 2) + 4 * hf) * OUT_STRIDE + j] = o0e[r];
#pragma unroll
  for (int o = 0; o < 3; ++o)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      xT[(COL_1O + 3 * (3 * hf + o) + c) * OUT_STRIDE + j] = k1o[3 * o + c];
      if constexpr (OUT >= 2) xT[(COL_1E + 3 * (3 * hf + o) + c) * OUT_STRIDE + j] = k1e[3 * o + c];
    }
  if constexpr (OUT >= 3) {
#pragma unroll
    for (int o = 0; o < 3; ++o) xT[(COL_0O + 3 * hf + o) * OUT_STRIDE + j] = k0o[o];
  }
  __syncthreads();
  const int tile_local = tile0 + wave;
  float* const fs = G.first_sum + (size_t)tile_local * NODE_STRIDE;
  float* const ls = G.last_sum + (size_t)tile_local * NODE_STRIDE;
  for (int col = lane; col < S.out_dim; col += 64) {
    const float* oc = xT + col * OUT_STRIDE;
    float sum = 0.f;
    int cur = srcl[0], a0 = 0;
    for (int jj = 0; jj < 32; ++jj) {
      const int sj = srcl[jj];
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

template <int IN, int OUT, int NW>
static hipError_t launch_wg(const ConvArgs& a, int grid, hipStream_t s) {
  constexpr int lds_bytes = V2_STAGES * V2_STAGE_BYTES + NW * V2_WAVE_BYTES;
  static bool attr_set = false;
  if (!attr_set) {   // > 64 KB of dynamic LDS needs the opt-in
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv_wg_kernel<IN, OUT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL((tp_conv_wg_kernel<IN, OUT, NW>), dim3(grid), dim3(64 * NW), lds_bytes, s, a);
  return hipGetLastError();
}

// Waves per workgroup of the bf16 kernel: 8 (one workgroup per CU, 256 edges share every weight tile) unless CBD_BF16_NW=4 (two
// workgroups per CU, 128 edges per tile).  The engine sizes its grids with bf16_wg_waves().
int bf16_wg_waves() {
  static const int nw = [] { const char* p = getenv("CBD_BF16_NW"); const int v = p ? atoi(p) : 8; return v == 4 ? 4 : 8; }();
  return nw;
}

// grid: number of workgroups (sum over groups of ceil(cap / (32 * NW)))
hipError_t launch_tp_conv_bf16(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s) {
  if (grid <= 0) return hipSuccess;
  const bool w8 = bf16_wg_waves() == 8;
  if (in_level == 0 && out_level == 1) return w8 ? launch_wg<0, 1, 8>(a, grid, s) : launch_wg<0, 1, 4>(a, grid, s);
  if (in_level == 1 && out_level == 2) return w8 ? launch_wg<1, 2, 8>(a, grid, s) : launch_wg<1, 2, 4>(a, grid, s);
  if (in_level == 2 && out_level == 3) return w8 ? launch_wg<2, 3, 8>(a, grid, s) : launch_wg<2, 3, 4>(a, grid, s);
  if (in_level == 3 && out_level == 3) return w8 ? launch_wg<3, 3, 8>(a, grid, s) : launch_wg<3, 3, 4>(a, grid, s);
  return hipErrorInvalidValue;
}

}  // namespace cbd
