// bf16 tensor-product message passing, role split with LDS-RESIDENT weight tiles (BASELINE.json configs[3]; options bf16 + bf16_roles).
//
// What bounds tp_conv64_kernel (tp_conv_bf16.hip) is the CU's vector-memory return path: every wave streams its group's whole FCBlock
// (57 tiles x 6 KB = 342 KB per 64 edges) plus its gathers through it, 8 waves per CU ask for 64 B/clk at the full MFMA rate, the path
// delivers ~43 (PMC round 4: matrix pipe busy 0.52).  Sharing tiles between the waves of a workgroup through an LDS
// ring was tried three times and lost to the per-tile rendezvous.  Here nothing is handed over at run time:
//   * the cross / receptor edge groups of an interaction layer run as three virtual SLICES (engine.hip): 0e tiles [0, 19), 0e tiles
//     [19, 38), vector blocks.  The vector slice (and the small ligand-ligand group) keeps the streaming kernel;
//   * THIS kernel runs the two 0e slices: persistent workgroups of 8 waves, one per CU; a workgroup is bound to ONE role = (FCBlock,
//     0e half), copies that role's 19 tiles (114 KB) into LDS once, and its waves then loop over 64-edge units of the role's groups,
//     reading every weight fragment from LDS (ds_read_b128: 256 B/clk/CU, its own return path) -- no barrier after the initial fill.
//     A 0e slice needs no gathered-row image: its mids are the 32 scalar features of the destination row (already gathered for the
//     first Linear) and six 1o . direction dots -- 5 KB of LDS per wave instead of 19.8;
//   * the first Linear (3 tiles) is recomputed per slice from the streamed tiles (18 KB per unit instead of 342).
// Workgroups are assigned to roles in proportion to the roles' unit counts, computed on the device from the edge counts (every
// workgroup does the same integer arithmetic); the units of a role are dealt round-robin to the waves of its workgroups.
// Results: the same pieces (first_sum / last_sum / run_acc, columns [0, NS)) as the streaming kernel's 0e slices -- bitwise.
#include <cstdlib>

#include "kernels.h"
#include "tp_conv_dev.h"
#include "tp_conv_bf16_dev.h"

namespace cbd {

constexpr int P_WAVES = 8;
constexpr int P_MAX_ROLES = 8;
constexpr int P_RES_TILES = 19;                                   // tiles of a 0e half (t0e = 38 at 74 -> 74)
constexpr int P_W_FLOATS = P_RES_TILES * V2_TILE_FRAGS * 4;       // resident weights: 19 x 6 KB, as floats
constexpr int P_MID_ROWS = P_RES_TILES;                           // mids of the slice: [19][64 edges]
constexpr int P_WAVE_FLOATS = P_MID_ROWS * 64 + 64;               // + the 64 aggregating-node ids; >= one message tile [NS][33]
static_assert(P_MID_ROWS * 64 >= NS * OUT_STRIDE, "the message tile re-uses the mid table");
constexpr int P_LDS_BYTES = (P_W_FLOATS + P_WAVES * P_WAVE_FLOATS) * 4;
static_assert(P_W_FLOATS % 2 == 0 && P_WAVE_FLOATS % 2 == 0, "reduce_runs reads the message tile as f32x2: even tile bases");
static_assert(P_LDS_BYTES <= 160 * 1024, "one workgroup per CU");

struct RoleTable {
  int n_roles;
  unsigned char role_of[CONV_MAX_GROUPS];      // role of every entry of ConvArgs::g
  int lo[P_MAX_ROLES], hi[P_MAX_ROLES];        // 0e tiles [lo, hi) of the role
  const float* wstream[P_MAX_ROLES];
};

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// LDS write -> read ordering inside ONE wave (the waves of a persistent workgroup never meet at a barrier after the weight fill)
// -- LDS instructions of one wave execute in issue order: only the compiler has to be kept from moving a lane's reads above other
// lanes' writes, and the data must have returned before registers are re-used
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int DIAG = 0>
__global__ __launch_bounds__(P_WAVES * 64, 1) void tp_conv64p_kernel(ConvArgs args, RoleTable rt) {
  constexpr ConvShape S = conv_shape(3, 3, true);      // the bf16 stream's layout (merged vector tails)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int j = lane & 31, hf = lane >> 5;
  const int lane4hf = 16 * hf;
  float* const wl = lds;                                            // resident tiles [tile][q][lane] x 16 B
  float* const mids = lds + P_W_FLOATS + wave * P_WAVE_FLOATS;      // [19][64]; later the message tile [NS][33]
  int* const srcl = reinterpret_cast<int*>(mids + P_MID_ROWS * 64);

  // ---- roles: units per entry (lane g <-> entry g), workgroups per role, this workgroup's role and rank
  int units = 0, my_role = -1;
  if (lane < args.n_groups) {
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    const int* cp = *reinterpret_cast<const int* const*>(ka + offsetof(ConvArgs, g) + (size_t)lane * sizeof(ConvGroup) + offsetof(ConvGroup, count));
    units = (*cp + 63) / 64;
    my_role = *reinterpret_cast<const unsigned char*>(ka + ((sizeof(ConvArgs) + alignof(RoleTable) - 1) / alignof(RoleTable)) * alignof(RoleTable) + offsetof(RoleTable, role_of) + lane);
  }
  int total = wave_sum(units);
  if (total == 0) return;
  const int n_wg = gridDim.x;
  int role = -1, rank = 0, n_role_wg = 1, role_units = 0, acc_units = 0, wg_lo = 0;
#pragma unroll 1
  for (int r = 0; r < rt.n_roles; ++r) {
    const int w = wave_sum(my_role == r ? units : 0);
    acc_units += w;
    int wg_hi = (int)((long long)n_wg * acc_units / total);
    if (w > 0 && wg_hi <= wg_lo) wg_hi = wg_lo + 1;                 // every role with work gets a workgroup
    if (r == rt.n_roles - 1 || wg_hi > n_wg) wg_hi = n_wg;
    if (role < 0 && (int)blockIdx.x >= wg_lo && (int)blockIdx.x < wg_hi && w > 0) { role = r; rank = blockIdx.x - wg_lo; n_role_wg = wg_hi - wg_lo; role_units = w; }
    wg_lo = wg_hi;
  }
  if (role < 0) return;
  // DIAG 4: per-wave phase clocks summed over the wave's units (same record layout as the streaming kernel; tools/conv_clock.py)
  unsigned long long st_t0 = 0, st_r0 = 0, c_gather = 0, c_lin = 0, c_0e = 0, c_units = 0;
  if constexpr (DIAG == 4) { st_t0 = stamp(); st_r0 = __builtin_amdgcn_s_memrealtime(); }
  const int i_lo = rt.lo[role], i_hi = rt.hi[role];
  const int n_res = i_hi - i_lo;

  // ---- the role's tiles -> LDS, once
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(rt.wstream[role]) + (size_t)(3 + i_lo) * V2_TILE_FRAGS;
    f32x4* dst = reinterpret_cast<f32x4*>(wl);
    for (int k = threadIdx.x; k < n_res * V2_TILE_FRAGS; k += P_WAVES * 64) dst[k] = src[k];
  }
  __syncthreads();

  // exclusive prefix of the role's units over the entries (unit -> entry)
  const int mine = my_role == role ? units : 0;
  int incl = mine;
#pragma unroll
  for (int d = 1; d < CONV_MAX_GROUPS; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if (lane >= d) incl += v;
  }

#pragma unroll 1
  for (int u = rank * P_WAVES + wave; u < role_units; u += P_WAVES * n_role_wg) {
    const unsigned long long owner = __ballot(u >= incl - mine && u < incl);
    const int grp = __builtin_ctzll(owner);
    const int unit_in_group = u - (__builtin_amdgcn_readlane(incl, grp) - __builtin_amdgcn_readlane(mine, grp));
    const ConvGroup G = args.g[grp];
    const int cnt = *G.count;
    const int e0 = unit_in_group * 64;
    unsigned long long u0 = 0, u1 = 0, u2 = 0;
    if constexpr (DIAG == 4) u0 = stamp();
    const int tile_local = 2 * unit_in_group;

    // ---- weight stream of the first Linear (tiles 0..2 from global memory, as in the streaming kernel) and its bias rows
    const GFrag gp = (GFrag)reinterpret_cast<const bf16x8*>(G.wstream);
    bf16x8 a[V2_NFRAG];
#pragma unroll
    for (int q = 0; q < V2_NFRAG; ++q) a[q] = gp[q * 64 + lane];
    const GBias gbias = (GBias)reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(G.wstream) + (size_t)(S.ntiles + 1) * V2_TILE_FRAGS);
    f32x16 cb;
    {
      const GPtr<f32x4> gb4 = (GPtr<f32x4>)gbias;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 b = gb4[hf + 2 * qq];
        cb[4 * qq + 0] = b.x; cb[4 * qq + 1] = b.y; cb[4 * qq + 2] = b.z; cb[4 * qq + 3] = b.w;
      }
    }
    // ---- gathers: indices of both sub-tiles, then attributes and the two 32-column node segments (+ the 1o columns for the dot mids)
    Act6 Bx0, Bx1;
    float v0[3], v1[3];
    int src_r[2], dstn[2], aidx[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const int e = e0 + 32 * sub + j;
      const bool valid = e < cnt;
      const int ec = valid ? e : cnt - 1;
      src_r[sub] = G.src[ec]; dstn[sub] = G.dst[ec]; aidx[sub] = G.attr_idx[ec];
      const f32x4 vv = reinterpret_cast<const f32x4*>(G.vec)[ec];
      if (sub) { v1[0] = vv.x; v1[1] = vv.y; v1[2] = vv.z; } else { v0[0] = vv.x; v0[1] = vv.y; v0[2] = vv.z; }
      if (hf == 0) srcl[32 * sub + j] = valid ? src_r[sub] : -1;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 ta[2][4], ts[2][4], td[2][4];
    float d1o[2][9];
    const bool dots = i_hi > NS;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx[sub] * 32 + 16 * hf);
      const f32x4* ps = reinterpret_cast<const f32x4*>(G.node_in + (size_t)src_r[sub] * NODE_STRIDE + 16 * hf);
      const f32x4* pd = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dstn[sub] * NODE_STRIDE + 16 * hf);
#pragma unroll
      for (int q = 0; q < 4; ++q) { ta[sub][q] = pa[q]; ts[sub][q] = ps[q]; td[sub][q] = pd[q]; }
      if (dots) {   // lane half hf takes the 1o vectors 3 hf .. 3 hf + 2 of the destination row
        const float* p1 = G.node_in + (size_t)dstn[sub] * NODE_STRIDE + COL_1O + 9 * hf;
#pragma unroll
        for (int c = 0; c < 9; ++c) d1o[sub][c] = p1[c];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      Act6& Bx = sub ? Bx1 : Bx0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        v2_set_in(Bx, 0, q, ta[sub][q]);
        v2_set_in(Bx, 1, q, ts[sub][q]);
        v2_set_in(Bx, 2, q, td[sub][q]);
      }
    }
    // ---- mids of the slice -> LDS [i - i_lo][edge]: scalar features (= columns of td) and the 1o . direction dots
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t = td[sub][q];
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int col = 16 * hf + 4 * q + c;
          if (col >= i_lo && col < i_hi) mids[(col - i_lo) * 64 + 32 * sub + j] = tv[c];
        }
      }
    if (dots) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const float* vv = sub ? v1 : v0;
#pragma unroll
        for (int uu = 0; uu < 3; ++uu) {
          const int i = NS + 3 * hf + uu;
          const float m = d1o[sub][3 * uu] * vv[0] + d1o[sub][3 * uu + 1] * vv[1] + d1o[sub][3 * uu + 2] * vv[2];
          if (i >= i_lo && i < i_hi) mids[(i - i_lo) * 64 + 32 * sub + j] = m;
        }
      }
    }
    wave_lds_fence();
    if constexpr (DIAG == 4) u1 = stamp();

    // ---- first Linear (3 streamed tiles): h = ReLU(W1 x + b1) in the C/D layout = B operand of the second Linear
    f32x16 acc0, acc1;
    Act6 h0, h1;
    float raw_next = gbias[32 + (lane & 31)];
    v2_gemm<DIAG>(a, cb, gp + (size_t)1 * V2_TILE_FRAGS, gbias + (size_t)2 * 32, raw_next, lane, lane4hf, Bx0, Bx1, acc0, acc1);
    v2_set_hidden(h0, 0, acc0);
    v2_set_hidden(h1, 0, acc1);
    bias_ready(cb);
    v2_gemm<DIAG>(a, cb, gp + (size_t)2 * V2_TILE_FRAGS, gbias, raw_next, lane, lane4hf, Bx0, Bx1, acc0, acc1);
    v2_set_hidden(h0, 1, acc0);
    v2_set_hidden(h1, 1, acc1);
    bias_ready(cb);
    v2_gemm<DIAG | 16>(a, cb, gp, gbias, raw_next, lane, lane4hf, Bx0, Bx1, acc0, acc1);   // | 16: nothing is streamed behind the last tile
    mfma_operand_guard();                                                                  // (the fragments are refilled from LDS below)
    v2_set_hidden(h0, 2, acc0);
    v2_set_hidden(h1, 2, acc1);
    bias_ready(cb);

    if constexpr (DIAG == 4) u2 = stamp();
    // ---- the slice's 0e tiles from LDS
    const bf16x8* wq = reinterpret_cast<const bf16x8*>(wl) + lane;      // fragment q of resident tile t: wq[(t * 6 + q) * 64]
#pragma unroll
    for (int q = 0; q < V2_NFRAG; ++q) a[q] = wq[q * 64];
    f32x16 o0 = {}, o1 = {};
    const float* mc = mids + j;
#pragma unroll 1
    for (int t = 0; t < n_res; ++t) {
      const float m0 = mc[t * 64], m1 = mc[t * 64 + 32];
      const bf16x8* nx = wq + (size_t)(t + 1 < n_res ? t + 1 : t) * V2_TILE_FRAGS;
      const f32x16 zero = {};
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < V2_NFRAG; ++q) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], h0.v[q], q == 0 ? zero : acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], h1.v[q], q == 0 ? zero : acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // fragment q - 2 is refilled two pairs late (an LDS read returns faster than an L2 one: the MFMAs that read the register
        // have long started), the last two behind the chain
        if (q >= 2) a[q - 2] = nx[(q - 2) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
      a[4] = nx[4 * 64];
      a[5] = nx[5 * 64];
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] = fmaf(m0, acc0[r], o0[r]); o1[r] = fmaf(m1, acc1[r], o1[r]); }
    }
    // ---- bias of the slice's tiles: sum_i b_i m_i as one small matrix product (the block's bias rows as a [32 x 48] bf16 tile)
    {
      const GFrag gb0e = (GFrag)reinterpret_cast<const bf16x8*>(reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(G.wstream) + (size_t)(S.ntiles + 1) * V2_TILE_FRAGS) + (size_t)(S.ntiles + 1) * 32);
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) {
        const bf16x8 ab = gb0e[s3 * 64 + lane];
        bf16x8 bm0, bm1;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int i = 16 * s3 + 8 * hf + jj;
          const bool in = i >= i_lo && i < i_hi;
          const float m0 = in ? mc[(i - i_lo) * 64] : 0.f, m1 = in ? mc[(i - i_lo) * 64 + 32] : 0.f;
          bm0[jj] = (__bf16)m0;
          bm1[jj] = (__bf16)m1;
        }
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bm0, o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bm1, o1, 0, 0, 0);
      }
    }
    if constexpr (DIAG == 4) { const unsigned long long u3 = stamp(); c_gather += u1 - u0; c_lin += u2 - u1; c_0e += u3 - u2; ++c_units; }
    // ---- messages -> LDS tile [NS][33] (over the mid table), run-length sums per aggregating node; one sub-tile after the other
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
      wave_lds_fence();          // the mids (first pass) / the previous sub-tile's reads are done
      const f32x16& o = sub ? o1 : o0;
#pragma unroll
      for (int r = 0; r < 16; ++r) mids[((r & 3) + 8 * (r >> 2) + 4 * hf) * OUT_STRIDE + j] = o[r];
      wave_lds_fence();
      reduce_runs<NODE_STRIDE, OUT_STRIDE, NS, 0>(mids, srcl + 32 * sub, lane, NS,      // (tile base P_W_FLOATS + wave * P_WAVE_FLOATS: even, asserted below)
                                                  G.first_sum + (size_t)(tile_local + sub) * NODE_STRIDE,
                                           G.last_sum + (size_t)(tile_local + sub) * NODE_STRIDE, G.run_acc);
    }
    wave_lds_fence();
  }
  if constexpr (DIAG == 4) {
    const int rec = blockIdx.x * P_WAVES + wave;
    if (lane == 0 && args.stamps && rec < 8192 && c_units) {
      unsigned long long* o = args.stamps + (size_t)rec * 8;
      o[0] = st_t0; o[1] = st_r0; o[2] = stamp(); o[3] = __builtin_amdgcn_s_memrealtime();
      o[4] = st_t0 + c_gather; o[5] = o[4] + c_lin; o[6] = o[5] + c_0e; o[7] = o[4] + c_units;
    }
  }
}

// a: the 0e-slice entries of a layer (vec_on == 0); grid: workgroups (<= CUs).  Roles = distinct (weight stream, tile range) pairs.
hipError_t launch_tp_conv_bf16p(const ConvArgs& a, int n_wg, hipStream_t s) {
  if (a.n_groups <= 0) return hipSuccess;
  RoleTable rt{};
  for (int g = 0; g < a.n_groups; ++g) {
    const ConvGroup& G = a.g[g];
    if (G.vec_on || G.i0e_hi - G.i0e_lo > P_RES_TILES || G.i0e_hi <= G.i0e_lo) return hipErrorInvalidValue;
    int r = -1;
    for (int k = 0; k < rt.n_roles; ++k)
      if (rt.wstream[k] == G.wstream && rt.lo[k] == G.i0e_lo && rt.hi[k] == G.i0e_hi) r = k;
    if (r < 0) {
      if (rt.n_roles == P_MAX_ROLES) return hipErrorInvalidValue;
      r = rt.n_roles++;
      rt.wstream[r] = G.wstream; rt.lo[r] = G.i0e_lo; rt.hi[r] = G.i0e_hi;
    }
    rt.role_of[g] = (unsigned char)r;
  }
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64p_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64p_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  static const int diag = getenv("CBD_BF16_DIAG") ? atoi(getenv("CBD_BF16_DIAG")) : 0;
  if (diag == 4) hipLaunchKernelGGL((tp_conv64p_kernel<4>), dim3(n_wg), dim3(P_WAVES * 64), P_LDS_BYTES, s, a, rt);
  else hipLaunchKernelGGL((tp_conv64p_kernel<0>), dim3(n_wg), dim3(P_WAVES * 64), P_LDS_BYTES, s, a, rt);
  return hipGetLastError();
}

}  // namespace cbd
