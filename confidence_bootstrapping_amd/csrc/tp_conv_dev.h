// Device-side building blocks shared by the inference kernels (tp_conv.hip) and the training kernels (tp_train.hip):
// MFMA operand policies, one-tile GEMM with register prefetch, and the CG "mid" evaluators of FasterTensorProduct.
#pragma once
#include <cstddef>

#include "kernels.h"
#include "reduce_runs.h"

namespace cbd {

// Weight tiles are read with global loads whose base address is wave-uniform (an SGPR pair, advanced per tile on the scalar unit) plus
// the constant per-lane offset in one VGPR plus an immediate (< 4 KB, hence one base per four 1-KB fragments): no vector address
// arithmetic per tile, exact vmcnt waits.  `pin_s` keeps such a base opaque and scalar across the MFMA chain.
// (`pin_s`, `GPtr`: reduce_runs.h)
// diagnostic stamp (CBD_CONV_VARIANT=8 build only): s_memtime pinned in place (cdna_hip_programming.md section 7)
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}

// Which edge group does workgroup `t` of a tensor-product launch own, and which of its tiles?  The edge counts live on the device
// (ConvGroup::count), so every wave has to read them: lane g reads group g's count (two dependent vector loads for all groups at
// once -- a scalar loop over the groups costs two dependent scalar loads PER GROUP, ~6 k cycles for a wave of the 16th group), a wave
// prefix sum turns the tile counts into tile ranges, and a ballot finds the owner.  Returns false when t is past the last tile.
// `edges_per_wg` = 32 (tp_conv_kernel) or 64 (tp_conv64_kernel).
__device__ __forceinline__ bool find_group(const ConvArgs& args, int t, int lane, int edges_per_wg, int& grp, int& tile_in_group, int& cnt) {
  const int ng = args.n_groups;
  int c = 0;
  if (lane < ng) {
    // per-lane read of the kernel-argument segment (indexing `args` by lane would make hipcc copy the 4 KB argument to scratch)
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();   // constant address space -> generic
    const int* cp = *reinterpret_cast<const int* const*>(ka + offsetof(ConvArgs, g) + (size_t)lane * sizeof(ConvGroup) + offsetof(ConvGroup, count));
    c = *cp;
  }
  const int nt = (c + edges_per_wg - 1) / edges_per_wg;
  int incl = nt;
#pragma unroll
  for (int d = 1; d < CONV_MAX_GROUPS; d <<= 1) {
    const int v = __shfl_up(incl, d);
    if (lane >= d) incl += v;
  }
  const unsigned long long owner = __ballot(t >= incl - nt && t < incl);
  if (owner == 0) return false;
  grp = __builtin_ctzll(owner);
  cnt = __builtin_amdgcn_readlane(c, grp);
  tile_in_group = t - (__builtin_amdgcn_readlane(incl, grp) - __builtin_amdgcn_readlane(nt, grp));
  return true;
}

constexpr int OUT_STRIDE = 34;   // message tile stride: lane = column reads two edges per ds_read_b64, 32 lanes on 64 distinct banks (reduce_runs.h)

// ---- operand policies: how the two Linears of the radial MLP run on the matrix cores ------------------------------------
// One 32x32 tile: acc = bias + A_tile * B.  The A fragments of the CURRENT tile are in registers, loaded one tile ahead
// straight from global/L2 in MFMA operand order; each fragment register is refilled with the NEXT tile's data right after its
// last use, so the loads have a whole tile of MFMA time to land and no barrier or LDS staging is involved.
//
// OpsF32: exact fp32, v_mfma_f32_32x32x2_f32, 48 k-steps, 12 float4 fragments (12 KB tile).
struct OpsF32 {
  static constexpr bool EXACT_F32 = true;
  static constexpr bool NODE_PROJ = true;    // node parts of the first Linear through per-node projections (ConvGroup::psrc / pdst)
  using Frag = f32x4;
  static constexpr int NFRAG = KSTEPS / 4;                 // 12
  static constexpr int TILE_FRAGS = TILE_W_FLOATS / 4;     // 768 fragments of 16 B per tile
  struct Act { float v[KSTEPS]; };                         // B operand: lane (j, hf) holds act[edge j][k(s, hf)] for the 48 k-steps
  // gathered input columns 16hf + 4q .. +3 of part `seg` (edge_attr | x_src | x_dst)
  static __device__ __forceinline__ void set_in(Act& B, int seg, int q, f32x4 x) {
    B.v[16 * seg + 4 * q + 0] = x.x; B.v[16 * seg + 4 * q + 1] = x.y; B.v[16 * seg + 4 * q + 2] = x.z; B.v[16 * seg + 4 * q + 3] = x.w;
  }
  // ReLU'd accumulator of hidden tile m IS the B operand of the second Linear (W2's k order follows the C/D layout)
  static __device__ __forceinline__ void set_hidden(Act& h, int m, const f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) h.v[16 * m + r] = relu1(acc[r]);
  }
  static __device__ __forceinline__ void gemm(Frag (&a)[NFRAG], const Frag* __restrict__ next, const Act& B, f32x16& acc) {
#pragma unroll
    for (int sg = 0; sg < NFRAG; ++sg) {
      const f32x4 w = a[sg];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, B.v[4 * sg + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, B.v[4 * sg + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, B.v[4 * sg + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, B.v[4 * sg + 3], acc, 0, 0, 0);
      a[sg] = next[sg * 64];
      // keep the refill right behind its last use: without this hipcc sinks all 12 loads to the end of the tile and the
      // next tile then starts by waiting a full L2 round trip
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // first Linear, edge part only (K = 32: the edge_attr columns = fragments 0..3 of a tile); the node parts enter through `acc`
  // (ConvGroup::psrc / pdst).  The three first-Linear tiles need 3 x 4 fragments = the whole register tile: the kernel loads them all
  // in its prologue (load_first), so the first Linear runs without a single wait on memory, and tile m's four registers are
  // refilled in place with fragments 4m..4m+3 of the first second-Linear tile (`next`) right after their use.
  // ---- the same three routines on a wave-uniform tile base (tp_conv_kernel; bond_conv / tp_train keep the per-lane pointer forms)
  // The refill of fragment sg directly follows its four MFMAs (no one-pair delay as in tp_conv_bf16.hip::v2_gemm).  That is safe here
  // because (i) the bases p0..p2 are pinned in SGPRs before the chain and every load uses an immediate, so NO VALU instruction can be
  // scheduled into the just-died registers, and (ii) a returning global load is ordered behind the operand reads of the MFMAs issued
  // before it: the fp32 MFMA reads its scalar-per-lane operands over its 16 passes, but the load's data needs an L2 round trip (> 500
  // cycles) where the four MFMAs take 256.  Covered by the bitwise-repeat tests of all three operand modes and tools/bf16_repeat.py.
  static __device__ __forceinline__ void gemm_u(Frag (&a)[NFRAG], GPtr<Frag> next, int lane, const Act& B, f32x16& acc) {
    GPtr<Frag> p0 = next, p1 = next + 4 * 64, p2 = next + 8 * 64;
    pin_s(p0); pin_s(p1); pin_s(p2);
#pragma unroll
    for (int sg = 0; sg < NFRAG; ++sg) {
      const f32x4 w = a[sg];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, B.v[4 * sg + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, B.v[4 * sg + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, B.v[4 * sg + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, B.v[4 * sg + 3], acc, 0, 0, 0);
      a[sg] = (sg < 4 ? p0 : sg < 8 ? p1 : p2)[lane + (sg & 3) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  static __device__ __forceinline__ void load_first_u(Frag (&a)[NFRAG], GPtr<Frag> gp, int lane) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int s = 0; s < NFRAG / 3; ++s) a[(NFRAG / 3) * m + s] = gp[(size_t)m * TILE_FRAGS + s * 64 + lane];
  }
  template <int M>
  static __device__ __forceinline__ void gemm_first_u(Frag (&a)[NFRAG], GPtr<Frag> next, int lane, const Act& B, f32x16& acc) {
    GPtr<Frag> pm = next + (NFRAG / 3) * M * 64;   // fragments 4M .. 4M+3 of the first second-Linear tile
    pin_s(pm);
#pragma unroll
    for (int s = 0; s < NFRAG / 3; ++s) {
      constexpr int base = (NFRAG / 3) * M;
      const f32x4 w = a[base + s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, B.v[4 * s + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, B.v[4 * s + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, B.v[4 * s + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, B.v[4 * s + 3], acc, 0, 0, 0);
      a[base + s] = pm[lane + s * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  static constexpr int FIRST_FRAGS = NFRAG / 3;            // 4
  static __device__ __forceinline__ void load_first(Frag (&a)[NFRAG], const Frag* __restrict__ gp) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int s = 0; s < FIRST_FRAGS; ++s) a[FIRST_FRAGS * m + s] = gp[(size_t)m * TILE_FRAGS + s * 64];
  }
  template <int M>
  static __device__ __forceinline__ void gemm_first(Frag (&a)[NFRAG], const Frag* __restrict__ next, const Act& B, f32x16& acc) {
#pragma unroll
    for (int s = 0; s < FIRST_FRAGS; ++s) {
      constexpr int base = FIRST_FRAGS * M;
      const f32x4 w = a[base + s];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.x, B.v[4 * s + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.y, B.v[4 * s + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.z, B.v[4 * s + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.w, B.v[4 * s + 3], acc, 0, 0, 0);
      a[base + s] = next[(base + s) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// Measured on gfx950 (ROCm 7.2 hipcc): a VALU write into a VGPR that an already-issued v_mfma_f32_32x32x16_bf16 has not yet read
// as its A/B operand corrupts that operand.  The MFMA reads its operands when it starts executing, which with two waves sharing
// the matrix pipe can be tens of cycles after issue; hipcc re-uses the registers of a fragment that died at its last MFMA for
// the address arithmetic of the refill loads and pads nothing in between (seen as run-to-run differences of ~1e-4 relative in
// single output columns of the OpsBf16x3 kernel; 8 wait states were not enough, 64 were).  The bf16 policies therefore
//   * compute the refill address of a group BEFORE its MFMAs and pin it (no VALU instruction sits between the MFMAs and the
//     refill loads, and while the loads are in flight their destination registers are not free for re-use), and
//   * close the one group whose operands die without a refill (end of the first Linear) with mfma_operand_guard().
// tests/test_gpu_parity.py / test_gpu_bf16.py compare repeated trajectories bitwise in all three modes as the tripwire.
__device__ __forceinline__ void mfma_operand_guard() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15");
  __builtin_amdgcn_sched_barrier(0);
}

// OpsBf16: bf16 operands, fp32 accumulate, v_mfma_f32_32x32x16_bf16, 6 k-steps of 16, 6 fragments of 8 bf16 (6 KB tile).
// Lane (r = lane&31, h = lane>>5) holds A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7 (cdna_hip_programming.md
// section 3).  Registers 8s..8s+7 of hidden tile m are the fragment of k-step 2m+s: element j of lane half h is hidden unit
// 32m + 16s + 8(j>>2) + 4h + (j&3); W2's k order is permuted to match at pack time (pack_conv_stream_bf16).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct OpsBf16 {
  static constexpr bool EXACT_F32 = false;
  static constexpr bool NODE_PROJ = false;   // the whole first Linear stays on the matrix cores: this policy is not MFMA-bound, the
                                             // projection kernel (20 us per layer) would cost more than the two tiles it saves
  using Frag = bf16x8;
  static constexpr int NFRAG = KDIM / 16;                  // 6
  static constexpr int TILE_FRAGS = NFRAG * 64;            // 384 fragments of 16 B per tile
  struct Act { bf16x8 v[NFRAG]; };
  // k-step 2*seg + sub covers input columns 16hf + 8sub .. +7 of part seg; q = 2*sub + half
  static __device__ __forceinline__ void set_in(Act& B, int seg, int q, f32x4 x) {
    const int k = 2 * seg + (q >> 1), o = 4 * (q & 1);
    B.v[k][o + 0] = (__bf16)x.x; B.v[k][o + 1] = (__bf16)x.y; B.v[k][o + 2] = (__bf16)x.z; B.v[k][o + 3] = (__bf16)x.w;
  }
  static __device__ __forceinline__ void set_hidden(Act& h, int m, const f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) h.v[2 * m + (r >> 3)][r & 7] = (__bf16)relu1(acc[r]);
  }
  static __device__ __forceinline__ void gemm(Frag (&a)[NFRAG], const Frag* __restrict__ next, const Act& B, f32x16& acc) {
#pragma unroll
    for (int q = 0; q < NFRAG; ++q) {
      const Frag* p = next + q * 64;
      pin(p);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B.v[q], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a[q] = *p;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  static __device__ __forceinline__ void gemm_first(Frag (&a)[NFRAG], const Frag* __restrict__ next, const Act& B, f32x16& acc) {
#pragma unroll
    for (int q = 0; q < NFRAG; ++q) {
      const Frag* p = next + q * 64;
      pin(p);
      __builtin_amdgcn_sched_barrier(0);
      if (q < NFRAG / 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B.v[q], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a[q] = *p;
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

// OpsBf16x3: fp32 operands represented EXACTLY as the sum of three bf16 planes (hi + mid + lo = 3 x 8 significand bits) and
// multiplied on the bf16 matrix cores with fp32 accumulation: of the nine plane products the six largest are kept
// (hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi); the dropped ones are <= 2^-24 relative each, i.e. at the level of one fp32
// rounding.  36 bf16 MFMAs per tile (1152 cycles) instead of 48 fp32 ones (3072): the "fp32 emulation on low-precision matrix
// cores" scheme (cf. the BF16x9 mode of vendor BLAS libraries), selected by cbd_set_option("f32_split", 1); accuracy against
// the exact-fp32 policy is measured in tests/test_gpu_bf16.py.  Fragment 3q + p = plane p of k-step q (18 KB tile).
struct OpsBf16x3 {
  static constexpr bool EXACT_F32 = false;
  static constexpr bool NODE_PROJ = true;
  using Frag = bf16x8;
  static constexpr int NFRAG = 3 * (KDIM / 16);            // 18
  static constexpr int TILE_FRAGS = NFRAG * 64;            // 1152 fragments of 16 B per tile
  struct Act { bf16x8 v[NFRAG]; };
  static __device__ __forceinline__ void split(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    float r = x - (float)h;      // exact
    m = (__bf16)r;
    r -= (float)m;               // exact
    l = (__bf16)r;
  }
  static __device__ __forceinline__ void set_in(Act& B, int seg, int q, f32x4 x) {
    const int k = 3 * (2 * seg + (q >> 1)), o = 4 * (q & 1);
    const float xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      __bf16 h, m, l;
      split(xs[c], h, m, l);
      B.v[k][o + c] = h; B.v[k + 1][o + c] = m; B.v[k + 2][o + c] = l;
    }
  }
  static __device__ __forceinline__ void set_hidden(Act& h, int m, const f32x16& acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = 3 * (2 * m + (r >> 3));
      __bf16 hh, mm, ll;
      split(relu1(acc[r]), hh, mm, ll);
      h.v[k][r & 7] = hh; h.v[k + 1][r & 7] = mm; h.v[k + 2][r & 7] = ll;
    }
  }
  static __device__ __forceinline__ void gemm(Frag (&a)[NFRAG], const Frag* __restrict__ next, const Act& B, f32x16& acc) {
#pragma unroll
    for (int q = 0; q < NFRAG / 3; ++q) {
      const int k = 3 * q;   // smallest terms first
      const Frag* p = next + k * 64;
      pin(p);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 2], B.v[k + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 0], B.v[k + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 1], B.v[k + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 1], B.v[k + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 0], B.v[k + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 0], B.v[k + 0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a[k + 0] = p[0];
      a[k + 1] = p[64];
      a[k + 2] = p[128];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // first Linear: K = 32 = 2 k-steps x 3 planes = 6 fragments per tile, 3 tiles = the whole register tile (see OpsF32::gemm_first)
  // ---- on a wave-uniform tile base (see OpsF32::gemm_u): fragment f is base[f / 4][lane + (f % 4) * 64].  The three planes of a k-step
  //      are refilled right behind its six MFMAs (192 cycles of matrix-pipe time ahead of an L2 round trip; no VALU in the chain).
  static __device__ __forceinline__ void gemm_u(Frag (&a)[NFRAG], GPtr<Frag> next, int lane, const Act& B, f32x16& acc) {
    GPtr<Frag> pb[5] = {next, next + 4 * 64, next + 8 * 64, next + 12 * 64, next + 16 * 64};
#pragma unroll
    for (int i = 0; i < 5; ++i) pin_s(pb[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NFRAG / 3; ++q) {
      const int k = 3 * q;   // smallest terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 2], B.v[k + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 0], B.v[k + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 1], B.v[k + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 1], B.v[k + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 0], B.v[k + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k + 0], B.v[k + 0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 3; ++i) a[k + i] = pb[(k + i) / 4][lane + ((k + i) % 4) * 64];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  static __device__ __forceinline__ void load_first_u(Frag (&a)[NFRAG], GPtr<Frag> gp, int lane) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int s = 0; s < NFRAG / 3; ++s) a[(NFRAG / 3) * m + s] = gp[(size_t)m * TILE_FRAGS + s * 64 + lane];
  }
  template <int M0, int M1>   // first-Linear tiles [M0, M1) only
  static __device__ __forceinline__ void load_first_part(Frag (&a)[NFRAG], GPtr<Frag> gp, int lane) {
#pragma unroll
    for (int m = M0; m < M1; ++m)
#pragma unroll
      for (int s = 0; s < NFRAG / 3; ++s) a[(NFRAG / 3) * m + s] = gp[(size_t)m * TILE_FRAGS + s * 64 + lane];
  }
  template <int M>
  static __device__ __forceinline__ void gemm_first_u(Frag (&a)[NFRAG], GPtr<Frag> next, int lane, const Act& B, f32x16& acc) {
    constexpr int base = (NFRAG / 3) * M;          // fragments 6M .. 6M+5 of the first second-Linear tile
    GPtr<Frag> pm0 = next + base * 64, pm1 = next + (base + 3) * 64;
    pin_s(pm0); pin_s(pm1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < (NFRAG / 3) / 3; ++q) {
      const int k = 3 * q, ka = base + 3 * q;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 2], B.v[k + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 0], B.v[k + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 1], B.v[k + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 1], B.v[k + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 0], B.v[k + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 0], B.v[k + 0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      GPtr<Frag> p = q == 0 ? pm0 : pm1;
      a[ka + 0] = p[lane];
      a[ka + 1] = p[lane + 64];
      a[ka + 2] = p[lane + 128];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  static constexpr int FIRST_FRAGS = NFRAG / 3;            // 6
  static __device__ __forceinline__ void load_first(Frag (&a)[NFRAG], const Frag* __restrict__ gp) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int s = 0; s < FIRST_FRAGS; ++s) a[FIRST_FRAGS * m + s] = gp[(size_t)m * TILE_FRAGS + s * 64];
  }
  template <int M>
  static __device__ __forceinline__ void gemm_first(Frag (&a)[NFRAG], const Frag* __restrict__ next, const Act& B, f32x16& acc) {
#pragma unroll
    for (int q = 0; q < FIRST_FRAGS / 3; ++q) {
      constexpr int base = FIRST_FRAGS * M;
      const int k = 3 * q, ka = base + 3 * q;
      const Frag* p = next + ka * 64;
      pin(p);
      __builtin_amdgcn_sched_barrier(0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 2], B.v[k + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 0], B.v[k + 2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 1], B.v[k + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 1], B.v[k + 0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 0], B.v[k + 1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ka + 0], B.v[k + 0], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      a[ka + 0] = p[0];
      a[ka + 1] = p[64];
      a[ka + 2] = p[128];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
};

template <class Ops>
__device__ __forceinline__ void gemm_tile(typename Ops::Frag (&a)[Ops::NFRAG], const typename Ops::Frag* __restrict__ next,
                                          const float* __restrict__ bias_l, const typename Ops::Act& B, f32x16& acc, int hf) {
  const f32x4* bp = reinterpret_cast<const f32x4*>(bias_l);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 b = bp[2 * q + hf];
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
  Ops::gemm(a, next, B, acc);
}

template <class Ops>
__device__ __forceinline__ void gemm_tile_u(typename Ops::Frag (&a)[Ops::NFRAG], GPtr<typename Ops::Frag> next, int lane,
                                            const float* __restrict__ bias_l, const typename Ops::Act& B, f32x16& acc, int hf) {
  const f32x4* bp = reinterpret_cast<const f32x4*>(bias_l);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    f32x4 b = bp[2 * q + hf];
    acc[4 * q + 0] = b.x; acc[4 * q + 1] = b.y; acc[4 * q + 2] = b.z; acc[4 * q + 3] = b.w;
  }
  Ops::gemm_u(a, next, lane, B, acc);
}

// "mid" evaluators: value of the CG intermediate with index i for edge j (xc = &xT[0][j], column stride 32).
// Index spaces follow reference tensor_layers.py:72-85 (concatenation order of out_dict[...] lists).
template <int IN>
__device__ __forceinline__ float mid0e(const float* xc, int i, const float (&v)[3]) {
  if (i < NS) return xc[i * 32];
  if (IN >= 1) {
    const float* p = xc + (COL_1O + 3 * (i - NS)) * 32;
    return p[0] * v[0] + p[32] * v[1] + p[64] * v[2];
  }
  return 0.f;
}

__device__ __forceinline__ void cross3(const float* p, const float (&v)[3], float (&m)[3]) {
  const float a0 = p[0], a1 = p[32], a2 = p[64];
  m[0] = a1 * v[2] - a2 * v[1];
  m[1] = a2 * v[0] - a0 * v[2];
  m[2] = a0 * v[1] - a1 * v[0];
}

template <int IN>
__device__ __forceinline__ void mid1o(const float* xc, int i, const float (&v)[3], float (&m)[3]) {
  constexpr ConvShape S = conv_shape(IN, 3);
  if (i < NS) {
    const float s = xc[i * 32];
    m[0] = s * v[0]; m[1] = s * v[1]; m[2] = s * v[2];
  } else if (i < NS + S.n1o) {
    const float* p = xc + (COL_1O + 3 * (i - NS)) * 32;
    m[0] = p[0]; m[1] = p[32]; m[2] = p[64];
  } else if (i < S.fan1o) {
    cross3(xc + (COL_1E + 3 * (i - NS - S.n1o)) * 32, v, m);
  } else {
    m[0] = m[1] = m[2] = 0.f;
  }
}

template <int IN>
__device__ __forceinline__ void mid1e(const float* xc, int i, const float (&v)[3], float (&m)[3]) {
  constexpr ConvShape S = conv_shape(IN, 3);
  if (i < S.n1o) {
    cross3(xc + (COL_1O + 3 * i) * 32, v, m);
  } else if (i < S.n1o + S.n1e) {
    const float* p = xc + (COL_1E + 3 * (i - S.n1o)) * 32;
    m[0] = p[0]; m[1] = p[32]; m[2] = p[64];
  } else if (i < S.fan1e) {
    const float s = xc[(COL_0O + (i - S.n1o - S.n1e)) * 32];
    m[0] = s * v[0]; m[1] = s * v[1]; m[2] = s * v[2];
  } else {
    m[0] = m[1] = m[2] = 0.f;
  }
}

template <int IN>
__device__ __forceinline__ float mid0o(const float* xc, int i, const float (&v)[3]) {
  constexpr ConvShape S = conv_shape(IN, 3);
  if (i < S.n1e) {
    const float* p = xc + (COL_1E + 3 * i) * 32;
    return p[0] * v[0] + p[32] * v[1] + p[64] * v[2];
  }
  if (i < S.fan0o) return xc[(COL_0O + (i - S.n1e)) * 32];
  return 0.f;
}

constexpr int XT_FLOATS = NODE_STRIDE * 32;              // per-wave transposed copy of the gathered rows, later the message tile
static_assert(XT_FLOATS >= NODE_DIM * OUT_STRIDE, "the message tile re-uses the gathered-row tile");
__host__ __device__ constexpr int conv_lds_floats(int ntiles) { return ntiles * 32 + XT_FLOATS + 32; }

}  // namespace cbd
