// Device code shared by the two bf16 tensor-product kernels (tp_conv_bf16.hip: one wave per workgroup, weight tiles streamed from L2;
// tp_conv_bf16p.hip: persistent workgroups with a 0e slice's weight tiles resident in LDS): fragment types, operand packing, the MFMA
// chains with in-place fragment refill.  See tp_conv_bf16.hip for the design notes.
#pragma once
#include <type_traits>

#include "kernels.h"
#include "tp_conv_dev.h"

namespace cbd {

constexpr int V2_NFRAG = 6;                      // 6 k-steps of 16 = the 96 inputs; the bias enters as the C operand of the first pair
constexpr int V2_TILE_FRAGS = V2_NFRAG * 64;     // 16-byte fragments per tile
constexpr int V2_SUB_FLOATS = NODE_DIM * OUT_STRIDE;   // 74 x 34 floats per 32-edge sub-tile (>= 76 x 32 of the gather image)
static_assert(V2_SUB_FLOATS >= 76 * 32, "gather image must fit the message tile");

struct Act6 { bf16x8 v[V2_NFRAG]; };

// Weight tiles and bias rows are read with global loads whose base address is wave-uniform (an SGPR pair, advanced per tile on the
// scalar unit) plus the constant per-lane offset in one VGPR plus an immediate: no vector address arithmetic per tile at all.
typedef GPtr<bf16x8> GFrag;
typedef GPtr<float> GBias;

__device__ __forceinline__ void v2_set_in(Act6& B, int seg, int q, f32x4 x) {
  const int k = 2 * seg + (q >> 1), o = 4 * (q & 1);
  B.v[k][o + 0] = (__bf16)x.x; B.v[k][o + 1] = (__bf16)x.y; B.v[k][o + 2] = (__bf16)x.z; B.v[k][o + 3] = (__bf16)x.w;
}
__device__ __forceinline__ void v2_set_hidden(Act6& h, int m, const f32x16& acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r) h.v[2 * m + (r >> 3)][r & 7] = (__bf16)relu1(acc[r]);
}

// acc_s = bias + A_tile * B_s (s = 0, 1).  The bias (fp32, one value per weight row) sits in 16 registers in the accumulator layout
// and is the C operand of the first MFMA pair (D != C): no bias k-step, no accumulator initialisation.
// The A fragments and the bias registers are refilled in place with the NEXT tile's data.  Two rules keep the in-flight MFMA operand
// hazard of tp_conv_dev.h out (a VALU write, or a returning load, into a register that an issued MFMA has not read yet):
//   * the whole chain contains NO VALU instruction: the base addresses are computed and pinned before the first MFMA and every load
//     uses an immediate offset (fragments 0..3: base + q KB, fragments 4..5: base + 4 KB + (q - 4) KB; the field holds < 4 KB), so a
//     register that is momentarily dead cannot be handed to address arithmetic (measured, round 2: the same delayed schedule with
//     per-fragment address computation between the pairs gave run-to-run differences in 8 of 9 repeats of tools/bf16_repeat.py);
//   * fragment q-1 is re-loaded after the MFMA pair of fragment q has been issued -- one pair late, so that even a load that hits in
//     L1 (~120 cycles) lands after the pair that read the register has started; the bias follows the second pair, the last fragment
//     its own pair directly.
template <int DIAG = 0>
__device__ __forceinline__ void v2_gemm(bf16x8 (&a)[V2_NFRAG], f32x16& cb, GFrag next, GPtr<float> next_bias, float& raw_next, int lane,
                                        int lane4hf, const Act6& B0, const Act6& B1, f32x16& acc0, f32x16& acc1) {
  GFrag pa = next;            // uniform: tile base (fragments 0..3: immediate offsets 0..3 KB)
  GFrag pb = next + 4 * 64;   // fragments 4..5 (the immediate field holds < 4 KB)
  GPtr<float> pc = next_bias; // uniform: the next tile's 32 bias floats
  pin_s(pa); pin_s(pb); pin_s(pc);
  float raw_new = 0.f;   // bias of the tile AFTER the next one (`next_bias`), spread one chain from now: two tiles of memory latency
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) {
    if (q == 0) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B0.v[q], cb, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B1.v[q], cb, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B0.v[q], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B1.v[q], acc1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // keep the bias registers a live 16-register tuple of their own: otherwise hipcc lets an accumulator take them over after the first
    // pair, loads the next bias somewhere else and copies it back with 16 v_mov behind a vmcnt(0) at the end of every tile
    if (q == 0) asm volatile("" : "+v"(cb));
    // the next tile's 32 bias floats: ONE dword per lane (lane l gets float l & 31: 256 B through the vector-memory return path
    // instead of the 4 KB of four broadcast dwordx4 loads -- timing-only diagnostics put those at 11 % of the kernel)
    if (q == 0 && !(DIAG & 8)) raw_new = pc[lane & 31];
    if (q > 0 && !(DIAG & 16)) a[q - 1] = q - 1 < 4 ? pa[lane + (q - 1) * 64] : pb[lane + (q - 5) * 64];
    if (q == V2_NFRAG - 1 && !(DIAG & 16)) a[q] = pb[lane + (q - 4) * 64];
    __builtin_amdgcn_sched_barrier(0);
  }
  // ... spread into the accumulator layout through the LDS crossbar (ds_bpermute_b32, no LDS memory): register r of lane half hf is
  // weight row (r & 3) + 8 (r >> 2) + 4 hf.  Issued behind the last pair: pair 0, which read cb as its C operand, has executed long ago,
  // and the tile's CG epilogue covers the crossbar latency.
  if constexpr (!(DIAG & 8)) {
    // inline asm: the builtin takes no offset, and hipcc then keeps 16 address registers (spills); with the instruction's offset
    // field one address register (byte address of lane 4 hf) serves all 16.  The results are NOT tracked by the compiler's waitcnt
    // insertion: bias_ready() (s_waitcnt lgkmcnt(0)) closes the tile's epilogue before cb is read again.
    float t[16];
#define CBD_BP4(R, O0, O1, O2, O3)                                                                                      \
    asm volatile("ds_bpermute_b32 %0, %4, %5 offset:" #O0 "\n\tds_bpermute_b32 %1, %4, %5 offset:" #O1                \
                 "\n\tds_bpermute_b32 %2, %4, %5 offset:" #O2 "\n\tds_bpermute_b32 %3, %4, %5 offset:" #O3            \
                 : "=&v"(t[R]), "=&v"(t[R + 1]), "=&v"(t[R + 2]), "=&v"(t[R + 3]) : "v"(lane4hf), "v"(raw_next))
    CBD_BP4(0, 0, 4, 8, 12);
    CBD_BP4(4, 32, 36, 40, 44);
    CBD_BP4(8, 64, 68, 72, 76);
    CBD_BP4(12, 96, 100, 104, 108);
#undef CBD_BP4
#pragma unroll
    for (int r = 0; r < 16; ++r) cb[r] = t[r];
    raw_next = raw_new;
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Software-pipelined form of v2_gemm for the 0e block (two accumulator sets): `epi(q)` -- a sixth of the PREVIOUS tile's CG epilogue,
// which reads the other accumulator set -- is issued behind MFMA pair q, so the wave's own VALU work runs in the shadow of its own
// MFMAs (each pair keeps the matrix pipe busy for >= 64 cycles; a chunk is 4-6 FMAs).  The epilogue only writes long-lived output
// registers; fragment q - 1, which is dead between its pair and its re-load one pair later, is kept allocated through a fake use so
// that no VALU temporary can be placed in a register an issued MFMA has not read yet (tp_conv_dev.h).  The next tile's bias is spread
// behind pair 3 (the raw dword was requested behind pair 0; its lines are hot in L1) and bias_ready() precedes the next chain.
template <int DIAG = 0, bool BIAS = true, class Epi>
__device__ __forceinline__ void v2_gemm_p(bf16x8 (&a)[V2_NFRAG], f32x16& cb, GFrag next, GPtr<float> next_bias, float& raw_next, int lane,
                                          int lane4hf, const Act6& B0, const Act6& B1, f32x16& acc0, f32x16& acc1, Epi epi) {
  GFrag pa = next;
  GFrag pb = next + 4 * 64;
  GPtr<float> pc = next_bias;
  pin_s(pa); pin_s(pb); pin_s(pc);
  float raw_new = 0.f;   // bias of the tile AFTER the next one (`next_bias`), spread one chain from now: two tiles of memory latency
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) {
    if (q == 0) {
      if constexpr (BIAS) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B0.v[q], cb, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B1.v[q], cb, 0, 0, 0);
      } else {   // 0e block: no per-tile bias at all (its contribution enters once per block, see the kernel): C = 0 (inline constant)
        const f32x16 zero = {};
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B0.v[q], zero, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B1.v[q], zero, 0, 0, 0);
      }
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B0.v[q], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], B1.v[q], acc1, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (q == 0 && BIAS) asm volatile("" : "+v"(cb));
    if (q == 0 && BIAS && !(DIAG & 8)) raw_new = pc[lane & 31];
    if (q > 0 && !(DIAG & 16)) a[q - 1] = q - 1 < 4 ? pa[lane + (q - 1) * 64] : pb[lane + (q - 5) * 64];
    if (q == V2_NFRAG - 1 && !(DIAG & 16)) a[q] = pb[lane + (q - 4) * 64];
    __builtin_amdgcn_sched_barrier(0);
    if (q == 3 && BIAS && !(DIAG & 8)) {
      float t[16];
#define CBD_BP4(R, O0, O1, O2, O3)                                                                                      \
      asm volatile("ds_bpermute_b32 %0, %4, %5 offset:" #O0 "\n\tds_bpermute_b32 %1, %4, %5 offset:" #O1                \
                   "\n\tds_bpermute_b32 %2, %4, %5 offset:" #O2 "\n\tds_bpermute_b32 %3, %4, %5 offset:" #O3            \
                   : "=&v"(t[R]), "=&v"(t[R + 1]), "=&v"(t[R + 2]), "=&v"(t[R + 3]) : "v"(lane4hf), "v"(raw_next))
      CBD_BP4(0, 0, 4, 8, 12);
      CBD_BP4(4, 32, 36, 40, 44);
      CBD_BP4(8, 64, 68, 72, 76);
      CBD_BP4(12, 96, 100, 104, 108);
#undef CBD_BP4
#pragma unroll
      for (int r = 0; r < 16; ++r) cb[r] = t[r];
    }
    epi(q);
    if (q < V2_NFRAG - 1) asm volatile("" ::"v"(a[q]));   // fragment q stays allocated until its re-load behind pair q + 1
    __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (BIAS && !(DIAG & 8)) raw_next = raw_new;
}

// closes a tile: the ds_bpermute results of v2_gemm (the next tile's bias registers) have landed
__device__ __forceinline__ void bias_ready(f32x16& cb) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cb)::"memory"); }


}  // namespace cbd
