// bf16-operand tensor-product message passing, third generation: the weight stream SHARED by the eight waves of a workgroup
// (BASELINE.json configs[3]; cbd_set_option("bf16", 1); CBD_BF16_KERNEL=0 selects the second generation, tp_conv_bf16.hip).
//
// Why: timing-only diagnostics of the second-generation kernel (profiles/r03_*): without its weight loads it
// runs at 0.63 of the bf16 peak, with them at 0.42 -- every wave pulls its own copy of every 6 KB tile through the CU's vector
// memory path, and at 64 edges per wave that path needs the same 64 B/clk per CU as the matrix cores' operand rate: the L1 return
// path, not L2 or HBM, is the second saturated resource.  Here a tile enters the CU ONCE per workgroup:
//   * workgroup = 8 waves = 512 edges of ONE edge group; every wave brings one 1 KB fragment of a tile (waves 6, 7 repeat fragments 0, 1) from L2 straight into LDS
//     (global_load_lds_dwordx4, no VGPRs), four tiles ahead, into a ring of four 6 KB slots;
//   * every wave re-fills its six fragment registers from the ring with ds_read_b128, in place and one MFMA pair late exactly like
//     the second generation did from global memory -- the LDS read path has 4x the bandwidth of the L1 return path and was idle;
//   * ONE s_barrier per tile: it says "tile p + 1 has landed" (every loader waited for its own DMA before arriving) and "slot p is
//     free" (every wave finished the chain that read tile p).  The 0e block's software pipeline (the previous tile's epilogue between
//     the MFMA pairs of the current chain) keeps a wave's own VALU work in its own MFMA shadow, so the lockstep the barrier imposes on
//     the two residents of a SIMD costs little there;
//   * LDS: the ring needs 24 KB, so the per-wave gather image shrinks from 19.5 KB to 15 KB: the 32 scalar columns are kept as bf16
//     (they enter the first Linear as bf16 anyway), the vector columns stay fp32; the message tiles of the two 32-edge sub-tiles are
//     written and reduced one after the other in the same 15 KB.  8 x 15.25 KB + 24 KB = 146 KB of the CU's 160 KB.
// Everything else (math, tile order, bias handling, deterministic run-length reduction, stream format) is the second generation's.
#include <cstdlib>
#include <type_traits>

#include "kernels.h"
#include "tp_conv_dev.h"

namespace cbd {
namespace shared_w {

constexpr int V2_NFRAG = 6;                      // 6 k-steps of 16 = the 96 inputs; the bias enters as the C operand of the first pair
constexpr int V2_TILE_FRAGS = V2_NFRAG * 64;     // 16-byte fragments per tile
constexpr int V2_SUB_FLOATS = NODE_DIM * OUT_STRIDE;   // 74 x 33 floats per 32-edge sub-tile (>= 76 x 32 of the gather image)
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
__device__ __forceinline__ void v2_gemm(bf16x8 (&a)[V2_NFRAG], f32x16& cb, const bf16x8* lnext, GPtr<float> next_bias, float& raw_next, int lane31x4,
                                        int lane4hf, const Act6& B0, const Act6& B1, f32x16& acc0, f32x16& acc1) {
  const bf16x8* pl = lnext;   // LDS: the next tile in the ring, this lane's 16 bytes of fragment 0 (fragment q: + q KB)
  pin(pl);
  GPtr<float> pc = next_bias; // uniform: the next tile's 32 bias floats
  pin_s(pc);
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
    if (q == 0 && !(DIAG & 8)) asm volatile("global_load_dword %0, %1, %2" : "=v"(raw_new) : "v"(lane31x4), "s"(pc) : "memory");
    if (q > 0 && !(DIAG & 16)) a[q - 1] = pl[(q - 1) * 64];
    if (q == V2_NFRAG - 1 && !(DIAG & 16)) a[q] = pl[q * 64];
    __builtin_amdgcn_sched_barrier(0);
  }
  // ... spread into the accumulator layout through the LDS crossbar (ds_bpermute_b32, no LDS memory): register r of lane half hf is
  // weight row (r & 3) + 8 (r >> 2) + 4 hf.  Issued behind the last pair: pair 0, which read cb as its C operand, has executed long ago,
  // and the tile's CG epilogue covers the crossbar latency.
  if constexpr (!(DIAG & 8)) {
    // inline asm: the builtin takes no offset, and hipcc then keeps 16 address registers (spills); with the instruction's offset
    // field one address register (byte address of lane 4 hf) serves all 16.  The results are NOT tracked by the compiler's waitcnt
    // insertion: bias_ready() (s_waitcnt lgkmcnt(0)) closes the tile's epilogue before cb is read again.
    asm volatile("s_waitcnt vmcnt(3)" : "+v"(raw_next));   // younger: this step's two DMAs and this chain's own bias dword
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
__device__ __forceinline__ void v2_gemm_p(bf16x8 (&a)[V2_NFRAG], f32x16& cb, const bf16x8* lnext, GPtr<float> next_bias, float& raw_next, int lane31x4,
                                          int lane4hf, const Act6& B0, const Act6& B1, f32x16& acc0, f32x16& acc1, Epi epi) {
  const bf16x8* pl = lnext;
  pin(pl);
  GPtr<float> pc = next_bias;
  pin_s(pc);
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
    if (q == 0 && BIAS && !(DIAG & 8)) asm volatile("global_load_dword %0, %1, %2" : "=v"(raw_new) : "v"(lane31x4), "s"(pc) : "memory");
    if (q > 0 && !(DIAG & 16)) a[q - 1] = pl[(q - 1) * 64];
    if (q == V2_NFRAG - 1 && !(DIAG & 16)) a[q] = pl[q * 64];
    __builtin_amdgcn_sched_barrier(0);
    if (q == 3 && BIAS && !(DIAG & 8)) {
      asm volatile("s_waitcnt vmcnt(3)" : "+v"(raw_next));   // younger: this step's two DMAs and this chain's own bias dword
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

// DIAG (timing only, WRONG results; CBD_BF16_DIAG=n): 1 = every tile re-reads weight tile 0 (the weight stream becomes L1-resident),
// 2 = no CG epilogue (the accumulators are only summed up), 3 = both; 4 = correct results + phase stamps (tools/conv_clock.py bf16);
// 8 = the bias registers are never re-loaded, 9 = 8 + 1, 16 = no weight or bias re-loads at all (the first tile's registers serve every
// tile: the kernel without its weight stream), 24 = 16 + 8, 32 = the bias bpermutes are not waited for
constexpr int SW_WAVES = 4;                               // waves per workgroup: 256 edges of one group share every weight tile; TWO
                                                          // workgroups per CU (one wave of each per SIMD), so the residents of a SIMD
                                                          // are NOT in lockstep and a workgroup's prologue overlaps the other's tiles
constexpr int SW_RING = 3;                                // ring slots of one 6 KB tile each
constexpr int SW_VEC_FLOATS = (76 - NS) * 32;             // fp32 part of a sub-tile's gather image: columns 32..75
constexpr int SW_SUB_BYTES = SW_VEC_FLOATS * 4 + NS * 32 * 2;   // + the 32 scalar columns as bf16: 7680 B
constexpr int SW_WAVE_BYTES = 2 * SW_SUB_BYTES + 256;     // two sub-tiles + srcl[2][32]
static_assert(2 * SW_SUB_BYTES >= NODE_DIM * OUT_STRIDE * 4, "one message tile must fit a wave's two gather images");
constexpr int SW_LDS_BYTES = SW_RING * V2_TILE_FRAGS * 16 + SW_WAVES * SW_WAVE_BYTES;

// bf16 bits -> fp32
__device__ __forceinline__ float bf_up(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

// One 1 KB fragment global -> LDS without registers: lane l's 12 bytes land at lds_dst + 12 l (global_load_lds_dwordx3: 4 waves x 2 x 768 B = one 6 KB tile; M0 carries the
// wave-uniform LDS byte address and is restored -- hipcc reserves it).  NOT counted by hipcc's s_waitcnt bookkeeping: the kernel waits
// with explicit vmcnt counts (cdna_hip_programming.md section 7).
__device__ __forceinline__ void glds12(const void* gsrc_lane, unsigned lds_dst) {   // lane l's 12 bytes land at lds_dst + 12 l
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
}

template <int IN, int OUT, int DIAG = 0>
__global__ __launch_bounds__(64 * SW_WAVES, 2) void tp_conv64s_kernel(ConvArgs args) {
  constexpr ConvShape S = conv_shape(IN, OUT);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int j = lane & 31, hf = lane >> 5;
  const int lane4hf = 16 * hf;   // byte address (4 x lane) of lane 4 hf: base of the bias bpermutes (v2_gemm)
  bf16x8* const ring = reinterpret_cast<bf16x8*>(lds_raw);                         // [SW_RING][6 fragments][64 lanes] x 16 B
  unsigned char* const mine = lds_raw + SW_RING * V2_TILE_FRAGS * 16 + wave * SW_WAVE_BYTES;
  // sub-tile s: fp32 [44 cols][32] (columns 32..75) then bf16 [32 cols][32] (columns 0..31).  xc0 / xc1 are biased so that the usual
  // `xc[col * 32]` addressing works for the VECTOR columns (col >= 32); the scalar columns are read through xs0 / xs1.
  float* const xv0 = reinterpret_cast<float*>(mine);
  float* const xv1 = reinterpret_cast<float*>(mine + SW_SUB_BYTES);
  unsigned short* const xsb0 = reinterpret_cast<unsigned short*>(mine + SW_VEC_FLOATS * 4);
  unsigned short* const xsb1 = reinterpret_cast<unsigned short*>(mine + SW_SUB_BYTES + SW_VEC_FLOATS * 4);
  float* const msg = reinterpret_cast<float*>(mine);                              // message tile of ONE sub-tile at a time, [col][33]
  int* const srcl = reinterpret_cast<int*>(mine + 2 * SW_SUB_BYTES);               // [2][32]

  // ---- which group / edge range?  A workgroup owns 512 consecutive edges of one group; wave w the 64 from 64 w on.  Waves (and
  //      lanes) past the end of the group work on its last edge and are dropped through src = -1: they must still take part in the
  //      workgroup's barriers.
  int grp = -1, e0 = 0, cnt = 0, tile_local = 0;
  {
    int wg_in_group = 0;
    if (!find_group(args, blockIdx.x, lane, 64 * SW_WAVES, grp, wg_in_group, cnt)) return;   // workgroup-uniform
    e0 = (wg_in_group * SW_WAVES + wave) * 64;
    tile_local = 2 * (wg_in_group * SW_WAVES + wave);
  }
  const ConvGroup G = args.g[grp];
  unsigned long long st_t0 = 0, st_r0 = 0, st_t1 = 0, st_t2 = 0, st_t3 = 0, st_0e = 0;
  if constexpr (DIAG == 4) { st_t0 = stamp(); st_r0 = __builtin_amdgcn_s_memrealtime(); }

  // ---- the tile sequence of this workgroup and the start of the ring
  const GFrag gp = (GFrag)reinterpret_cast<const bf16x8*>(G.wstream);   // tile T fragment q of lane l: gp[T * V2_TILE_FRAGS + q * 64 + l]
  const int i_lo = G.i0e_lo, i_hi = G.i0e_hi;
  const bool vec_on = G.vec_on != 0;
  const int T_vec = 3 + S.t0e;
  const int n0e = i_hi > i_lo ? i_hi - i_lo : 0;
  // stream index of the tile at position p of the sequence (first Linear, the 0e tiles of this slice, the vector blocks, then the
  // zero tile for good): selects only
  auto seq = [&](int p) {
    const int v = vec_on ? T_vec + (p - 3 - n0e) : S.ntiles;
    const int t = p < 3 ? p : (p - 3 < n0e ? 3 + i_lo + (p - 3) : v);
    return t < S.ntiles ? t : S.ntiles;
  };
  int pos = 0;                       // position of the tile whose chain runs next
  // EVERY wave brings a quarter of each tile: two DMAs of 64 x 12 B (bytes [1536 w, 1536 w + 1536) of the tile) -- all four waves run
  // the same instruction stream with the same vmcnt arithmetic, and the tile loops stay free of branches
  const unsigned ring_lds = (unsigned)reinterpret_cast<size_t>(lds_raw);   // LDS byte address of the ring: the low 32 bits of a generic LDS pointer
  auto issue_tile = [&](int p) {     // this wave's quarter of tile seq(p) -> slot p % 3 (asynchronous, no registers)
    const unsigned char* src = reinterpret_cast<const unsigned char*>(G.wstream) + (size_t)seq(p) * (V2_TILE_FRAGS * 16) + wave * 1536 + lane * 12;
    const unsigned dst = ring_lds + (p % SW_RING) * (V2_TILE_FRAGS * 16) + wave * 1536;
    glds12(src, dst);
    glds12(src + 768, dst + 768);
  };
#pragma unroll
  for (int p = 0; p < SW_RING; ++p) issue_tile(p);
  // fp32 bias rows behind the (ntiles + 1) tiles: [ntiles + 1][32]; this lane half's float4s are 2q' + hf
  const GBias gbias = (GBias)reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(G.wstream) + (size_t)(S.ntiles + 1) * V2_TILE_FRAGS);   // uniform
  f32x16 cb;
  {
    const GPtr<f32x4> gb4 = (GPtr<f32x4>)gbias;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const f32x4 b = gb4[hf + 2 * qq];
      cb[4 * qq + 0] = b.x; cb[4 * qq + 1] = b.y; cb[4 * qq + 2] = b.z; cb[4 * qq + 3] = b.w;
    }
  }

  // ---- gather both sub-tiles.  Lanes past the end of the group read the group's last edge (unconditional loads) and are dropped
  //      at the end through src = -1.
  Act6 Bx0, Bx1;
  float v0[3], v1[3];
  // Two rounds of memory latency instead of four: the edge indices of BOTH sub-tiles first, then every gather that depends on them in
  // one batch (attributes, the two 32-column node segments, sub-tile 0's full destination row: 136 registers in flight), sub-tile 1's
  // row while the first is transposed into LDS.  (In source order per sub-tile the compiler waited for each sub-tile's indices and
  // gathers in turn: 21.5 k cycles of a 113 k lifetime, in-kernel stamps.)
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
  f32x4 ta[2][4], ts[2][4], td[2][4], tr[10];
  const f32x4* pr[2];
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    const f32x4* pa = reinterpret_cast<const f32x4*>(G.attr + (size_t)aidx[sub] * 32 + 16 * hf);
    const f32x4* ps = reinterpret_cast<const f32x4*>(G.node_in + (size_t)src_r[sub] * NODE_STRIDE + 16 * hf);
    const f32x4* pd = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dstn[sub] * NODE_STRIDE + 16 * hf);
    // full destination row; lane half hf takes cols 40hf .. 40hf+39 (cols 76..79 are the row's padding: loaded, not stored)
    pr[sub] = reinterpret_cast<const f32x4*>(G.node_in + (size_t)dstn[sub] * NODE_STRIDE + 40 * hf);
#pragma unroll
    for (int q = 0; q < 4; ++q) { ta[sub][q] = pa[q]; ts[sub][q] = ps[q]; td[sub][q] = pd[q]; }
  }
#pragma unroll
  for (int q = 0; q < 10; ++q) tr[q] = pr[0][q];
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
#pragma unroll
  for (int sub = 0; sub < 2; ++sub) {
    float* xv = sub ? xv1 : xv0;
    unsigned short* xsb = sub ? xsb1 : xsb0;
#pragma unroll
    for (int q = 0; q < 10; ++q) {   // transposed LDS copy: lane half hf holds columns 40hf + 4q .. + 3 of edge j
      const f32x4 r = tr[q];
      if (sub == 0) tr[q] = pr[1][q];
      const int c0 = 40 * hf + 4 * q;          // runtime through hf only: q < 8 of the lower half are the scalar columns
      const float rv[4] = {r.x, r.y, r.z, r.w};
      if (c0 < 76) {
        if (c0 < NS) {
#pragma unroll
          for (int k = 0; k < 4; ++k) xsb[(c0 + k) * 32 + j] = (unsigned short)(__builtin_bit_cast(unsigned, (float)(__bf16)rv[k]) >> 16);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) xv[(c0 + k - NS) * 32 + j] = rv[k];
        }
      }
    }
  }
  // the first four tiles have landed (every loader waits for its own DMAs, then the workgroup meets); tile 0 -> registers
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  bf16x8 a[V2_NFRAG];
#pragma unroll
  for (int q = 0; q < V2_NFRAG; ++q) a[q] = ring[q * 64 + lane];
  const int lane31x4 = 4 * (lane & 31);
  if constexpr (DIAG == 4) st_t1 = stamp();
  // Start of the step that runs the chain of tile `pos`: tile pos + 1 must be complete in the ring (its fragments are read during this
  // chain) and the slot of tile pos is free once every wave has left the chain that read it -> ONE barrier, then the DMAs of tile pos + 3
  // into that slot.  A step issues two DMAs (tile pos + 3) and, outside the 0e block, one bias dword behind them; what may still be in
  // flight at the next step's start is exactly that step's operations -- vmcnt(3) if it issued a bias dword (RAWPREV), else vmcnt(2):
  // the DMAs of tile pos + 1, issued two steps ago, have landed.  lgkmcnt(0): this wave's own reads of the slot are complete.
#define SW_SYNC(RAWPREV)                                                            \
  {                                                                                 \
    if constexpr (RAWPREV) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                                                   \
    issue_tile(pos + SW_RING);                                                      \
  }
#define SW_NEXT() (ring + ((pos + 1) % SW_RING) * V2_TILE_FRAGS + lane)

  int T = 0;
  f32x16 acc0, acc1;
  Act6 h0, h1;
  // The bias rows are requested TWO tiles ahead (every chain names the tile after its successor: NEXT2).  This wave's sequence: first
  // Linear 0..2, the 0e tiles [3 + i_lo, 3 + i_hi), the vector blocks if vec_on, then the zero tile S.ntiles.  All selects, no branches.
  const int past0e = vec_on ? T_vec : S.ntiles;                                  // first tile behind the 0e block
  const int first0e = i_lo < i_hi ? 3 + i_lo : past0e;                           // tile behind the first Linear
  const int past0e_1 = vec_on ? (T_vec + 1 < S.ntiles ? T_vec + 1 : S.ntiles) : S.ntiles;   // ... and the one behind `past0e`
  const int second0e = i_lo < i_hi ? (i_lo + 1 < i_hi ? 4 + i_lo : past0e) : past0e_1;     // tile behind `first0e`
  float raw_next = gbias[32 + (lane & 31)];   // bias of tile 1
#define V2_TILE(RAWPREV, BA, BB, NEXT, NEXT2)                                               \
  {                                                                         \
    const int tn_ = (NEXT);                                                 \
    SW_SYNC(RAWPREV);                                                       \
    v2_gemm<DIAG>(a, cb, SW_NEXT(), gbias + (size_t)(NEXT2) * 32, raw_next, lane31x4, lane4hf, BA, BB, acc0, acc1); \
    ++pos;                                                                  \
    T = tn_;                                                                \
  }
  // ---- first Linear (3 tiles): h = ReLU(W1 x + b1), kept in the C/D register layout = B operand of the second Linear
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    if (m == 0) { V2_TILE(false, Bx0, Bx1, T + 1, 2); } else { V2_TILE(true, Bx0, Bx1, m < 2 ? T + 1 : first0e, m == 1 ? first0e : second0e); }
    if (m == 2) mfma_operand_guard();   // the first-Linear operands die here without a refill
    v2_set_hidden(h0, m, acc0);
    v2_set_hidden(h1, m, acc1);
    if constexpr (!(DIAG & 40)) bias_ready(cb);
  }

  if constexpr (DIAG == 4) st_t2 = stamp();
  const float* xc0 = xv0 - NS * 32 + j;     // xc[col * 32] is column col >= 32 of edge j (the scalar columns are NOT behind this pointer)
  const float* xc1 = xv1 - NS * 32 + j;
  const unsigned short* xs0 = xsb0 + j;      // bf_up(xs[i * 32]): scalar column i < 32 of edge j
  const unsigned short* xs1 = xsb1 + j;
  // ---- block 0e: one tile per mid index, 32 output scalars
  // The 0e tiles carry NO bias: sum_i m_i (w_i + b_i) = sum_i m_i w_i + sum_i b_i m_i, and the second sum is ONE small matrix product
  // per block -- A = the block's bias rows as a [32 outputs x 48 mids] bf16 tile (3 fragments behind the bias table of the stream),
  // B = the edge's mids -- accumulated straight into the output registers, which have the accumulator layout.  38 of the 57 tiles
  // then need neither the 16 bias registers' re-load nor its wait (a per-tile bias costs 11 % of the kernel whichever way it is
  // brought in: four broadcast dwordx4 loads or one dword + 16 ds_bpermute; timing-only diagnostics).
  f32x16 o0e0 = {}, o0e1 = {};
  bf16x8 ab0e[3];
  if (i_lo < i_hi) {
    const GFrag gb0e = (GFrag)reinterpret_cast<const bf16x8*>(reinterpret_cast<const float*>(reinterpret_cast<const bf16x8*>(G.wstream) + (size_t)(S.ntiles + 1) * V2_TILE_FRAGS) + (size_t)(S.ntiles + 1) * 32);
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) ab0e[s3] = gb0e[s3 * 64 + lane];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw_next));   // the bias dword the last first-Linear chain requested (asm load) is abandoned here
    if (vec_on && !(DIAG & 8)) {   // the vector blocks behind this one still take their bias as the C operand: request it now
      const GPtr<f32x4> gb4 = (GPtr<f32x4>)(gbias + (size_t)T_vec * 32);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const f32x4 b = gb4[hf + 2 * qq];
        cb[4 * qq + 0] = b.x; cb[4 * qq + 1] = b.y; cb[4 * qq + 2] = b.z; cb[4 * qq + 3] = b.w;
      }
      raw_next = gbias[(size_t)past0e_1 * 32 + (lane & 31)];
    }
  }
  // the mids are read from LDS BEFORE the MFMA chain (the fences inside v2_gemm keep the reads above it, their wait lands at the first
  // use below it): the LDS latency is covered by the MFMAs instead of being exposed after them.  Two loops, one per kind of mid (the
  // scalar features themselves, then the 1o . direction dot products), so that neither body branches on the mid index.
  // Software-pipelined (two accumulator sets A = acc0/acc1 and B = accB0/accB1; this block has ~70 registers of headroom below the
  // vector blocks' peak): the chain of tile i + 1 carries the epilogue of tile i between its MFMA pairs.
  f32x16 accB0, accB1;
  const int EPI_LO[V2_NFRAG + 1] = {0, 3, 6, 9, 12, 14, 16};
  auto next_of = [&](int i) { return i + 1 < i_hi ? 4 + i : past0e; };                              // stream index of the tile after 0e tile i
  // (macros, not lambdas taking the accumulators by reference: through reference parameters hipcc keeps all four accumulator tuples
  //  in scratch memory)
#define V2_EPI_ALL(Y0, Y1, M0, M1)                                                                                   \
  {                                                                                                                  \
    if constexpr (DIAG & 2) { o0e0[0] += Y0[0] + M0; o0e1[0] += Y1[0] + M1; } else {                                 \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) { o0e0[r] = fmaf(M0, Y0[r], o0e0[r]); o0e1[r] = fmaf(M1, Y1[r], o0e1[r]); } \
    }                                                                                                                \
  }
#define V2_CHAIN_PLAIN(I, X0, X1)                                                                                    \
  {                                                                                                                  \
    const int tn_ = next_of(I);                                                                                      \
    SW_SYNC(false);                                                                                                  \
    v2_gemm_p<DIAG, false>(a, cb, SW_NEXT(), gbias, raw_next, lane31x4, lane4hf, h0, h1,                       \
                    X0, X1, [](int) {});                                                                             \
    ++pos;                                                                                                           \
    T = tn_;                                                                                                         \
  }
#define V2_CHAIN_EPI(I, X0, X1, Y0, Y1, M0, M1)                                                                      \
  {                                                                                                                  \
    const int tn_ = next_of(I);                                                                                      \
    SW_SYNC(false);                                                                                                  \
    v2_gemm_p<DIAG, false>(a, cb, SW_NEXT(), gbias, raw_next, lane31x4, lane4hf, h0, h1,                       \
                    X0, X1, [&](int q) __attribute__((always_inline)) {                                              \
                      if constexpr (DIAG & 2) { if (q == 0) { o0e0[0] += Y0[0] + M0; o0e1[0] += Y1[0] + M1; } } else { \
                        _Pragma("unroll") for (int r = 0; r < 16; ++r)                                               \
                          if (r >= EPI_LO[q] && r < EPI_LO[q + 1]) { o0e0[r] = fmaf(M0, Y0[r], o0e0[r]); o0e1[r] = fmaf(M1, Y1[r], o0e1[r]); } \
                      }                                                                                              \
                    });                                                                                              \
    ++pos;                                                                                                           \
    T = tn_;                                                                                                         \
  }
  // tiles [IB, IE) whose mids come from MID(i, m0, m1) (LDS reads, issued one chain before they are used)
#define V2_RUN0E(IB, IE, MID)                                                                                        \
  if ((IB) < (IE)) {                                                                                                 \
    float ma0, ma1, mb0 = 0.f, mb1 = 0.f;                                                                            \
    MID((IB), ma0, ma1);                                                                                             \
    V2_CHAIN_PLAIN((IB), acc0, acc1);                                                                                \
    const int pairs_ = ((IE) - (IB) - 1) >> 1;   /* double steps: set B then set A again */                          \
    int i = (IB);                                                                                                    \
    _Pragma("unroll 1") for (int p_ = 0; p_ < pairs_; ++p_) {                                                        \
      MID(i + 1, mb0, mb1);                                                                                          \
      V2_CHAIN_EPI(i + 1, accB0, accB1, acc0, acc1, ma0, ma1);                                                       \
      MID(i + 2, ma0, ma1);                                                                                          \
      V2_CHAIN_EPI(i + 2, acc0, acc1, accB0, accB1, mb0, mb1);                                                       \
      i += 2;                                                                                                        \
    }                                                                                                                \
    if (i + 1 < (IE)) {   /* one tile left: set B, then drain it */                                                  \
      MID(i + 1, mb0, mb1);                                                                                          \
      V2_CHAIN_EPI(i + 1, accB0, accB1, acc0, acc1, ma0, ma1);                                                       \
      V2_EPI_ALL(accB0, accB1, mb0, mb1);                                                                            \
    } else {                                                                                                         \
      V2_EPI_ALL(acc0, acc1, ma0, ma1);                                                                              \
    }                                                                                                                \
  }
#define V2_MID_SCALAR(I, M0, M1) { M0 = bf_up(xs0[(I) * 32]); M1 = bf_up(xs1[(I) * 32]); }
#define V2_MID_DOT(I, M0, M1)                                                                                        \
  {                                                                                                                  \
    const float* p0_ = xc0 + (COL_1O + 3 * ((I) - NS)) * 32;                                                          \
    const float* p1_ = xc1 + (COL_1O + 3 * ((I) - NS)) * 32;                                                          \
    M0 = p0_[0] * v0[0] + p0_[32] * v0[1] + p0_[64] * v0[2];                                                         \
    M1 = p1_[0] * v1[0] + p1_[32] * v1[1] + p1_[64] * v1[2];                                                         \
  }
  const int i_mid = i_hi < NS ? i_hi : NS;
  V2_RUN0E(i_lo, i_mid, V2_MID_SCALAR);
  if constexpr (IN >= 1) {
    const int i_b = i_lo > NS ? i_lo : NS;
    V2_RUN0E(i_b, i_hi, V2_MID_DOT);
  }
  if (i_lo < i_hi) {
    // sum_i b_i m_i for the mids [i_lo, i_hi) of this wave (a virtual slice of a group runs only part of the 0e tiles)
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
      bf16x8 bm0, bm1;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int i = 16 * s3 + 8 * hf + jj;       // mid index of element jj of this lane half's k-slice (runtime through hf only)
        float m0 = 0.f, m1 = 0.f;
        if (s3 < 2) { m0 = bf_up(xs0[i * 32]); m1 = bf_up(xs1[i * 32]); }
        else if constexpr (IN >= 1) {
          if (jj < S.n1o) {                         // mids 32 .. 37 live in the lower lane half's slice (i = 32 + jj); the upper half's are padding
            const float* p0 = xc0 + (COL_1O + 3 * jj) * 32;
            const float* p1 = xc1 + (COL_1O + 3 * jj) * 32;
            m0 = p0[0] * v0[0] + p0[32] * v0[1] + p0[64] * v0[2];
            m1 = p1[0] * v1[0] + p1[32] * v1[1] + p1[64] * v1[2];
          }
          if (hf) { m0 = 0.f; m1 = 0.f; }
        }
        const bool in = i >= i_lo && i < i_hi;
        bm0[jj] = (__bf16)(in ? m0 : 0.f);
        bm1[jj] = (__bf16)(in ? m1 : 0.f);
      }
      o0e0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab0e[s3], bm0, o0e0, 0, 0, 0);
      o0e1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab0e[s3], bm1, o0e1, 0, 0, 0);
    }
  }
#undef V2_RUN0E
#undef V2_CHAIN_EPI
#undef V2_CHAIN_PLAIN
#undef V2_EPI_ALL
#undef V2_MID_SCALAR
#undef V2_MID_DOT

  if constexpr (DIAG == 4) st_0e = stamp();
  // ---- vector / pseudoscalar blocks: tile = 5 mid indices x 6 outputs; lane half hf owns outputs 3hf..3hf+2
  float k1o0[9], k1e0[9], k0o0[3], k1o1[9], k1e1[9], k0o1[3];
#pragma unroll
  for (int r = 0; r < 9; ++r) { k1o0[r] = 0.f; k1e0[r] = 0.f; k1o1[r] = 0.f; k1e1[r] = 0.f; }
  k0o0[0] = k0o0[1] = k0o0[2] = 0.f;
  k0o1[0] = k0o1[1] = k0o1[2] = 0.f;

  // The tile loops of the vector blocks are FULLY unrolled: the mid index is then a compile-time constant, the kind of every mid
  // (scalar x direction, copy, cross product, padding) is resolved by the compiler, and the LDS reads of a tile's ten mids are issued
  // together before the chain.  Rolled, every mid was a chain of scalar branches around LDS reads that were each waited for in turn:
  // 44 s_waitcnt and ~3.6 k cycles per vector tile against ~0.7 k for a scalar tile (in-kernel stamps, round 2).
  // Mids of the form (scalar feature) x (edge direction) -- the first NS mids of block 1o, the last n0o of block 1e -- are not
  // multiplied out: sum_i (x_i v_c) w_io = v_c sum_i x_i w_io, so they cost one FMA per output instead of three and the direction is
  // applied once per block (`is_scalar(i)` / `scalar_of(x, i)` describe them; padded slots i >= fan are skipped altogether).
  auto vec_block = [&](auto mid_fn, auto is_scalar, auto scalar_of, auto ntile_c, auto fan_c, float (&keep0)[9], float (&keep1)[9])
                       __attribute__((always_inline)) {
    constexpr int ntile = decltype(ntile_c)::value, fan = decltype(fan_c)::value;
    float s0[3] = {0.f, 0.f, 0.f}, s1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < ntile; ++t) {
      // vector-valued mids are evaluated before the chain (LDS latency and the cross products under the MFMAs); the scalar ones are
      // read behind it -- ten more live registers across the chain would spill
      float ma[VEC_TILE_I][3], mb[VEC_TILE_I][3];
      if constexpr (!(DIAG & 2)) {
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          const int i = VEC_TILE_I * t + q;
          if (i >= fan || is_scalar(i)) continue;
          mid_fn(xc0, i, v0, ma[q]); mid_fn(xc1, i, v1, mb[q]);
        }
      }
      if (t == 0) { V2_TILE(false, h0, h1, T + 1, T + 2 < S.ntiles ? T + 2 : S.ntiles); } else { V2_TILE(true, h0, h1, T + 1, T + 2 < S.ntiles ? T + 2 : S.ntiles); }
      if constexpr (DIAG & 2) { keep0[0] += acc0[0]; keep1[0] += acc1[0]; if constexpr (!(DIAG & 40)) bias_ready(cb); continue; }
#pragma unroll
      for (int q = 0; q < VEC_TILE_I; ++q) {
        const int i = VEC_TILE_I * t + q;
        if (i >= fan) continue;
        float xa = 0.f, xb = 0.f;
        if (is_scalar(i)) { xa = scalar_of(xc0, xs0, i); xb = scalar_of(xc1, xs1, i); }
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const float wa = acc0[3 * q + o], wb = acc1[3 * q + o];
          if (is_scalar(i)) {
            s0[o] = fmaf(xa, wa, s0[o]);
            s1[o] = fmaf(xb, wb, s1[o]);
          } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              keep0[3 * o + c] = fmaf(ma[q][c], wa, keep0[3 * o + c]);
              keep1[3 * o + c] = fmaf(mb[q][c], wb, keep1[3 * o + c]);
            }
          }
        }
      }
      // the tile's FMAs have no side effect, so nothing ties them to this place in the fully unrolled code: without the pins hipcc
      // parks the accumulators of the scalar tiles in scratch and does their FMAs several tiles later (1 KB of spills per lane)
#pragma unroll
      for (int o = 0; o < 3; ++o) { pin(s0[o]); pin(s1[o]); }
      if constexpr (!(DIAG & 40)) bias_ready(cb);
    }
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        keep0[3 * o + c] = fmaf(v0[c], s0[o], keep0[3 * o + c]);
        keep1[3 * o + c] = fmaf(v1[c], s1[o], keep1[3 * o + c]);
      }
  };
  if (vec_on) {
    vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1o<IN>(x, i, vv, m); },
              [](int i) { return i < NS; }, [](const float*, const unsigned short* xs, int i) { return bf_up(xs[i * 32]); },
              std::integral_constant<int, S.t1o>{}, std::integral_constant<int, S.fan1o>{}, k1o0, k1o1);
    if constexpr (OUT >= 2)
      vec_block([](const float* x, int i, const float (&vv)[3], float (&m)[3]) __attribute__((always_inline)) { mid1e<IN>(x, i, vv, m); },
                [](int i) { return i >= S.n1o + S.n1e; }, [](const float* x, const unsigned short*, int i) { return x[(COL_0O + (i - S.n1o - S.n1e)) * 32]; },
                std::integral_constant<int, S.t1e>{}, std::integral_constant<int, S.fan1e>{}, k1e0, k1e1);
    if constexpr (OUT >= 3) {
#pragma unroll
      for (int t = 0; t < S.t0o; ++t) {
        float ma[VEC_TILE_I], mb[VEC_TILE_I];
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          if (VEC_TILE_I * t + q >= S.fan0o) continue;
          ma[q] = mid0o<IN>(xc0, VEC_TILE_I * t + q, v0); mb[q] = mid0o<IN>(xc1, VEC_TILE_I * t + q, v1);
        }
        if (t == 0) { V2_TILE(false, h0, h1, T + 1, T + 2 < S.ntiles ? T + 2 : S.ntiles); } else { V2_TILE(true, h0, h1, T + 1, T + 2 < S.ntiles ? T + 2 : S.ntiles); }
        if constexpr (DIAG & 2) { k0o0[0] += acc0[0]; k0o1[0] += acc1[0]; if constexpr (!(DIAG & 40)) bias_ready(cb); continue; }
#pragma unroll
        for (int q = 0; q < VEC_TILE_I; ++q) {
          if (VEC_TILE_I * t + q >= S.fan0o) continue;
#pragma unroll
          for (int o = 0; o < 3; ++o) { k0o0[o] = fmaf(ma[q], acc0[3 * q + o], k0o0[o]); k0o1[o] = fmaf(mb[q], acc1[3 * q + o], k0o1[o]); }
        }
        if constexpr (!(DIAG & 40)) bias_ready(cb);
      }
    }
  }
#undef V2_TILE

  // ---- messages -> LDS and run-length sums per aggregating node, ONE sub-tile at a time in the wave's 15 KB (the message tile of a
  //      sub-tile, [74][33] fp32, overlays both gather images: every mid has been read by now)
  if constexpr (DIAG == 4) st_t3 = stamp();
  // drains the DMAs of the tiles behind the end of the sequence (nobody reads them; they must not land in the LDS of the NEXT workgroup
  // on this CU) and the last bias dword (its register is re-used from here on)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(raw_next)::"memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll 1
  for (int sub = 0; sub < 2; ++sub) {
    const f32x16& o0e = sub ? o0e1 : o0e0;
    const float* k1o = sub ? k1o1 : k1o0;
    const float* k1e = sub ? k1e1 : k1e0;
    const float* k0o = sub ? k0o1 : k0o0;
#pragma unroll
    for (int r = 0; r < 16; ++r) msg[((r & 3) + 8 * (r >> 2) + 4 * hf) * OUT_STRIDE + j] = o0e[r];
#pragma unroll
    for (int o = 0; o < 3; ++o)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        msg[(COL_1O + 3 * (3 * hf + o) + c) * OUT_STRIDE + j] = k1o[3 * o + c];
        if constexpr (OUT >= 2) msg[(COL_1E + 3 * (3 * hf + o) + c) * OUT_STRIDE + j] = k1e[3 * o + c];
      }
    if constexpr (OUT >= 3) {
#pragma unroll
      for (int o = 0; o < 3; ++o) msg[(COL_0O + 3 * hf + o) * OUT_STRIDE + j] = k0o[o];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's message stores are in LDS (the tile is wave-private)
    reduce_runs<NODE_STRIDE, OUT_STRIDE>(msg, srcl + 32 * sub, lane, S.out_dim, G.first_sum + (size_t)(tile_local + sub) * NODE_STRIDE,
                G.last_sum + (size_t)(tile_local + sub) * NODE_STRIDE, G.run_acc);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // ... and read back before the next sub-tile overwrites them
  }
  if constexpr (DIAG == 4) {   // same record layout as tp_conv_kernel's CBD_CONV_VARIANT=8 stamps (tools/conv_clock.py)
    if (lane == 0 && args.stamps && blockIdx.x * SW_WAVES + wave < 8192) {
      unsigned long long* o = args.stamps + (size_t)(blockIdx.x * SW_WAVES + wave) * 8;
      o[0] = st_t0; o[1] = st_r0; o[2] = stamp(); o[3] = __builtin_amdgcn_s_memrealtime();
      o[4] = st_t1; o[5] = st_t2; o[6] = st_t3; o[7] = st_0e;   // slot 7: end of the 0e block (fp32 kernel: end of the first first-Linear tile)
    }
  }
}

template <int IN, int OUT>
static hipError_t launch_one64s(const ConvArgs& a, int grid, hipStream_t s) {
  static const int diag = getenv("CBD_BF16_DIAG") ? atoi(getenv("CBD_BF16_DIAG")) : 0;
  static bool attr_done = false;
  if (!attr_done) {   // > 64 KB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64s_kernel<IN, OUT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS_BYTES);
    if (IN == 3) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_conv64s_kernel<IN, OUT, (IN == 3 ? 4 : 0)>), hipFuncAttributeMaxDynamicSharedMemorySize, SW_LDS_BYTES);
    attr_done = true;
  }
  if (IN == 3 && diag == 4) hipLaunchKernelGGL((tp_conv64s_kernel<IN, OUT, (IN == 3 ? 4 : 0)>), dim3(grid), dim3(64 * SW_WAVES), SW_LDS_BYTES, s, a);
  else hipLaunchKernelGGL((tp_conv64s_kernel<IN, OUT>), dim3(grid), dim3(64 * SW_WAVES), SW_LDS_BYTES, s, a);
  return hipGetLastError();
}

}  // namespace shared_w

// grid64: number of 64-edge waves the capacities of the groups need (sum over groups of ceil(cap / 64), what the second-generation
// launcher takes): the workgroups of 512 edges are at most grid64 / 4 + one per group
hipError_t launch_tp_conv_bf16s(int in_level, int out_level, const ConvArgs& a, int grid64, hipStream_t s) {
  if (grid64 <= 0) return hipSuccess;
  const int grid = (grid64 + shared_w::SW_WAVES - 1) / shared_w::SW_WAVES + a.n_groups;
  if (in_level == 0 && out_level == 1) return shared_w::launch_one64s<0, 1>(a, grid, s);
  if (in_level == 1 && out_level == 2) return shared_w::launch_one64s<1, 2>(a, grid, s);
  if (in_level == 2 && out_level == 3) return shared_w::launch_one64s<2, 3>(a, grid, s);
  if (in_level == 3 && out_level == 3) return shared_w::launch_one64s<3, 3>(a, grid, s);
  return hipErrorInvalidValue;
}

}  // namespace cbd
