// Shared device/host definitions of the MI355X docking engine (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cbd {

// Weight streams are read through address-space-1 pointers (`GPtr`, reduce_runs.h) with a WAVE-UNIFORM base: global_load with an SGPR
// base pair advanced on the scalar unit + one constant per-lane VGPR offset + an immediate, i.e. exact vmcnt(N) waits and no vector
// address arithmetic inside or between the MFMA chains (tp_conv, tp_conv_bf16, fctp_conv, tp_train).  History: a first attempt that
// only cast the per-lane FLAT pointers to address space 1 changed nothing in throughput and exposed the in-flight-MFMA-operand
// hazard of the in-place fragment refill in the bf16 kernel (tp_conv_dev.h); with the bases pinned in SGPRs BEFORE a chain the chain
// holds only MFMAs and loads and the hazard is out (repeatability soak: tools/bf16_repeat.py, all three operand modes).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- data layout in HBM -------------------------------------------------------------------------------------
// Node features: one row of NODE_STRIDE floats per node (74 used: 32x0e | 6x1o | 6x1e | 6x0o, rest zero).
// Joint node index = ligand nodes of all samples first ([b*Nl + a]), then receptor nodes ([nL + b*Nr + r]),
// exactly the concatenation order of the reference (models/score_model.py:354).
constexpr int NODE_STRIDE = 80;
constexpr int NS = 32;
constexpr int NV = 6;
constexpr int COL_1O = 32, COL_1E = 50, COL_0O = 68, NODE_DIM = 74;

// ---- tensor-product convolution tiling ----------------------------------------------------------------------
// One wave (= one workgroup) owns 32 consecutive edges of ONE edge group (the N dimension of v_mfma_f32_32x32x2_f32).
constexpr int WAVE_EDGES = 32;
constexpr int CONV_WG_EDGES = WAVE_EDGES;      // one wave per workgroup
constexpr int KDIM = 96;                       // radial-MLP width (3*ns)
constexpr int KSTEPS = KDIM / 2;               // 48 MFMA k-steps of 2
constexpr int TILE_W_FLOATS = KSTEPS * 64;     // 3072 weight floats per 32-row tile
// Weight stream of one FCBlock: (ntiles + 1) tiles of TILE_W_FLOATS (the last one zero: prefetch target of the last
// real tile), followed by ntiles x 32 bias floats.
__host__ __device__ constexpr size_t conv_stream_floats(int ntiles) { return (size_t)(ntiles + 1) * TILE_W_FLOATS + (size_t)ntiles * 32; }
// Row layout of a vector / pseudoscalar block tile (m_out = 6): accumulator register reg < 15 of lane half hf holds
// (mid index i = 5*tile + reg/3, output o = 3*hf + reg%3); 30 of the 32 rows carry weights (reg 15 is zero).
constexpr int VEC_TILE_I = 5;

// Number of 32-row weight tiles of a layer with input level IN (0..3) and output level OUT (1..3):
// 3 tiles of the first Linear, then the second Linear regrouped per output irrep block.
struct ConvShape {
  int n1o, n1e, n0o;          // input multiplicities
  int fan0e, fan1o, fan1e, fan0o;
  int t0e, t1o, t1e, t0o;     // tiles per block
  int vmerged;                // 1: block 1e's partly filled last tile is merged into block 0o's (conv_shape(.., merged = true))
  int ntiles;
  int weight_numel;           // reference weight_numel (1216/1480/1588/1660)
  int in_dim, out_dim;
};

// `merged` (the inference kernels of tp_conv.hip): the vector blocks' last tiles are partly filled (fan % 5 of 5 mid slots).  Where the
// remainders of blocks 1e and 0o fit one tile, block 1e's tail mids ride in the free slots behind block 0o's own in 0o's last tile and
// block 1e runs one tile less: 57 -> 56 tiles for the 74 -> 74 layers.  The training kernels (tp_train.hip) and the bf16 kernel keep the
// plain layout.
__host__ __device__ constexpr ConvShape conv_shape(int IN, int OUT, bool merged = false) {
  ConvShape s{};
  s.n1o = IN >= 1 ? NV : 0;
  s.n1e = IN >= 2 ? NV : 0;
  s.n0o = IN >= 3 ? NV : 0;
  s.fan0e = NS + s.n1o;
  s.fan1o = NS + s.n1o + s.n1e;
  s.fan1e = OUT >= 2 ? s.n1o + s.n1e + s.n0o : 0;
  s.fan0o = OUT >= 3 ? s.n1e + s.n0o : 0;
  s.t0e = s.fan0e;
  s.t1o = (s.fan1o + VEC_TILE_I - 1) / VEC_TILE_I;
  s.t1e = (s.fan1e + VEC_TILE_I - 1) / VEC_TILE_I;
  s.t0o = (s.fan0o + VEC_TILE_I - 1) / VEC_TILE_I;
  s.vmerged = merged && OUT >= 3 && s.fan1e % VEC_TILE_I > 0 && s.fan0o % VEC_TILE_I > 0 &&
              s.fan1e % VEC_TILE_I + s.fan0o % VEC_TILE_I <= VEC_TILE_I ? 1 : 0;
  s.ntiles = 3 + s.t0e + s.t1o + s.t1e + s.t0o - s.vmerged;
  s.weight_numel = s.fan0e * NS + s.fan1o * NV + s.fan1e * NV + s.fan0o * NV;
  s.in_dim = NS + 3 * s.n1o + 3 * s.n1e + s.n0o;
  s.out_dim = NS + 3 * NV + (OUT >= 2 ? 3 * NV : 0) + (OUT >= 3 ? NV : 0);
  return s;
}

struct ConvGroup {
  const int* src;        // [cap] aggregating node (reference edge_index[0]), joint index, sorted ascending
  const int* dst;        // [cap] node whose features are read (edge_index[1])
  const int* attr_idx;   // [cap] row of `attr`
  const float* vec;      // [cap][4] unit edge vector (xyz, 0): sh = [1, sqrt3 * v]
  const float* attr;     // [.][32] embedded edge attributes
  const float* wstream;  // conv_stream_floats(ntiles) re-packed weights of this group's FCBlock
  const int* count;      // device scalar: number of edges
  // deterministic segmented reduction (no atomics): per 32-edge tile the sum of its first run, of its last run, and
  // direct stores for runs that lie strictly inside the tile (the aggregating node then has no other tile in this group)
  float* first_sum;      // [tiles][NODE_STRIDE]
  float* last_sum;       // [tiles][NODE_STRIDE]
  float* run_acc;        // [N][NODE_STRIDE], row = aggregating node
  const float* node_in;  // [N][NODE_STRIDE] node features this group's src / dst indices refer to
  // First Linear of the radial MLP, node part: W1 [edge_attr | x_src[:32] | x_dst[:32]] = W1a edge_attr + (W1s x_src) + (W1d x_dst).
  // The two node terms are the same for every edge of a node, so they are projected ONCE PER NODE (node_proj_kernel, 2 x 96 x 32
  // MACs per node instead of per edge) and enter the edge kernel through the accumulator; only the K = 32 edge-attribute part of
  // the first Linear runs on the matrix cores per edge (2 of the 57 tiles' worth of MFMA work less: +4 % measured as a bound).
  const float* psrc;     // [N][KDIM] W1s x[:, :32] for the rows this group uses as aggregating node
  const float* pdst;     // [N][KDIM] W1d x[:, :32] for the rows it reads
  // weight-tile slice executed for this group (run_conv fills the full range): 0e tiles [i0e_lo, i0e_hi) and, if vec_on, the
  // 1o/1e/0o blocks.  Virtual slices of one edge group write their own piece buffers; the finalize kernel adds them.
  int i0e_lo, i0e_hi, vec_on;
  // relative cost of one 32-edge unit of this group in 1/64 (0 = 64): what the persistent bf16 kernel weighs a group's units with when it
  // cuts a launch into equal pieces of WORK per workgroup (tp_conv_bf16s.hip; measured per role).  Other kernels ignore it.
  int cost_w;
};

// Host-side view of a group: plus the node-row ranges its src / dst indices fall in (what node_proj_kernel has to cover).  Kept out
// of ConvGroup so that 32 of them fit the 4 KB kernel-argument limit.
struct ConvGroupH : ConvGroup {
  int src_lo = 0, src_n = 0, dst_lo = 0, dst_n = 0;
};

// Up to 32 groups per launch: the 4 edge groups of one batch, or of up to EIGHT batches (engines working on different
// complexes co-scheduled by cbd_sample_multi so that one launch carries several times the waves -- the per-launch drain is
// amortised).
constexpr int CONV_MAX_GROUPS = 32;
// node_proj_kernel jobs of one layer: P[row][0..95] = sum_k WT[k][.] * node[row][k] for rows [lo, lo + n)
constexpr int PROJ_MAX_JOBS = 64;   // 4 FCBlocks x {aggregating, read} role x up to 8 co-scheduled batches
struct ProjJob { const float* WT; float* P; const float* node_in; int lo, n; };
struct ProjArgs { ProjJob job[PROJ_MAX_JOBS]; int n_jobs; };
static_assert(sizeof(ProjArgs) <= 4096, "ProjArgs is passed by value");

struct ConvArgs {
  ConvGroup g[CONV_MAX_GROUPS];
  int n_groups;
  unsigned long long* stamps;   // diagnostic build only (CBD_CONV_VARIANT=8): [grid][4] s_memtime/s_memrealtime at start/end
};
static_assert(sizeof(ConvArgs) <= 4096, "ConvArgs is passed by value: HIP kernel arguments are limited to 4 KB");

// One edge group as seen by the finalize kernel: CSR range of every node + the partial sums written by tp_conv.
struct FinGroup {
  const int* start;        // [nodes of this type] first edge of the node in the group's edge list
  const int* cnt;          // [nodes of this type] number of edges
  const int* total;        // device scalar: number of edges in the group
  const float* first_sum;
  const float* last_sum;
  const float* run_acc;
  int node_mod;            // > 0: the group is shared by all samples, node k = i % node_mod (layer-0 receptor edges)
  int deg_weight;          // 1: its edges count towards the node's in-degree; 0: a further slice of an already counted group
  int col_hi;              // > 0: this slice's pieces hold only the columns [0, col_hi) (a 0e-only slice of the bf16 role split)
};
struct FinArgs {
  FinGroup g[4];
  int n_groups;
};

}  // namespace cbd
