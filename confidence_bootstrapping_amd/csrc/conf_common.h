// Shared definitions of the all-atom CONFIDENCE model engine (gfx950 only).
// Architecture = workdir/pretrained_confidence/model_parameters.yml of the reference: ns = 24, nv = 6, sh_lmax = 2,
// node features 24x0e | 6x1o | 6x1e | 24x0o (84 floats), radial MLP 72 -> 72 -> W, 9 edge groups.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

namespace cbd {

// ---- node layout ---------------------------------------------------------------------------------------------
// Joint node index = ligand atoms of all poses [b*Nl + a], then residues [nL + b*Nr + r], then receptor atoms
// [nL + B*Nr + b*Na + k] (the concatenation order of reference models/all_atom_score_model.py:398).  Every pose keeps
// a slot for every residue / atom of the un-cropped complex; nodes dropped by crop_beyond simply have no edges.
constexpr int CNS = 24;
constexpr int CNV = 6;
constexpr int CN_STRIDE = 84;
constexpr int CC_1O = 24, CC_1E = 42, CC_0O = 60;

// ---- fused convolution tiling (same scheme as tp_conv.hip, K = 72) ---------------------------------------------
constexpr int CKDIM = 72;                       // radial-MLP width (3 * ns)
constexpr int CKSTEPS = CKDIM / 2;              // 36 MFMA k-steps of v_mfma_f32_32x32x2_f32
constexpr int CTILE_W_FLOATS = CKSTEPS * 64;    // 2304 weight floats per 32-row tile
constexpr int C_SC_TILE_I = 4;                  // scalar-block tile: 4 mid indices x 8 outputs (3 tiles cover 24 outputs)
constexpr int C_VEC_TILE_I = 5;                 // vector-block tile: 5 mid indices x 6 outputs
constexpr int CONF_MAX_GROUPS = 9;
__host__ __device__ constexpr size_t fctp_stream_floats(int ntiles) { return (size_t)(ntiles + 1) * CTILE_W_FLOATS + (size_t)ntiles * 32; }

// Layer shape for input level IN (0: 0e, 1: +1o, 2: +1e, 3: +0o) and output level OUT (1..3).  The e3nn
// FullyConnectedTensorProduct(in, 1x0e+1x1o+1x2e, out) paths regrouped per OUTPUT irrep ("block"); the mid index space
// of each block concatenates its paths in the order listed:
//   0e : [0e x Y0 (24)] [1o . Y1 (n1o)]
//   1o : [0e x Y1 (24)] [1o x Y0 (n1o)] [1o x Y2 (n1o)] [1e x Y1 (n1e)]
//   1e : [1o x Y1 (n1o)] [1e x Y0 (n1e)] [1e x Y2 (n1e)] [0o x Y1 (n0o)]
//   0o : [1e . Y1 (n1e)] [0o x Y0 (n0o)]
struct FctpShape {
  int n1o, n1e, n0o;
  int fan0e, fan1o, fan1e, fan0o;
  int g0e, t1o, t1e, g0o;     // scalar blocks: groups of 4 mids (3 tiles each; 2 for a tail of <= 2 mids); vector blocks: tiles of 5 mids
  int merged;                 // 1: the half-empty second tail tile of block 0e rides in the free rows of block 0o's (see fctp_shape)
  int vmerged;                // 1: the mids of block 1o's partly filled last tile ride in the free slots of block 1e's last tile
  int ntiles;
  int weight_numel;           // 720 / 972 / 1224 / 1944
  int in_dim, out_dim;
};

// Tiles of a scalar block with `fan` mids: three per full group of 4 mids; a TAIL group of one or two mids takes two denser tiles instead
// of three half-empty ones -- tile A = slots (mid0, outputs 0..7) (mid1, 0..7) (mid0, 8..15) (mid1, 8..15), tile B = (mid0, 16..23)
// (mid1, 16..23) + 16 zero rows.  (Every tail of the shipped architecture has two mids: fan = 30.)
__host__ __device__ constexpr bool sc_tail_dense(int fan, int g) { return fan - C_SC_TILE_I * g >= 1 && fan - C_SC_TILE_I * g <= 2; }
__host__ __device__ constexpr int sc_block_tiles(int fan) {
  const int groups = (fan + C_SC_TILE_I - 1) / C_SC_TILE_I;
  return groups == 0 ? 0 : 3 * (groups - 1) + (sc_tail_dense(fan, groups - 1) ? 2 : 3);
}

__host__ __device__ constexpr FctpShape fctp_shape(int IN, int OUT) {
  FctpShape s{};
  s.n1o = IN >= 1 ? CNV : 0;
  s.n1e = IN >= 2 ? CNV : 0;
  s.n0o = IN >= 3 ? CNS : 0;
  s.fan0e = CNS + s.n1o;
  s.fan1o = CNS + 2 * s.n1o + s.n1e;
  s.fan1e = OUT >= 2 ? s.n1o + 2 * s.n1e + s.n0o : 0;
  s.fan0o = OUT >= 3 ? s.n1e + s.n0o : 0;
  s.g0e = (s.fan0e + C_SC_TILE_I - 1) / C_SC_TILE_I;
  s.t1o = (s.fan1o + C_VEC_TILE_I - 1) / C_VEC_TILE_I;
  s.t1e = (s.fan1e + C_VEC_TILE_I - 1) / C_VEC_TILE_I;
  s.g0o = (s.fan0o + C_SC_TILE_I - 1) / C_SC_TILE_I;
  // Both scalar blocks of the 2 -> 3 and 3 -> 3 layers end in a dense tail whose tile B has 16 live rows (octet 2 of two mids): block 0e's
  // are packed into slots 2, 3 of block 0o's tile B instead of a tile of their own (-1 tile of 67 / 48).
  s.merged = OUT >= 3 && s.g0e > 0 && s.g0o > 0 && sc_tail_dense(s.fan0e, s.g0e - 1) && sc_tail_dense(s.fan0o, s.g0o - 1) ? 1 : 0;
  // The vector blocks' last tiles are partly filled too (fan % 5 of 5 mid slots): when both remainders fit one tile, block 1o's tail
  // mids take the slots behind block 1e's (block 1o then runs t1o - 1 tiles of its own).
  s.vmerged = OUT >= 2 && s.fan1o % C_VEC_TILE_I > 0 && s.fan1e % C_VEC_TILE_I > 0 &&
              s.fan1o % C_VEC_TILE_I + s.fan1e % C_VEC_TILE_I <= C_VEC_TILE_I ? 1 : 0;
  s.ntiles = 3 + sc_block_tiles(s.fan0e) + s.t1o + s.t1e + sc_block_tiles(s.fan0o) - s.merged - s.vmerged;
  s.weight_numel = s.fan0e * CNS + s.fan1o * CNV + s.fan1e * CNV + s.fan0o * CNS;
  s.in_dim = CNS + 3 * s.n1o + 3 * s.n1e + s.n0o;
  s.out_dim = CNS + 3 * CNV + (OUT >= 2 ? 3 * CNV : 0) + (OUT >= 3 ? CNS : 0);
  return s;
}

struct CGroup {
  const int* src;        // [cap] aggregating node (edge_index[0]), joint index, ascending
  const int* dst;        // [cap] node whose features are read (edge_index[1])
  const int* attr_idx;   // [cap] row of `attr` AND of `vec` (flipped groups reuse the forward edge's attributes and
                         //       spherical harmonics, all_atom_score_model.py:409-415)
  const float* vec;      // [.][4] unit edge vector
  const float* attr;     // [.][24] embedded edge attributes
  const float* wstream;  // fctp_stream_floats(ntiles) re-packed FCBlock of this group
  const int* count;      // device scalar
  float* first_sum;      // [tiles][CN_STRIDE]
  float* last_sum;       // [tiles][CN_STRIDE]
  float* run_acc;        // row = aggregating node (joint index; pointer pre-offset by the node type's base)
  const float* node_in;  // [N][CN_STRIDE] node features of the pose batch this group belongs to
};

// One launch carries the groups of up to four pose batches (cbd_conf_score_multi: the final poses of the complexes of a co-scheduled
// group are scored together, so that a launch covers ~60 rounds of resident waves instead of ~15 and the last, partly filled round
// costs 1 % instead of 6 %).
constexpr int CONF_MAX_BATCHES = 4;
constexpr int CONF_LAUNCH_GROUPS = CONF_MAX_GROUPS * CONF_MAX_BATCHES;
struct CArgs {
  CGroup g[CONF_LAUNCH_GROUPS];
  int n_groups;
};
static_assert(sizeof(CArgs) <= 4096, "CArgs is passed by value: HIP kernel arguments are limited to 4 KB");

struct CFinGroup {
  const int* start;      // [nodes of the type]
  const int* cnt;
  const float* first_sum;
  const float* last_sum;
  const float* run_acc;  // pre-offset like CGroup::run_acc
};
struct CFinArgs {
  CFinGroup g[3];
  int n_groups;
};

hipError_t launch_fctp_conv(int in_level, int out_level, const CArgs& a, int grid, hipStream_t s);
hipError_t launch_fctp_finalize(const CFinArgs& fa, const float* node_in, float* node_out, const float* bn_scale,
                                const float* bn_mean, const float* bn_bias, int n_nodes, int in_dim, int out_dim,
                                int node_off, hipStream_t s);

}  // namespace cbd
