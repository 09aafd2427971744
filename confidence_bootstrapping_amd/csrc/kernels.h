// Launcher declarations of the non-GEMM kernels (graph construction, edge featurisation, heads, pose update).
#pragma once
#include "common.h"

namespace cbd {

// A 2-layer edge MLP whose first layer is split into [per-step constant part | gaussian part | optional 4 bond flags]:
//   hid = relu(part + WgT^T gauss(d) + WbT^T bond4) ; out = b1 + W1T^T hid          (all 32 wide)
struct EdgeMlp {
  const float* part;    // [32] per-step constant incl. first bias (or just the bias)
  const float* WgT;     // [32 k][32 o]
  const float* WbT;     // [4][32 o] or nullptr
  const float* W1T;     // [32 k][32 o]
  const float* b1;      // [32]
  const float* offset;  // [32] gaussian centres
  float coeff;
};

struct GraphStatic {     // per complex, device pointers
  int Nl, Nr, R, nbd, Err;
  int rec_off;                   // joint index of receptor node (b=0, r=0): max_batch * Nl (independent of B)
  const float* rec_pos;          // [Nr][3]
  const int* bond_row;           // [Nl+1] CSR over bond directions sorted by src atom
  const int* bond_dst;           // [nbd] (sorted by src)
  const float* bond_attr;        // [nbd][4] (sorted by src)
  const int* rr_deg0;            // [Nr] in-degree (as aggregating node) in the receptor kNN graph
  const int* rot_u; const int* rot_v;   // [R]
  const uint8_t* mask_rotate;    // [R][Nl]
};

struct GraphDyn {        // per forward pass, device pointers (capacity sized)
  float* pos;                    // [B][Nl][3]
  int* cnt_ll; int* cnt_lr; int* cnt_rl;        // per node counts   [B*Nl], [B*Nl], [B*Nr]
  int* start_ll; int* start_lr; int* start_rl;  // exclusive scans
  int* counts;                   // [8]: ll, lr, rr, rl, tor, ...
  int *ll_src, *ll_dst, *ll_aidx; float* ll_vec; float* ll_dist; float* ll_bond4;
  int *lr_src, *lr_dst, *lr_aidx; float* lr_vec; float* lr_dist;
  int *rl_src, *rl_dst, *rl_aidx; float* rl_vec;
  int* pair_eid;                 // [B*Nl*Nr]
};

// ---------------------------------------------------------------------------------------------------------------------------
// One launch, several complexes.  Every kernel of the step loop takes a table of up to MAX_COSCHED pose batches (B poses of ONE
// complex each, different Nl / Nr / R per batch) and a prefix sum of the workgroups each batch contributes; a workgroup finds its
// batch with <= 8 scalar comparisons and then runs exactly the single-complex body on that batch's descriptor.  The descriptors
// live in device memory (one per engine, written by set_desc_kernel at the start of a call) so the kernel-argument block stays
// small and a captured hipGraph can be replayed after the descriptors were refreshed.
constexpr int MAX_COSCHED = 8;

struct StepVectors {     // per-step device vectors, 32 floats each (functions of the diffusion time and the weights only: one set
  float *rec_sigma_emb, *ll_part, *lr_part, *center_part, *lig_node_c, *tr_part, *rot_part;   // serves all co-scheduled batches)
};

struct PoseBatch {
  GraphStatic gs;
  GraphDyn gd;                   // gd.pos: the poses this call works on
  int B;                         // poses in this call (<= max_batch)
  int cap_ll, cap_x;             // edge capacities of this call: B * cap_ll_per_sample, B * Nl * Nr
  float* X[2];                   // node-feature ping-pong buffers [max_batch*(Nl+Nr)][NODE_STRIDE]
  const float* lig_static32;     // [Nl][32]
  const float* rec_static;       // [Nr][NODE_STRIDE] time-independent receptor embedding
  const float* rr_attr0;         // [Err][32] embedded receptor edge attributes without the sigma part
  float* rr_attr_t;              // [Err][32] + rec_sigma_emb of this step
  float *ll_attr, *lr_attr;      // embedded edge attributes of the ligand / cross graph
  float *tr_out, *rot_out, *tor_out;   // scores of the current step [B,3], [B,3], [B*R]
  float *center_msg, *dbg_global, *dbg_torfeat;
  int *tor_nb, *tor_nb_cnt;
  const float *z_tr, *z_rot, *z_tor;   // pre-drawn noise of the whole call [S][B][3], [S][B][3], [S][B*R] (or null)
  unsigned long long* stats;     // [4] work counters
  FinArgs fin_emb, fin_lig, fin_rec, fin_rec_shared;   // segmented-sum descriptions of the layer kinds (static per complex)
#ifdef CBD_EXPERIMENTS
  FinArgs fin_lig_r, fin_rec_r, fin_rec_shared_r;      // the same with the second 0e slice of the cross / receptor groups (bf16 role split, diagnostic library only)
#endif
};

struct Multi {
  int n;
  const PoseBatch* d[MAX_COSCHED];
  int off[MAX_COSCHED + 1];      // off[k] = first workgroup of batch k in THIS launch, off[n] = grid size
};

hipError_t launch_set_desc(const PoseBatch& v, PoseBatch* dst, hipStream_t s);

hipError_t launch_graph_count(const Multi& m, float lig_r, int lig_cap, float cutoff, hipStream_t s);
hipError_t launch_graph_scan(const Multi& m, hipStream_t s);
hipError_t launch_graph_fill(const Multi& m_lig, const Multi& m_rec, float lig_r, int lig_cap, float cutoff, hipStream_t s);

// one launch over several edge lists, each with its own MLP weights / inputs / count
constexpr int EDGE_MLP_MAX_SEG = 2 * MAX_COSCHED;
struct EdgeSeg { EdgeMlp m; const float* dist; const float* bond4; const int* count; int cap; float* out; };
struct EdgeMlpArgs { EdgeSeg seg[EDGE_MLP_MAX_SEG]; int n; };
hipError_t launch_edge_mlp(const EdgeMlpArgs& a, hipStream_t s);

struct StepWeights {     // small dense weights used by step_prep / heads (device pointers, row-major [out][in])
  const float *rec_sig_w0, *rec_sig_b0, *rec_sig_w1, *rec_sig_b1;    // rec_sigma_embedding 32->32->32
  const float *lig_edge_w0, *lig_edge_b0;                            // [32][68]
  const float *cross_w0, *cross_b0;                                  // [32][64]
  const float *center_w0, *center_b0;                                // [32][64]
  const float *lig_node_w, *lig_node_b;                              // additional_features_embedder [32][64]
  const float *tr_w0, *tr_b0, *rot_w0, *rot_b0;                      // [32][33]
};
hipError_t launch_step_prep(const StepWeights& w, const StepVectors& v, const float* sigma_emb_dev, hipStream_t s);

// node feature / edge attribute initialisation of a step
hipError_t launch_lig_node_init(const Multi& m, const float* lig_node_c, int xi, hipStream_t s);          // X[xi] ligand rows
hipError_t launch_rec_time_init(const Multi& m, const float* rec_sigma_emb, int xi, hipStream_t s);       // X[xi] receptor rows + rr_attr_t

struct CenterHead {
  const float *ce_WgT, *ce_W1T, *ce_b1, *offset; float coeff;       // center_edge_embedding (gauss part / layer 2)
  const float *fc_w0, *fc_b0, *fc_w1, *fc_b1;                        // final_conv.fc: [64][64], [124][64]
  const float *bn_scale;                                             // [4] weight*rsqrt(var+eps)
  const float *tr_w0n, *tr_w1, *tr_b1, *rot_w0n, *rot_w1, *rot_b1;   // first-layer norm column [32], second layer [32], [1]
};
// centre convolution -> tr / rot scores of every batch (node features read from X[xi])
hipError_t launch_center_head(const CenterHead& h, const StepVectors& v, const Multi& m_atoms, const Multi& m_samples, int xi,
                              float tr_sigma, float rot_norm, hipStream_t s);

struct BondHead {
  EdgeMlp fe;                                                        // final_edge_embedding (part = b0)
  const float *fc_w0, *fc_b0, *fc_w1, *fc_b1;                        // tor_bond_conv.fc: [96][96], [384][96]
  const float *bn_scale, *bn_mean, *bn_bias;                         // [64] per output column
  const float *tf_w0, *tf_w1;                                        // tor_final_layer [32][64], [32]
};
struct SdeCoefs { float tr_s, tr_n, rot_s, rot_n, tor_s, tor_n; };
// perturbation + modify_conformer_batch of every batch for step `step` of the call (noise rows of that step from the descriptors;
// a noise term whose coefficient is 0 is skipped).  use_coefs = 0: tr/rot/tor outputs ARE the updates (cbd_modify_conformer).
hipError_t launch_pose_update(const Multi& m, int step, const SdeCoefs& coefs, int use_coefs, int with_torsion, int max_nl, hipStream_t s);

hipError_t launch_rec_node_embed(const float* rec_x, int Nr, int lm_dim, const float* emb_table, const float* w, const float* b,
                                 float* node, hipStream_t s);
hipError_t launch_edge_geom(const float* pos, const int* src, const int* dst, int n, float* vec4, float* dist, hipStream_t s);
hipError_t launch_symm_rmsd(int B, int N, int K, const float* pos, const float* ref, const int* idx_ref, const int* idx_pos, float* out,
                            int* argmin, hipStream_t s);
hipError_t launch_fill_i32(int* p, int v, int n, hipStream_t s);
hipError_t launch_node_proj(const ProjArgs& a, hipStream_t s);   // per-node part of the first Linear of a layer's FCBlocks

// tp_conv.hip
// torsion head on the matrix cores: neighbour search (one wave per rotatable bond) then bond_conv_kernel
hipError_t launch_bond_nb(const Multi& m, float lig_r, int cap, hipStream_t s);
hipError_t launch_bond_conv(const BondHead& h, const Multi& m, int xi, const float* wstream, float tor_norm_sqrt, hipStream_t s);
constexpr int BOND_CONV_TILES = 15;
hipError_t launch_tp_conv(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s);
hipError_t launch_tp_conv_bf16(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s);
#ifdef CBD_EXPERIMENTS
// the 0e-only slices of the bf16 role split (74 -> 74 layers) with LDS-resident weight tiles: persistent workgroups, one per CU
// (experiments/csrc/tp_conv_bf16p.hip, linked into the diagnostic library only)
hipError_t launch_tp_conv_bf16p(const ConvArgs& a, int n_wg, hipStream_t s);
#endif
// the 74 -> 74 layers of the bf16 policy with register-stationary weights: persistent workgroups of four waves, one per CU (tp_conv_bf16s.hip)
hipError_t launch_tp_conv_bf16s(const ConvArgs& a, int n_wg, hipStream_t s);
bool tp_conv_bf16s_fits(const ConvArgs& a);      // false: the launch does not fit that kernel (virtual slices, more FCBlocks than its role table holds) -> streaming kernel
hipError_t launch_tp_conv_x3(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s);     // bf16x3 weight streams
// segmented sum -> mean -> BatchNorm -> residual of one layer for every batch.  kind: which node types / group sets take part
enum FinKind { FIN_EMB = 0, FIN_FIRST = 1, FIN_MID = 2, FIN_LAST = 3,     // ligand embedding layer; interaction layer 0; 1..3; 4
               FIN_FIRST_R = 5, FIN_MID_R = 6, FIN_LAST_R = 7 };           // ... of the bf16 role split (fin_*_r group sets; diagnostic library only)
hipError_t launch_conv_finalize_multi(const Multi& m, int kind, int xi_in, int xi_out, const float* bn_scale, const float* bn_mean,
                                      const float* bn_bias, int in_dim, int out_dim, hipStream_t s);
// single list of nodes with explicit groups (receptor embedding at set-up time)
hipError_t launch_conv_finalize(const FinArgs& fa, const float* node_in, float* node_out, const float* bn_scale,
                                const float* bn_mean, const float* bn_bias, int n_nodes, int in_dim, int out_dim,
                                int node_off, hipStream_t s);

// host helper: Multi with per-batch workgroup counts
template <class F>
inline Multi make_multi(int n, const PoseBatch* const* descs, F blocks_of) {
  Multi m{};
  m.n = n;
  int o = 0;
  for (int k = 0; k < n; ++k) { m.d[k] = descs[k]; m.off[k] = o; o += blocks_of(k); }
  for (int k = n; k <= MAX_COSCHED; ++k) m.off[k] = o;
  return m;
}

}  // namespace cbd
