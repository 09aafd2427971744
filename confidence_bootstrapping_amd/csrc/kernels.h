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

hipError_t launch_graph_count(const GraphStatic& gs, const GraphDyn& gd, int B, float lig_r, int lig_cap, float cutoff, hipStream_t s);
hipError_t launch_graph_scan(const GraphStatic& gs, const GraphDyn& gd, int B, unsigned long long* stats, hipStream_t s);
hipError_t launch_graph_fill(const GraphStatic& gs, const GraphDyn& gd, int B, float lig_r, int lig_cap, float cutoff, hipStream_t s);
hipError_t launch_edge_mlp(const EdgeMlp& m, const float* dist, const float* bond4, const int* count, int cap, float* out, hipStream_t s);

struct StepWeights {     // small dense weights used by step_prep / heads (device pointers, row-major [out][in])
  const float *rec_sig_w0, *rec_sig_b0, *rec_sig_w1, *rec_sig_b1;    // rec_sigma_embedding 32->32->32
  const float *lig_edge_w0, *lig_edge_b0;                            // [32][68]
  const float *cross_w0, *cross_b0;                                  // [32][64]
  const float *center_w0, *center_b0;                                // [32][64]
  const float *lig_node_w, *lig_node_b;                              // additional_features_embedder [32][64]
  const float *tr_w0, *tr_b0, *rot_w0, *rot_b0;                      // [32][33]
};
struct StepVectors {     // per-step device vectors, 32 floats each
  float *rec_sigma_emb, *ll_part, *lr_part, *center_part, *lig_node_c, *tr_part, *rot_part;
};
hipError_t launch_step_prep(const StepWeights& w, const StepVectors& v, const float* sigma_emb_dev, hipStream_t s);

// node feature initialisation
hipError_t launch_lig_node_init(const float* lig_static32, const float* lig_node_c, float* node, int B, int Nl, hipStream_t s);
hipError_t launch_rec_node_init(const float* rec_static, const float* rec_sigma_emb, float* node, int B, int rec_off, int Nr, hipStream_t s);
hipError_t launch_add_rows(const float* a, const float* v32, float* out, int rows, hipStream_t s);   // out[r][c] = a[r][c] + v[c], 32 wide

struct CenterHead {
  const float *ce_WgT, *ce_W1T, *ce_b1, *offset; float coeff;       // center_edge_embedding (gauss part / layer 2)
  const float *fc_w0, *fc_b0, *fc_w1, *fc_b1;                        // final_conv.fc: [64][64], [124][64]
  const float *bn_scale;                                             // [4] weight*rsqrt(var+eps)
  const float *tr_w0n, *tr_w1, *tr_b1, *rot_w0n, *rot_w1, *rot_b1;   // first-layer norm column [32], second layer [32], [1]
};
hipError_t launch_center_head(const CenterHead& h, const StepVectors& v, const float* pos, const float* node, int B, int Nl,
                              float tr_sigma, float rot_norm, float* tr_out, float* rot_out, float* dbg_global, float* msg_ws,
                              hipStream_t s);   // msg_ws: [B*Nl][12]

struct BondHead {
  EdgeMlp fe;                                                        // final_edge_embedding (part = b0)
  const float *fc_w0, *fc_b0, *fc_w1, *fc_b1;                        // tor_bond_conv.fc: [96][96], [384][96]
  const float *bn_scale, *bn_mean, *bn_bias;                         // [64] per output column
  const float *tf_w0, *tf_w1;                                        // tor_final_layer [32][64], [32]
};
struct SdeCoefs { float tr_s, tr_n, rot_s, rot_n, tor_s, tor_n; };
// perturbation (if scores != null) + modify_conformer_batch.  If coefs == null the tr/rot/tor inputs are the updates.
hipError_t launch_pose_update(const GraphStatic& gs, float* pos, int B, const float* tr, const float* rot, const float* tor,
                              const float* z_tr, const float* z_rot, const float* z_tor, const SdeCoefs* coefs, hipStream_t s);

hipError_t launch_rec_node_embed(const float* rec_x, int Nr, int lm_dim, const float* emb_table, const float* w, const float* b,
                                 float* node, hipStream_t s);
hipError_t launch_edge_geom(const float* pos, const int* src, const int* dst, int n, float* vec4, float* dist, hipStream_t s);
hipError_t launch_symm_rmsd(int B, int N, int K, const float* pos, const float* ref, const int* idx_ref, const int* idx_pos, float* out,
                            int* argmin, hipStream_t s);
hipError_t launch_fill_i32(int* p, int v, int n, hipStream_t s);
hipError_t launch_node_proj(const ProjArgs& a, hipStream_t s);   // per-node part of the first Linear of a layer's FCBlocks

// tp_conv.hip
// torsion head on the matrix cores (replaces the msg/final stages of launch_bond_head); nb/nb_cnt from launch_bond_nb
hipError_t launch_bond_nb(const GraphStatic& gs, const float* pos, int B, float lig_r, int cap, int* nb_ws, int* nb_cnt_ws, int* tor_edge_count, hipStream_t s);
hipError_t launch_bond_conv(const BondHead& h, const GraphStatic& gs, const float* pos, const float* node, int B, const int* nb,
                            const int* nb_cnt, const float* wstream, float tor_norm_sqrt, float* tor_out, float* dbg_feat, hipStream_t s);
constexpr int BOND_CONV_TILES = 15;
hipError_t launch_tp_conv(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s);
hipError_t launch_conv_finalize2(const FinArgs& fa0, int n0, int off0, const FinArgs& fa1, int n1, int off1, const float* node_in,
                                 float* node_out, const float* bn_scale, const float* bn_mean, const float* bn_bias, int in_dim,
                                 int out_dim, hipStream_t s);
hipError_t launch_tp_conv_bf16(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s);   // bf16 weight streams
hipError_t launch_tp_conv_x3(int in_level, int out_level, const ConvArgs& a, int grid, hipStream_t s);     // bf16x3 weight streams
hipError_t launch_conv_finalize(const FinArgs& fa, const float* node_in, float* node_out, const float* bn_scale,
                                const float* bn_mean, const float* bn_bias, int n_nodes, int in_dim, int out_dim,
                                int node_off, hipStream_t s);

}  // namespace cbd
