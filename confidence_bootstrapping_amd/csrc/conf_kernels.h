// Host-visible declarations of the graph / featurisation / head kernels of the all-atom confidence engine (gfx950).
#pragma once
#include "conf_common.h"

namespace cbd {

// Edge-group indices = order of the reference's joint edge list and of conv_layers[l].fc[g]
// (models/all_atom_score_model.py:409-421): lig-lig, lig->res, lig->atom, res-res, res->lig (flipped lr),
// res->atom (flipped ar), atom-atom, atom->lig (flipped la), atom->res.
enum ConfGroupId { G_LL = 0, G_LR, G_LA, G_RR, G_RL, G_RA, G_AA, G_AL, G_AR };

struct ConfStatic {       // one complex (device pointers), built by cbd_conf_set_complex
  int Nl, Nr, Na, nbd, Err, Eaa;
  const float* rec_pos;   // [Nr][3]
  const float* atom_pos;  // [Na][3]
  const int* atom_res;    // [Na] residue of every atom ('atom_rec_contact' row 1)
  const int* bond_row;    // [Nl+1] CSR of the bond list by its first atom
  const int* bond_dst;    // [nbd]
  const float* bond_attr; // [nbd][4]
  const int *rr_ptr, *rr_dst, *rr_eid;   // CSR of 'rec_contact' by aggregating residue (edge_index[0]); eid = original column
  const int *aa_ptr, *aa_dst, *aa_eid;   // CSR of 'atom_contact'
  const int *ra_ptr, *ra_atom;           // atoms of every residue
};

struct ConfDyn {          // one batch of poses (capacity-sized device buffers)
  const float* pos;       // [B][Nl][3]
  int* keep_res;          // [B][Nr] crop_beyond mask
  int* cnt[CONF_MAX_GROUPS];     // per aggregating node of the group's node type
  int* start[CONF_MAX_GROUPS];
  int* total;                    // [9]
  int* src[CONF_MAX_GROUPS];
  int* dst[CONF_MAX_GROUPS];
  int* aidx[CONF_MAX_GROUPS];
  float *ll_vec, *ll_dist, *ll_bond4;
  float *lr_vec, *lr_dist;
  float *la_vec, *la_dist;
  int* lr_pair;           // [B*Nl][Nr] edge id of (ligand atom, residue) or -1
  int* la_pair;           // [B*Nl][Na]
  int* overflow;          // set when a per-node cap was exceeded (the call then fails)
  int la_cap;             // max ligand->atom edges per ligand atom the buffers were sized for
};

struct ConfEdgeMlp {      // Linear(gauss[32] (+ bond one-hot[4])) -> ReLU -> Linear, constant inputs folded into b0 / b1
  const float* WgT;       // [32][24]
  const float* WbT;       // [4][24] or null
  const float* W1T;       // [24][24]
  const float* b0;        // [24]
  const float* b1;        // [24]
  const float* offset;    // [32] GaussianSmearing.offset buffer of the checkpoint
  float coeff;            // GaussianSmearing.coeff = -0.5 / (offset[1] - offset[0])^2
};

struct ConfHead {         // Linear-BN-ReLU-Linear-BN-ReLU-Linear with eval BatchNorm1d folded into scale/shift
  const float *W0, *s0, *t0;   // [24][in], [24], [24]:  h = relu((W0 x) * s0 + t0)
  const float *W1, *s1, *t1;   // [24][24]
  const float *W2, *b2;        // [out][24], [out]
  int in_dim, out_dim;
};

hipError_t conf_launch_keep(const ConfStatic& cs, const ConfDyn& cd, int B, float crop2, hipStream_t s);
hipError_t conf_launch_graph_lig(bool fill, const ConfStatic& cs, const ConfDyn& cd, int B, float lig_r2, int lig_cap, float cross_cut, hipStream_t s);
hipError_t conf_launch_graph_rec_atom(bool fill, const ConfStatic& cs, const ConfDyn& cd, int B, hipStream_t s);
hipError_t conf_launch_scan(const ConfDyn& cd, int g0, int g1, const int* n_nodes /*host [9]*/, hipStream_t s);
hipError_t conf_launch_edge_mlp(const ConfEdgeMlp& m, const float* dist, const float* bond4, const int* count, int cap, float* attr, hipStream_t s);
hipError_t conf_launch_static_geom(const float* pos_src, const float* pos_dst, const int* src, const int* dst, int n, float* vec, float* dist, hipStream_t s);
hipError_t conf_launch_node_embed(const float* x, int x_stride, int n_cat, const int* table_off, const float* tables, const float* W,
                                  int in_extra, const float* bias, int n, float* out, hipStream_t s);
hipError_t conf_launch_node_init(const float* lig_base, const float* rec_base, const float* atom_base, int B, int Nl, int Nr, int Na,
                                 float* node, hipStream_t s);
hipError_t conf_launch_heads(const ConfHead& atom_head, const ConfHead& conf_head, const float* node, int B, int Nl,
                             float* atom_conf, float* conf, hipStream_t s);

}  // namespace cbd
