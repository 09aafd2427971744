// Host side of the all-atom CONFIDENCE engine: the cbd_conf_* entry points of include/cbdock.h.
// Owns the weights (FCBlocks re-packed into MFMA tile streams, constant t = 0 inputs folded), the per-complex static data
// (receptor graphs, pose-independent embeddings) and the per-batch workspace, and sequences the kernels of one forward pass.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>

#include "conf_kernels.h"
#include "host_util.h"

using namespace cbd;

namespace {

constexpr int LIG_N_CAT = 16, ATOM_N_CAT = 4;
constexpr int LA_CAP_PER_ATOM = 256;   // ligand->receptor-atom edges per ligand atom the workspace is sized for (5 A sphere)

struct CLayerDev {
  int in_level = 0, out_level = 0, n_groups = 0;
  float* wstream[CONF_MAX_GROUPS] = {};
  float *bn_scale = nullptr, *bn_mean = nullptr, *bn_bias = nullptr;   // [CN_STRIDE]
};

struct EmbedDev {   // AtomEncoder pieces
  float* tables = nullptr;
  int* table_off = nullptr;
  float* W = nullptr;      // [24][24 + in_extra] or null
  float* bias = nullptr;   // [24]
  int in_extra = 0;
};

}  // namespace

struct cbd_conf_engine {
  cbd_conf_config cfg{};
  std::map<std::string, HostTensor> host_w;
  bool weights_ready = false, complex_ready = false;
  int* overflow_dev = nullptr;      // sticky capacity-overflow flag (cbd_conf_create / cbd_conf_check)
  DevPool wpool, cpool, bpool;
  // ---- weights
  CLayerDev conv[5];
  ConfEdgeMlp m_ll{}, m_lr{}, m_la{}, m_rr{}, m_aa{}, m_ar{};
  ConfHead atom_head{}, conf_head{};
  EmbedDev emb_lig, emb_rec, emb_atom;
  // ---- complex
  ConfStatic cs{};
  float *lig_base = nullptr, *rec_base = nullptr, *atom_base = nullptr;
  float *rr_attr = nullptr, *rr_vec = nullptr, *aa_attr = nullptr, *aa_vec = nullptr, *ar_attr = nullptr, *ar_vec = nullptr;
  // ---- batch workspace
  ConfDyn cd{};
  int cap[CONF_MAX_GROUPS] = {};
  float *X0 = nullptr, *X1 = nullptr;
  float *ll_attr = nullptr, *lr_attr = nullptr, *la_attr = nullptr;
  float *fsum[CONF_MAX_GROUPS] = {}, *lsum[CONF_MAX_GROUPS] = {};
  float* racc[CONF_MAX_GROUPS] = {};      // [nodes of the group's type][84]
  float *atom_conf_scratch = nullptr;
  hipStream_t last_stream = nullptr;
  int last_B = 0;
  // ---- debug / timing
  bool keep_debug = false;
  std::map<std::string, std::vector<float>> dbg;
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
  double t_total_ms = 0;
  int64_t t_n = 0;
};

static const HostTensor* find_w(cbd_conf_engine* e, const std::string& k) {
  auto it = e->host_w.find(k);
  return it == e->host_w.end() ? nullptr : &it->second;
}

static int need(cbd_conf_engine* e, const std::string& k, std::initializer_list<int64_t> shape, const HostTensor** out) {
  const HostTensor* t = find_w(e, k);
  if (!t) return fail(CBD_ERR_WEIGHT, "missing tensor '%s' (load_state_dict strict=True)", k.c_str());
  if (t->shape != std::vector<int64_t>(shape)) return fail(CBD_ERR_WEIGHT, "tensor '%s' has an unexpected shape", k.c_str());
  *out = t;
  return 0;
}

// ============================================================================================= weight re-packing
// e3nn FullyConnectedTensorProduct(in, 1x0e+1x1o+1x2e, out, shared_weights=False): instructions in the order
// `for in: for sh: for out` (allowed by |l1-l2| <= l3 <= l1+l2, p3 = p1 p2), per-edge weights = the instructions'
// [mul_in][1][mul_out] blocks concatenated (reference models/tensor_layers.py:185 with e3nn 0.5.0).
namespace {
struct Irr { int l, p, mul; };
const Irr kKinds[4] = {{0, +1, CNS}, {1, -1, CNV}, {1, +1, CNV}, {0, -1, CNS}};   // 0e, 1o, 1e, 0o
const Irr kSh[3] = {{0, +1, 1}, {1, -1, 1}, {2, +1, 1}};

struct PathTable {
  int off[4][3][4];   // [in kind][sh l][out kind] -> weight offset or -1
  int numel;
};

PathTable fctp_paths(int IN, int OUT) {
  PathTable t;
  for (auto& a : t.off) for (auto& b : a) for (int& c : b) c = -1;
  const bool in_present[4] = {true, IN >= 1, IN >= 2, IN >= 3};
  const bool out_present[4] = {true, true, OUT >= 2, OUT >= 3};
  int run = 0;
  for (int i1 = 0; i1 < 4; ++i1) {
    if (!in_present[i1]) continue;
    for (int i2 = 0; i2 < 3; ++i2)
      for (int io = 0; io < 4; ++io) {
        if (!out_present[io]) continue;
        const Irr a = kKinds[i1], b = kSh[i2], c = kKinds[io];
        if (c.l < std::abs(a.l - b.l) || c.l > a.l + b.l || c.p != a.p * b.p) continue;
        t.off[i1][i2][io] = run;
        run += a.mul * c.mul;
      }
  }
  t.numel = run;
  return t;
}

struct MidSeg { int in_kind, sh_l, count; float alpha; };   // one path of a block in the kernel's mid-index order
}  // namespace

static std::vector<float> pack_fctp_stream(int IN, int OUT, const float* W1, const float* b1, const float* W2, const float* b2) {
  const FctpShape S = fctp_shape(IN, OUT);
  const PathTable P = fctp_paths(IN, OUT);
  std::vector<float> out(fctp_stream_floats(S.ntiles), 0.f);
  float* const bias_tab = out.data() + (size_t)(S.ntiles + 1) * CTILE_W_FLOATS;
  auto widx = [](int s, int lane) { return ((s >> 2) * 64 + lane) * 4 + (s & 3); };
  int T = 0;
  for (int m = 0; m < 3; ++m, ++T) {   // first Linear: hidden units 32m .. 32m+31 (rows >= 72 are zero)
    float* tile = out.data() + (size_t)T * CTILE_W_FLOATS;
    for (int s = 0; s < CKSTEPS; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int i = lane & 31, h = lane >> 5, row = 32 * m + i;
        const int f = CNS * (s / 12) + 12 * h + (s % 12);
        tile[widx(s, lane)] = row < CKDIM ? W1[(size_t)row * CKDIM + f] : 0.f;
      }
    for (int r = 0; r < 32; ++r) bias_tab[(size_t)T * 32 + r] = 32 * m + r < CKDIM ? b1[32 * m + r] : 0.f;
  }
  auto kperm = [](int s, int h) { return s < 32 ? 32 * (s / 16) + ((s % 16) & 3) + 8 * ((s % 16) >> 2) + 4 * h : 64 + (s - 32) + 4 * h; };
  auto fill_tile = [&](int Tt, const int* wc, const float* scale) {   // wc[r] < 0 => zero row
    float* tile = out.data() + (size_t)Tt * CTILE_W_FLOATS;
    for (int s = 0; s < CKSTEPS; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int r = lane & 31, h = lane >> 5;
        tile[widx(s, lane)] = wc[r] < 0 ? 0.f : scale[r] * W2[(size_t)wc[r] * CKDIM + kperm(s, h)];
      }
    for (int r = 0; r < 32; ++r) bias_tab[(size_t)Tt * 32 + r] = wc[r] < 0 ? 0.f : scale[r] * b2[wc[r]];
  };
  // weight column and folded coefficient of (block = out kind, mid index i, output w)
  const float s3 = std::sqrt(3.0f), s15 = std::sqrt(1.5f), s45 = 3.0f / std::sqrt(2.0f);
  auto column = [&](int out_kind, const std::vector<MidSeg>& segs, int fan, int i, int w, int* col, float* coef) {
    int base = 0;
    for (const MidSeg& sg : segs) {
      if (i < base + sg.count) {
        const int u = i - base;
        *col = P.off[sg.in_kind][sg.sh_l][out_kind] + u * kKinds[out_kind].mul + w;
        *coef = sg.alpha / std::sqrt((float)fan);
        return true;
      }
      base += sg.count;
    }
    return false;
  };
  int wc[32];
  float sc[32];
  const std::vector<MidSeg> segs0e = {{0, 0, CNS, 1.f}, {1, 1, S.n1o, 1.f}};
  auto scalar_block = [&](int out_kind, const std::vector<MidSeg>& segs, int fan, int ngroups) {
    for (int g = 0; g < ngroups; ++g) {
      const bool dense = sc_tail_dense(fan, g);       // tail of <= 2 mids: two denser tiles (conf_common.h)
      // merged tails (FctpShape::merged): block 0e has no tile B of its own; its rows are slots 2, 3 of block 0o's tile B
      const int ntile_g = dense ? (S.merged && out_kind == 0 ? 1 : 2) : 3;
      for (int q = 0; q < ntile_g; ++q, ++T) {
        for (int r = 0; r < 32; ++r) {
          const int slot = r >> 3;
          int i = C_SC_TILE_I * g + slot, w = 8 * q + (r & 7);
          bool ok;
          if (dense && q == 1 && slot >= 2) {
            if (S.merged && out_kind == 3) {
              const int g0 = S.g0e - 1;
              ok = column(0, segs0e, S.fan0e, C_SC_TILE_I * g0 + (slot & 1), 16 + (r & 7), &wc[r], &sc[r]);
            } else ok = false;                        // zero rows
          } else {
            if (dense) {
              i = C_SC_TILE_I * g + (slot & 1);
              w = (q == 0 ? 8 * (slot >> 1) : 16) + (r & 7);
            }
            ok = column(out_kind, segs, fan, i, w, &wc[r], &sc[r]);
          }
          if (!ok) { wc[r] = -1; sc[r] = 0.f; }
        }
        fill_tile(T, wc, sc);
      }
    }
  };
  const std::vector<MidSeg> segs1o = {{0, 1, CNS, s3}, {1, 0, S.n1o, 1.f}, {1, 2, S.n1o, s45}, {2, 1, S.n1e, s15}};
  auto vector_block = [&](int out_kind, const std::vector<MidSeg>& segs, int fan, int ntile) {
    // merged tails (FctpShape::vmerged): block 1o stops one tile early; its tail mids sit behind block 1e's in 1e's last tile
    const int own = S.vmerged && out_kind == 1 ? ntile - 1 : ntile;
    const int r1o = S.fan1o % C_VEC_TILE_I, r1e = S.fan1e % C_VEC_TILE_I;
    for (int t = 0; t < own; ++t, ++T) {
      for (int r = 0; r < 32; ++r) {
        const int reg = (r & 3) + 4 * (r >> 3), hf = (r >> 2) & 1;
        const int q = reg / 3, w = 3 * hf + reg % 3;
        bool ok = false;
        if (reg < 15) {
          if (S.vmerged && out_kind == 2 && t == ntile - 1 && q >= r1e)
            ok = q - r1e < r1o && column(1, segs1o, S.fan1o, C_VEC_TILE_I * (S.t1o - 1) + (q - r1e), w, &wc[r], &sc[r]);
          else
            ok = column(out_kind, segs, fan, C_VEC_TILE_I * t + q, w, &wc[r], &sc[r]);
        }
        if (!ok) { wc[r] = -1; sc[r] = 0.f; }
      }
      fill_tile(T, wc, sc);
    }
  };
  scalar_block(0, segs0e, S.fan0e, S.g0e);
  vector_block(1, segs1o, S.fan1o, S.t1o);
  if (OUT >= 2) vector_block(2, {{1, 1, S.n1o, s15}, {2, 0, S.n1e, 1.f}, {2, 2, S.n1e, s45}, {3, 1, S.n0o, s3}}, S.fan1e, S.t1e);
  if (OUT >= 3) scalar_block(3, {{2, 1, S.n1e, 1.f}, {3, 0, S.n0o, 1.f}}, S.fan0o, S.g0o);
  return out;
}

static int layer_levels(int l, int* IN, int* OUT) {
  *IN = std::min(l, 3);
  *OUT = std::min(l + 1, 3);
  return 0;
}

static int build_conv_layer(cbd_conf_engine* e, int l, int groups, CLayerDev* L) {
  int IN, OUT;
  layer_levels(l, &IN, &OUT);
  const FctpShape S = fctp_shape(IN, OUT);
  if (fctp_paths(IN, OUT).numel != S.weight_numel) return fail(CBD_ERR_WEIGHT, "internal: path table / shape mismatch");
  L->in_level = IN; L->out_level = OUT; L->n_groups = groups;
  const std::string prefix = "conv_layers." + std::to_string(l);
  for (int g = 0; g < groups; ++g) {
    const std::string fc = prefix + ".fc." + std::to_string(g);
    const HostTensor *w0, *b0, *w1, *b1;
    CHK(need(e, fc + ".0.weight", {CKDIM, CKDIM}, &w0));
    CHK(need(e, fc + ".0.bias", {CKDIM}, &b0));
    CHK(need(e, fc + ".3.weight", {S.weight_numel, CKDIM}, &w1));
    CHK(need(e, fc + ".3.bias", {S.weight_numel}, &b1));
    HIPCHK(e->wpool.upload(&L->wstream[g], pack_fctp_stream(IN, OUT, w0->data.data(), b0->data.data(), w1->data.data(), b1->data.data())));
  }
  const int n0o = OUT >= 3 ? CNS : 0, n1e = OUT >= 2 ? CNV : 0;
  const int nf = CNS + CNV + n1e + n0o;
  const HostTensor *bw, *bb, *bm, *bv;
  CHK(need(e, prefix + ".batch_norm.weight", {nf}, &bw));
  CHK(need(e, prefix + ".batch_norm.bias", {CNS}, &bb));
  CHK(need(e, prefix + ".batch_norm.running_mean", {CNS}, &bm));
  CHK(need(e, prefix + ".batch_norm.running_var", {nf}, &bv));
  std::vector<float> sc(CN_STRIDE, 0.f), mean(CN_STRIDE, 0.f), bias(CN_STRIDE, 0.f);
  for (int c = 0; c < S.out_dim; ++c) {
    int chn;
    if (c < CC_1O) chn = c;
    else if (c < CC_1E) chn = CNS + (c - CC_1O) / 3;
    else if (c < CC_0O) chn = CNS + CNV + (c - CC_1E) / 3;
    else chn = CNS + 2 * CNV + (c - CC_0O);
    sc[c] = bw->data[chn] * (1.0f / std::sqrt(bv->data[chn] + 1e-5f));
    if (c < CNS) { mean[c] = bm->data[c]; bias[c] = bb->data[c]; }
  }
  HIPCHK(e->wpool.upload(&L->bn_scale, sc));
  HIPCHK(e->wpool.upload(&L->bn_mean, mean));
  HIPCHK(e->wpool.upload(&L->bn_bias, bias));
  return 0;
}

// sinusoidal_embedding(embedding_scale * 0, 32) = [sin(0) x16, cos(0) x16] (utils/diffusion_utils.py:99-110)
static void sigma_emb_t0(float (&s)[32]) {
  for (int k = 0; k < 32; ++k) s[k] = k < 16 ? 0.f : 1.f;
}

// edge embedding MLP `prefix` = Linear(in_dim, 24) ReLU Dropout Linear(24, 24); input layout [bond(4)?][sigma(32)?][gauss(32)]
static int build_edge_mlp(cbd_conf_engine* e, const std::string& prefix, bool has_bond, bool has_sigma, const std::string& offset_key,
                          const float* add_out /*[24] or null*/, ConfEdgeMlp* M) {
  const int in_dim = (has_bond ? 4 : 0) + (has_sigma ? 32 : 0) + 32;
  const HostTensor *w0, *b0, *w1, *b1, *off;
  CHK(need(e, prefix + ".0.weight", {CNS, in_dim}, &w0));
  CHK(need(e, prefix + ".0.bias", {CNS}, &b0));
  CHK(need(e, prefix + ".3.weight", {CNS, CNS}, &w1));
  CHK(need(e, prefix + ".3.bias", {CNS}, &b1));
  CHK(need(e, offset_key, {32}, &off));
  const int sig0 = has_bond ? 4 : 0, g0 = sig0 + (has_sigma ? 32 : 0);
  float sig[32];
  sigma_emb_t0(sig);
  std::vector<float> WgT(32 * CNS), WbT(4 * CNS), W1T(CNS * CNS), B0(CNS), B1(CNS);
  for (int o = 0; o < CNS; ++o) {
    float b = b0->data[o];
    if (has_sigma)
      for (int k = 0; k < 32; ++k) b += w0->data[(size_t)o * in_dim + sig0 + k] * sig[k];
    B0[o] = b;
    B1[o] = b1->data[o] + (add_out ? add_out[o] : 0.f);
    for (int k = 0; k < 32; ++k) WgT[k * CNS + o] = w0->data[(size_t)o * in_dim + g0 + k];
    if (has_bond)
      for (int k = 0; k < 4; ++k) WbT[k * CNS + o] = w0->data[(size_t)o * in_dim + k];
    for (int k = 0; k < CNS; ++k) W1T[k * CNS + o] = w1->data[(size_t)o * CNS + k];
  }
  float *d_WgT, *d_WbT = nullptr, *d_W1T, *d_b0, *d_b1, *d_off;
  HIPCHK(e->wpool.upload(&d_WgT, WgT));
  if (has_bond) HIPCHK(e->wpool.upload(&d_WbT, WbT));
  HIPCHK(e->wpool.upload(&d_W1T, W1T));
  HIPCHK(e->wpool.upload(&d_b0, B0));
  HIPCHK(e->wpool.upload(&d_b1, B1));
  HIPCHK(e->wpool.upload(&d_off, off->data));
  const double step = (double)off->data[1] - (double)off->data[0];   // GaussianSmearing.coeff, models/score_model.py:672
  *M = ConfEdgeMlp{d_WgT, d_WbT, d_W1T, d_b0, d_b1, d_off, (float)(-0.5 / (step * step))};
  return 0;
}

static int build_embed(cbd_conf_engine* e, const std::string& prefix, const std::vector<int>& dims, int in_extra_folded, int in_extra_live,
                       const float* extra_const /*[in_extra_folded]*/, const float* add_out /*[24] or null*/, EmbedDev* E) {
  std::vector<float> tables;
  std::vector<int> off;
  for (size_t i = 0; i < dims.size(); ++i) {
    const HostTensor* t;
    CHK(need(e, prefix + ".atom_embedding_list." + std::to_string(i) + ".weight", {dims[i], CNS}, &t));
    off.push_back((int)tables.size());
    tables.insert(tables.end(), t->data.begin(), t->data.end());
  }
  HIPCHK(e->wpool.upload(&E->tables, tables));
  HIPCHK(e->wpool.upload(&E->table_off, off));
  E->in_extra = in_extra_live;
  std::vector<float> bias(CNS, 0.f);
  if (in_extra_folded + in_extra_live > 0) {
    const int K = CNS + in_extra_folded + in_extra_live;
    const HostTensor *w, *b;
    CHK(need(e, prefix + ".additional_features_embedder.weight", {CNS, K}, &w));
    CHK(need(e, prefix + ".additional_features_embedder.bias", {CNS}, &b));
    // layout of the Linear input: [embedding(24) | extra...]; folded (constant) extras come first when present
    std::vector<float> W((size_t)CNS * (CNS + in_extra_live));
    for (int o = 0; o < CNS; ++o) {
      float bb = b->data[o];
      for (int k = 0; k < in_extra_folded; ++k) bb += w->data[(size_t)o * K + CNS + k] * extra_const[k];
      bias[o] = bb;
      for (int k = 0; k < CNS; ++k) W[(size_t)o * (CNS + in_extra_live) + k] = w->data[(size_t)o * K + k];
      for (int k = 0; k < in_extra_live; ++k) W[(size_t)o * (CNS + in_extra_live) + CNS + k] = w->data[(size_t)o * K + CNS + in_extra_folded + k];
    }
    HIPCHK(e->wpool.upload(&E->W, W));
  }
  if (add_out)
    for (int o = 0; o < CNS; ++o) bias[o] += add_out[o];
  HIPCHK(e->wpool.upload(&E->bias, bias));
  return 0;
}

// Linear-BatchNorm1d(eval)-ReLU-Dropout-Linear-BatchNorm1d-ReLU-Dropout-Linear (all_atom_score_model.py:203-226)
static int build_head(cbd_conf_engine* e, const std::string& prefix, int in_dim, int out_dim, ConfHead* H) {
  const HostTensor *w0, *b0, *w1, *b1, *w2, *b2;
  CHK(need(e, prefix + ".0.weight", {CNS, in_dim}, &w0));
  CHK(need(e, prefix + ".0.bias", {CNS}, &b0));
  CHK(need(e, prefix + ".4.weight", {CNS, CNS}, &w1));
  CHK(need(e, prefix + ".4.bias", {CNS}, &b1));
  CHK(need(e, prefix + ".8.weight", {out_dim, CNS}, &w2));
  CHK(need(e, prefix + ".8.bias", {out_dim}, &b2));
  auto fold = [&](const std::string& bn, const HostTensor* lin_b, std::vector<float>* s, std::vector<float>* t) -> int {
    const HostTensor *g, *bt, *m, *v;
    CHK(need(e, bn + ".weight", {CNS}, &g));
    CHK(need(e, bn + ".bias", {CNS}, &bt));
    CHK(need(e, bn + ".running_mean", {CNS}, &m));
    CHK(need(e, bn + ".running_var", {CNS}, &v));
    s->resize(CNS); t->resize(CNS);
    for (int o = 0; o < CNS; ++o) {
      const float sc = g->data[o] / std::sqrt(v->data[o] + 1e-5f);
      (*s)[o] = sc;
      (*t)[o] = (lin_b->data[o] - m->data[o]) * sc + bt->data[o];
    }
    return 0;
  };
  std::vector<float> s0, t0, s1, t1;
  CHK(fold(prefix + ".1", b0, &s0, &t0));
  CHK(fold(prefix + ".5", b1, &s1, &t1));
  float *dW0, *ds0, *dt0, *dW1, *ds1, *dt1, *dW2, *db2;
  HIPCHK(e->wpool.upload(&dW0, w0->data)); HIPCHK(e->wpool.upload(&ds0, s0)); HIPCHK(e->wpool.upload(&dt0, t0));
  HIPCHK(e->wpool.upload(&dW1, w1->data)); HIPCHK(e->wpool.upload(&ds1, s1)); HIPCHK(e->wpool.upload(&dt1, t1));
  HIPCHK(e->wpool.upload(&dW2, w2->data)); HIPCHK(e->wpool.upload(&db2, b2->data));
  *H = ConfHead{dW0, ds0, dt0, dW1, ds1, dt1, dW2, db2, in_dim, out_dim};
  return 0;
}

// CSR of an edge list by its row 0 (aggregating node), stable in the original column order
static void csr_by_row0(const int64_t* ei, int E, int n_nodes, std::vector<int>* ptr, std::vector<int>* dst, std::vector<int>* eid) {
  ptr->assign(n_nodes + 1, 0);
  for (int k = 0; k < E; ++k) (*ptr)[ei[k] + 1]++;
  for (int i = 0; i < n_nodes; ++i) (*ptr)[i + 1] += (*ptr)[i];
  dst->resize(E); eid->resize(E);
  std::vector<int> fillp(ptr->begin(), ptr->end() - 1);
  for (int k = 0; k < E; ++k) {
    const int p = fillp[ei[k]]++;
    (*dst)[p] = (int)ei[E + k];
    (*eid)[p] = k;
  }
}

// ================================================================================================================ C ABI
extern "C" {

int cbd_conf_create(const cbd_conf_config* cfg, cbd_conf_engine** out) {
  if (!cfg || !out) return fail(CBD_ERR_ARG, "null argument");
  if (cfg->ns != CNS || cfg->nv != CNV || cfg->num_conv_layers != 5)
    return fail(CBD_ERR_ARG, "the confidence engine covers ns=24, nv=6, 5 conv layers (pretrained_confidence) only");
  if (cfg->lm_embedding_dim != 0 && cfg->lm_embedding_dim != 1280) return fail(CBD_ERR_ARG, "lm_embedding_dim must be 0 or 1280");
  if (cfg->max_batch < 1) return fail(CBD_ERR_ARG, "max_batch must be >= 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(CBD_ERR_HIP, "no HIP device available: the confidence engine has no CPU path");
  HIPCHK(hipSetDevice(cfg->device));
  *out = new cbd_conf_engine();
  (*out)->cfg = *cfg;
  // capacity-overflow flag: lives as long as the engine and is STICKY -- set by any cbd_conf_score since the last cbd_conf_check, so a
  // caller may score several batches (of several complexes) and check once (sampling(): no host sync per complex)
  HIPCHK(hipMalloc(reinterpret_cast<void**>(&(*out)->overflow_dev), sizeof(int)));
  HIPCHK(hipMemset((*out)->overflow_dev, 0, sizeof(int)));
  return 0;
}

int cbd_conf_destroy(cbd_conf_engine* e) {
  if (!e) return 0;
  (void)hipSetDevice(e->cfg.device);
  (void)hipDeviceSynchronize();
  for (auto& p : e->ev_pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  e->bpool.release(); e->cpool.release(); e->wpool.release();
  if (e->overflow_dev) (void)hipFree(e->overflow_dev);
  delete e;
  return 0;
}

int cbd_conf_load_weight(cbd_conf_engine* e, const char* name, const float* data, const int64_t* shape, int32_t ndim) {
  if (!e || !name || (!data && ndim > 0)) return fail(CBD_ERR_ARG, "null argument");
  HostTensor t;
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
  t.data.assign(data, data + n);
  e->host_w[name] = std::move(t);
  e->weights_ready = false;
  return 0;
}

int cbd_conf_finalize_weights(cbd_conf_engine* e) {
  if (!e) return fail(CBD_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  e->wpool.release();
  for (int l = 0; l < 5; ++l) CHK(build_conv_layer(e, l, l == 4 ? 3 : 9, &e->conv[l]));
  // rec_sigma_embedding(sinusoidal(0)) is a constant vector added to the residue/atom scalars and to their edge attributes
  float sig[32];
  sigma_emb_t0(sig);
  float rec_sigma[CNS];
  {
    const HostTensor *w0, *b0, *w1, *b1;
    CHK(need(e, "rec_sigma_embedding.0.weight", {CNS, 32}, &w0));
    CHK(need(e, "rec_sigma_embedding.0.bias", {CNS}, &b0));
    CHK(need(e, "rec_sigma_embedding.3.weight", {CNS, CNS}, &w1));
    CHK(need(e, "rec_sigma_embedding.3.bias", {CNS}, &b1));
    float h[CNS];
    for (int o = 0; o < CNS; ++o) {
      float a = b0->data[o];
      for (int k = 0; k < 32; ++k) a += w0->data[o * 32 + k] * sig[k];
      h[o] = std::max(a, 0.f);
    }
    for (int o = 0; o < CNS; ++o) {
      float a = b1->data[o];
      for (int k = 0; k < CNS; ++k) a += w1->data[o * CNS + k] * h[k];
      rec_sigma[o] = a;
    }
  }
  CHK(build_edge_mlp(e, "lig_edge_embedding", true, true, "lig_distance_expansion.offset", nullptr, &e->m_ll));
  CHK(build_edge_mlp(e, "lr_edge_embedding", false, true, "cross_distance_expansion.offset", nullptr, &e->m_lr));
  CHK(build_edge_mlp(e, "la_edge_embedding", false, true, "lig_distance_expansion.offset", nullptr, &e->m_la));
  CHK(build_edge_mlp(e, "rec_edge_embedding", false, false, "rec_distance_expansion.offset", rec_sigma, &e->m_rr));
  CHK(build_edge_mlp(e, "atom_edge_embedding", false, false, "lig_distance_expansion.offset", rec_sigma, &e->m_aa));
  CHK(build_edge_mlp(e, "ar_edge_embedding", false, false, "rec_distance_expansion.offset", rec_sigma, &e->m_ar));
  CHK(build_embed(e, "lig_node_embedding", {119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2}, 32, 0, sig, nullptr, &e->emb_lig));
  CHK(build_embed(e, "rec_node_embedding", {38}, 0, e->cfg.lm_embedding_dim, nullptr, rec_sigma, &e->emb_rec));
  CHK(build_embed(e, "atom_node_embedding", {38, 119, 23, 38}, 0, 0, nullptr, rec_sigma, &e->emb_atom));
  CHK(build_head(e, "atom_confidence_predictor", 2 * CNS, 1 + CNS, &e->atom_head));
  CHK(build_head(e, "confidence_predictor", CNS, 1, &e->conf_head));
  e->weights_ready = true;
  return 0;
}

int cbd_conf_set_complex(cbd_conf_engine* e, int32_t Nl, int32_t Nr, int32_t Na, int32_t nbd, int32_t Err, int32_t Eaa,
                         const float* lig_x, const int64_t* bond_index, const float* bond_attr, const float* rec_x,
                         const float* rec_pos, const int64_t* rec_ei, const float* atom_x, const float* atom_pos,
                         const int64_t* atom_ei, const int64_t* atom_res) {
  if (!e) return fail(CBD_ERR_ARG, "null engine");
  if (!e->weights_ready) return fail(CBD_ERR_STATE, "cbd_conf_finalize_weights has not been called");
  if (Nl < 1 || Nr < 1 || Na < 1) return fail(CBD_ERR_ARG, "empty ligand / receptor");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  e->complex_ready = false;
  e->cpool.reset();
  e->bpool.reset();
  const int B = e->cfg.max_batch;
  for (int k = 0; k < 2 * nbd; ++k)
    if (bond_index[k] < 0 || bond_index[k] >= Nl) return fail(CBD_ERR_ARG, "ligand bond index out of range");
  for (int k = 0; k < 2 * Err; ++k)
    if (rec_ei[k] < 0 || rec_ei[k] >= Nr) return fail(CBD_ERR_ARG, "receptor edge index out of range");
  for (int k = 0; k < 2 * Eaa; ++k)
    if (atom_ei[k] < 0 || atom_ei[k] >= Na) return fail(CBD_ERR_ARG, "atom edge index out of range");
  for (int k = 0; k < Na; ++k)
    if (atom_res[k] < 0 || atom_res[k] >= Nr) return fail(CBD_ERR_ARG, "atom -> residue index out of range");
  {  // categorical features index embedding tables on the device: validate them here
    static const int lig_dims[LIG_N_CAT] = {119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2};
    static const int atom_dims[ATOM_N_CAT] = {38, 119, 23, 38};
    for (int a = 0; a < Nl; ++a)
      for (int f = 0; f < LIG_N_CAT; ++f) {
        const float v = lig_x[(size_t)a * LIG_N_CAT + f];
        if (!(v >= 0 && v < lig_dims[f]) || v != (float)(int)v) return fail(CBD_ERR_ARG, "ligand feature %d of atom %d out of range", f, a);
      }
    for (int a = 0; a < Na; ++a)
      for (int f = 0; f < ATOM_N_CAT; ++f) {
        const float v = atom_x[(size_t)a * ATOM_N_CAT + f];
        if (!(v >= 0 && v < atom_dims[f]) || v != (float)(int)v) return fail(CBD_ERR_ARG, "receptor-atom feature %d of atom %d out of range", f, a);
      }
    const int stride = 1 + e->cfg.lm_embedding_dim;
    for (int r = 0; r < Nr; ++r) {
      const float v = rec_x[(size_t)r * stride];
      if (!(v >= 0 && v < 38) || v != (float)(int)v) return fail(CBD_ERR_ARG, "residue type of residue %d out of range", r);
    }
  }

  ConfStatic& cs = e->cs;
  cs = ConfStatic{};
  cs.Nl = Nl; cs.Nr = Nr; cs.Na = Na; cs.nbd = nbd; cs.Err = Err; cs.Eaa = Eaa;
  DevPool& cp = e->cpool;
  // ---- positions, maps, CSR forms
  float *d_rec_pos, *d_atom_pos;
  HIPCHK(cp.upload(&d_rec_pos, std::vector<float>(rec_pos, rec_pos + 3 * (size_t)Nr)));
  HIPCHK(cp.upload(&d_atom_pos, std::vector<float>(atom_pos, atom_pos + 3 * (size_t)Na)));
  cs.rec_pos = d_rec_pos; cs.atom_pos = d_atom_pos;
  std::vector<int> ares(Na);
  for (int k = 0; k < Na; ++k) ares[k] = (int)atom_res[k];
  int* d_ares;
  HIPCHK(cp.upload(&d_ares, ares));
  cs.atom_res = d_ares;
  {
    std::vector<int> ptr, dst, eid;
    csr_by_row0(bond_index, nbd, Nl, &ptr, &dst, &eid);
    std::vector<float> battr((size_t)nbd * 4);
    for (int p = 0; p < nbd; ++p)
      for (int c = 0; c < 4; ++c) battr[(size_t)p * 4 + c] = bond_attr[(size_t)eid[p] * 4 + c];
    int *d_ptr, *d_dst; float* d_attr;
    HIPCHK(cp.upload(&d_ptr, ptr)); HIPCHK(cp.upload(&d_dst, dst)); HIPCHK(cp.upload(&d_attr, battr));
    cs.bond_row = d_ptr; cs.bond_dst = d_dst; cs.bond_attr = d_attr;
  }
  auto upload_csr = [&](const int64_t* ei, int E, int n, const int** p_ptr, const int** p_dst, const int** p_eid) -> int {
    std::vector<int> ptr, dst, eid;
    csr_by_row0(ei, E, n, &ptr, &dst, &eid);
    int *a, *b, *c;
    HIPCHK(cp.upload(&a, ptr)); HIPCHK(cp.upload(&b, dst)); HIPCHK(cp.upload(&c, eid));
    *p_ptr = a; *p_dst = b; *p_eid = c;
    return 0;
  };
  CHK(upload_csr(rec_ei, Err, Nr, &cs.rr_ptr, &cs.rr_dst, &cs.rr_eid));
  CHK(upload_csr(atom_ei, Eaa, Na, &cs.aa_ptr, &cs.aa_dst, &cs.aa_eid));
  {
    std::vector<int> ptr(Nr + 1, 0), atoms(Na);
    for (int k = 0; k < Na; ++k) ptr[ares[k] + 1]++;
    for (int r = 0; r < Nr; ++r) ptr[r + 1] += ptr[r];
    std::vector<int> fp(ptr.begin(), ptr.end() - 1);
    for (int k = 0; k < Na; ++k) atoms[fp[ares[k]]++] = k;
    int *a, *b;
    HIPCHK(cp.upload(&a, ptr)); HIPCHK(cp.upload(&b, atoms));
    cs.ra_ptr = a; cs.ra_atom = b;
  }
  // ---- pose-independent embeddings (t = 0): node scalars and the stored graphs' edge attributes / unit vectors
  hipStream_t s = nullptr;
  {
    const int lm = e->cfg.lm_embedding_dim;
    float *d_lx, *d_rx, *d_ax;
    HIPCHK(cp.upload(&d_lx, std::vector<float>(lig_x, lig_x + (size_t)Nl * LIG_N_CAT)));
    HIPCHK(cp.upload(&d_rx, std::vector<float>(rec_x, rec_x + (size_t)Nr * (1 + lm))));
    HIPCHK(cp.upload(&d_ax, std::vector<float>(atom_x, atom_x + (size_t)Na * ATOM_N_CAT)));
    HIPCHK(cp.alloc(&e->lig_base, (size_t)Nl * CNS));
    HIPCHK(cp.alloc(&e->rec_base, (size_t)Nr * CNS));
    HIPCHK(cp.alloc(&e->atom_base, (size_t)Na * CNS));
    HIPCHK(conf_launch_node_embed(d_lx, LIG_N_CAT, LIG_N_CAT, e->emb_lig.table_off, e->emb_lig.tables, e->emb_lig.W, 0, e->emb_lig.bias, Nl, e->lig_base, s));
    HIPCHK(conf_launch_node_embed(d_rx, 1 + lm, 1, e->emb_rec.table_off, e->emb_rec.tables, e->emb_rec.W, lm, e->emb_rec.bias, Nr, e->rec_base, s));
    HIPCHK(conf_launch_node_embed(d_ax, ATOM_N_CAT, ATOM_N_CAT, e->emb_atom.table_off, e->emb_atom.tables, nullptr, 0, e->emb_atom.bias, Na, e->atom_base, s));
  }
  auto static_edges = [&](const int64_t* ei, int E, const float* pos_src, const float* pos_dst, const ConfEdgeMlp& m, float** attr, float** vec) -> int {
    std::vector<int> src(E), dst(E);
    for (int k = 0; k < E; ++k) { src[k] = (int)ei[k]; dst[k] = (int)ei[E + k]; }
    int *d_src, *d_dst; float* d_dist;
    HIPCHK(cp.upload(&d_src, src)); HIPCHK(cp.upload(&d_dst, dst));
    HIPCHK(cp.alloc(&d_dist, (size_t)E));
    HIPCHK(cp.alloc(vec, (size_t)E * 4));
    HIPCHK(cp.alloc(attr, (size_t)E * CNS));
    HIPCHK(conf_launch_static_geom(pos_src, pos_dst, d_src, d_dst, E, *vec, d_dist, s));
    HIPCHK(conf_launch_edge_mlp(m, d_dist, nullptr, nullptr, E, *attr, s));
    return 0;
  };
  CHK(static_edges(rec_ei, Err, d_rec_pos, d_rec_pos, e->m_rr, &e->rr_attr, &e->rr_vec));
  CHK(static_edges(atom_ei, Eaa, d_atom_pos, d_atom_pos, e->m_aa, &e->aa_attr, &e->aa_vec));
  {
    std::vector<int64_t> ar(2 * (size_t)Na);
    for (int k = 0; k < Na; ++k) { ar[k] = k; ar[Na + k] = atom_res[k]; }
    CHK(static_edges(ar.data(), Na, d_atom_pos, d_rec_pos, e->m_ar, &e->ar_attr, &e->ar_vec));
  }

  // ---- batch workspace (capacity = max_batch poses)
  DevPool& bp = e->bpool;
  ConfDyn& cd = e->cd;
  cd = ConfDyn{};
  cd.la_cap = std::min(Na, LA_CAP_PER_ATOM);
  const int lcap = e->cfg.lig_radius_cap;
  const size_t nL = (size_t)B * Nl, nR = (size_t)B * Nr, nA = (size_t)B * Na;
  const size_t caps[CONF_MAX_GROUPS] = {
      (size_t)B * (nbd + (size_t)Nl * std::min(Nl - 1, lcap + 1)), nL * Nr, nL * cd.la_cap, (size_t)B * Err, nL * Nr, nA,
      (size_t)B * Eaa, nL * cd.la_cap, nA};
  const size_t nodes_of[CONF_MAX_GROUPS] = {nL, nL, nL, nR, nR, nR, nA, nA, nA};
  for (int g = 0; g < CONF_MAX_GROUPS; ++g) {
    if (caps[g] > (size_t)std::numeric_limits<int>::max() / 2) return fail(CBD_ERR_CAPACITY, "edge capacity of group %d overflows int32", g);
    e->cap[g] = (int)caps[g];
    HIPCHK(bp.alloc(&cd.cnt[g], nodes_of[g]));
    HIPCHK(bp.alloc(&cd.start[g], nodes_of[g]));
    HIPCHK(bp.alloc(&cd.src[g], caps[g]));
    HIPCHK(bp.alloc(&cd.dst[g], caps[g]));
    HIPCHK(bp.alloc(&cd.aidx[g], caps[g]));
    const size_t tiles = caps[g] / WAVE_EDGES + 2;
    HIPCHK(bp.alloc(&e->fsum[g], tiles * CN_STRIDE));
    HIPCHK(bp.alloc(&e->lsum[g], tiles * CN_STRIDE));
  }
  for (int g = 0; g < CONF_MAX_GROUPS; ++g)   // run_acc: one [nodes of the type][84] buffer per group
    HIPCHK(bp.alloc(&e->racc[g], nodes_of[g] * CN_STRIDE));
  HIPCHK(bp.alloc(&cd.total, CONF_MAX_GROUPS));
  cd.overflow = e->overflow_dev;
  HIPCHK(hipMemset(cd.total, 0, sizeof(int) * CONF_MAX_GROUPS));
  HIPCHK(bp.alloc(&cd.keep_res, nR));
  HIPCHK(bp.alloc(&cd.ll_vec, caps[G_LL] * 4)); HIPCHK(bp.alloc(&cd.ll_dist, caps[G_LL])); HIPCHK(bp.alloc(&cd.ll_bond4, caps[G_LL] * 4));
  HIPCHK(bp.alloc(&cd.lr_vec, caps[G_LR] * 4)); HIPCHK(bp.alloc(&cd.lr_dist, caps[G_LR]));
  HIPCHK(bp.alloc(&cd.la_vec, caps[G_LA] * 4)); HIPCHK(bp.alloc(&cd.la_dist, caps[G_LA]));
  HIPCHK(bp.alloc(&cd.lr_pair, nL * Nr));
  HIPCHK(bp.alloc(&cd.la_pair, nL * Na));
  HIPCHK(bp.alloc(&e->ll_attr, caps[G_LL] * CNS));
  HIPCHK(bp.alloc(&e->lr_attr, caps[G_LR] * CNS));
  HIPCHK(bp.alloc(&e->la_attr, caps[G_LA] * CNS));
  HIPCHK(bp.alloc(&e->X0, (nL + nR + nA) * CN_STRIDE));
  HIPCHK(bp.alloc(&e->X1, (nL + nR + nA) * CN_STRIDE));
  HIPCHK(bp.alloc(&e->atom_conf_scratch, nL));
  HIPCHK(hipDeviceSynchronize());
  e->complex_ready = true;
  return 0;
}

// One pose batch inside a (multi-)score call
namespace {
struct ConfPass {
  cbd_conf_engine* e;
  int B;
  ConfDyn cd;
  float *Xin, *Xout;
  float* racc[CONF_MAX_GROUPS];
  const float* attr_of[CONF_MAX_GROUPS];
  const float* vec_of[CONF_MAX_GROUPS];
  float* conf_out;
  float* atom_out;
};
}  // namespace

// CBD_CONF_TRACE=1: synchronise after every stage and name it on stderr (fault localisation)
static int conf_stage(const char* name, hipStream_t s) {
  static const bool trace = getenv("CBD_CONF_TRACE") != nullptr;
  if (!trace) return 0;
  fprintf(stderr, "[cbd_conf] %s ...", name);
  HIPCHK(hipStreamSynchronize(s));
  fprintf(stderr, " ok\n");
  return 0;
}

// crop + graphs + pose-dependent edge embeddings + initial node features of one pose batch
static int conf_prepare(ConfPass& P, const float* pos_dev, float crop_beyond, hipStream_t s) {
  cbd_conf_engine* e = P.e;
  const int B = P.B;
  const ConfStatic& cs = e->cs;
  P.cd = e->cd;
  P.cd.pos = pos_dev;
  ConfDyn& cd = P.cd;
  const int Nl = cs.Nl, Nr = cs.Nr, Na = cs.Na;
  const int nL = B * Nl, nR = B * Nr, nA = B * Na;
  e->last_stream = s;
  e->last_B = B;
  e->dbg.clear();
  const float crop2 = crop_beyond > 0 ? crop_beyond * crop_beyond : std::numeric_limits<float>::infinity();
  HIPCHK(conf_launch_keep(cs, cd, B, crop2, s));
  CHK(conf_stage("keep", s));
  const float r = e->cfg.lig_max_radius;
  const int n_nodes[CONF_MAX_GROUPS] = {nL, nL, nL, nR, nR, nR, nA, nA, nA};
  HIPCHK(conf_launch_graph_lig(false, cs, cd, B, r * r, e->cfg.lig_radius_cap, e->cfg.cross_cutoff, s));
  CHK(conf_stage("count lig", s));
  HIPCHK(conf_launch_scan(cd, G_LL, G_LA + 1, n_nodes, s));
  CHK(conf_stage("scan lig", s));
  HIPCHK(conf_launch_graph_lig(true, cs, cd, B, r * r, e->cfg.lig_radius_cap, e->cfg.cross_cutoff, s));
  CHK(conf_stage("fill lig", s));
  HIPCHK(conf_launch_graph_rec_atom(false, cs, cd, B, s));
  CHK(conf_stage("count rec/atom", s));
  HIPCHK(conf_launch_scan(cd, G_RR, G_AR + 1, n_nodes, s));
  CHK(conf_stage("scan rec/atom", s));
  HIPCHK(conf_launch_graph_rec_atom(true, cs, cd, B, s));
  CHK(conf_stage("fill rec/atom", s));
  // ---- pose-dependent edge embeddings
  HIPCHK(conf_launch_edge_mlp(e->m_ll, cd.ll_dist, cd.ll_bond4, cd.total + G_LL, e->cap[G_LL], e->ll_attr, s));
  HIPCHK(conf_launch_edge_mlp(e->m_lr, cd.lr_dist, nullptr, cd.total + G_LR, e->cap[G_LR], e->lr_attr, s));
  HIPCHK(conf_launch_edge_mlp(e->m_la, cd.la_dist, nullptr, cd.total + G_LA, e->cap[G_LA], e->la_attr, s));
  CHK(conf_stage("edge mlps", s));
  // ---- node features
  P.Xin = e->X0; P.Xout = e->X1;
  HIPCHK(conf_launch_node_init(e->lig_base, e->rec_base, e->atom_base, B, Nl, Nr, Na, P.Xin, s));
  const float* attr_of[CONF_MAX_GROUPS] = {e->ll_attr, e->lr_attr, e->la_attr, e->rr_attr, e->lr_attr, e->ar_attr, e->aa_attr, e->la_attr, e->ar_attr};
  const float* vec_of[CONF_MAX_GROUPS] = {cd.ll_vec, cd.lr_vec, cd.la_vec, e->rr_vec, cd.lr_vec, e->ar_vec, e->aa_vec, cd.la_vec, e->ar_vec};
  // run_acc is addressed by JOINT node index: pre-offset each buffer by its node type's base for this batch size
  const size_t type_base[CONF_MAX_GROUPS] = {0, 0, 0, (size_t)nL, (size_t)nL, (size_t)nL, (size_t)nL + nR, (size_t)nL + nR, (size_t)nL + nR};
  for (int g = 0; g < CONF_MAX_GROUPS; ++g) {
    P.attr_of[g] = attr_of[g]; P.vec_of[g] = vec_of[g];
    P.racc[g] = e->racc[g] - type_base[g] * CN_STRIDE;
  }
  return 0;
}

// The forward pass of up to CONF_MAX_BATCHES pose batches (one engine = one complex each): everything per batch except the fused conv
// kernel, which runs ONCE per layer for all of them.
static int conf_score_impl(int n, cbd_conf_engine* const* engines, const int32_t* Bs, const float* const* pos_dev, float crop_beyond,
                           float* const* confidence_dev, float* const* atom_confidence_dev, hipStream_t s) {
  if (n < 1 || n > CONF_MAX_BATCHES) return fail(CBD_ERR_ARG, "1 .. %d pose batches per call", CONF_MAX_BATCHES);
  ConfPass P[CONF_MAX_BATCHES];
  for (int k = 0; k < n; ++k) {
    cbd_conf_engine* e = engines[k];
    if (!e || !pos_dev[k] || !confidence_dev[k]) return fail(CBD_ERR_ARG, "null argument");
    if (!e->complex_ready) return fail(CBD_ERR_STATE, "cbd_conf_set_complex has not been called");
    if (Bs[k] < 1 || Bs[k] > e->cfg.max_batch) return fail(CBD_ERR_CAPACITY, "batch of %d poses exceeds the engine capacity %d", Bs[k], e->cfg.max_batch);
    if (e->cfg.device != engines[0]->cfg.device) return fail(CBD_ERR_ARG, "the engines of one call must live on one device");
    for (int q = 0; q < k; ++q)
      if (engines[q] == e) return fail(CBD_ERR_ARG, "an engine may appear once per call");
    P[k].e = e; P[k].B = Bs[k]; P[k].conf_out = confidence_dev[k];
    P[k].atom_out = atom_confidence_dev && atom_confidence_dev[k] ? atom_confidence_dev[k] : e->atom_conf_scratch;
  }
  cbd_conf_engine* e0 = engines[0];
  HIPCHK(hipSetDevice(e0->cfg.device));
  for (int k = 0; k < n; ++k) CHK(conf_prepare(P[k], pos_dev[k], crop_beyond, s));

  for (int l = 0; l < 5; ++l) {
    const CLayerDev& L0 = e0->conv[l];
    const FctpShape S = fctp_shape(L0.in_level, L0.out_level);
    CArgs a{};
    int grid = 0;
    for (int k = 0; k < n; ++k) {
      cbd_conf_engine* e = P[k].e;
      const CLayerDev& L = e->conv[l];
      const ConfDyn& cd = P[k].cd;
      // conv grid = tiles of the actual capacity for this B (blocks past the device-side counts exit immediately)
      const double frac = (double)P[k].B / e->cfg.max_batch;
      for (int g = 0; g < L.n_groups; ++g) {
        a.g[a.n_groups++] = CGroup{cd.src[g], cd.dst[g], cd.aidx[g], P[k].vec_of[g], P[k].attr_of[g], L.wstream[g], cd.total + g,
                                   e->fsum[g], e->lsum[g], P[k].racc[g], P[k].Xin};
        grid += (int)((size_t)std::ceil(e->cap[g] * frac) / WAVE_EDGES + 1);
      }
    }
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (e0->timing) {      // a merged launch is timed by the first engine
      if (e0->ev_used == e0->ev_pool.size()) {
        hipEvent_t x, y;
        HIPCHK(hipEventCreate(&x)); HIPCHK(hipEventCreate(&y));
        e0->ev_pool.emplace_back(x, y);
      }
      ev0 = e0->ev_pool[e0->ev_used].first; ev1 = e0->ev_pool[e0->ev_used].second;
      ++e0->ev_used;
      HIPCHK(hipEventRecord(ev0, s));
    }
    HIPCHK(launch_fctp_conv(L0.in_level, L0.out_level, a, grid, s));
    if (e0->timing) HIPCHK(hipEventRecord(ev1, s));
    CHK(conf_stage("conv", s));
    for (int k = 0; k < n; ++k) {
      cbd_conf_engine* e = P[k].e;
      const CLayerDev& L = e->conv[l];
      const ConfDyn& cd = P[k].cd;
      const ConfStatic& cs = e->cs;
      const int nL = P[k].B * cs.Nl, nR = P[k].B * cs.Nr, nA = P[k].B * cs.Na;
      const int n_types = l == 4 ? 1 : 3;
      const int type_nodes[3] = {nL, nR, nA}, type_off[3] = {0, nL, nL + nR};
      for (int t = 0; t < n_types; ++t) {
        CFinArgs fa{};
        fa.n_groups = 3;
        for (int q = 0; q < 3; ++q) {
          const int g = 3 * t + q;
          fa.g[q] = CFinGroup{cd.start[g], cd.cnt[g], e->fsum[g], e->lsum[g], P[k].racc[g]};
        }
        HIPCHK(launch_fctp_finalize(fa, P[k].Xin, P[k].Xout, L.bn_scale, L.bn_mean, L.bn_bias, type_nodes[t], S.in_dim, S.out_dim, type_off[t], s));
      }
      std::swap(P[k].Xin, P[k].Xout);
      if (e->keep_debug) {
        HIPCHK(hipStreamSynchronize(s));
        std::vector<float> h((size_t)nL * CN_STRIDE);
        HIPCHK(hipMemcpy(h.data(), P[k].Xin, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        e->dbg["lig_layer" + std::to_string(l + 1)] = std::move(h);
      }
    }
    CHK(conf_stage("finalize", s));
  }
  for (int k = 0; k < n; ++k) {
    cbd_conf_engine* e = P[k].e;
    HIPCHK(conf_launch_heads(e->atom_head, e->conf_head, P[k].Xin, P[k].B, e->cs.Nl, P[k].atom_out, P[k].conf_out, s));
    if (e->keep_debug) {
      HIPCHK(hipStreamSynchronize(s));
      std::vector<int> keep((size_t)P[k].B * e->cs.Nr);
      HIPCHK(hipMemcpy(keep.data(), P[k].cd.keep_res, keep.size() * sizeof(int), hipMemcpyDeviceToHost));
      e->dbg["keep_res"] = std::vector<float>(keep.begin(), keep.end());
    }
  }
  CHK(conf_stage("heads", s));
  return 0;
}

int cbd_conf_score(cbd_conf_engine* e, int32_t B, const float* pos_dev, float crop_beyond, float* confidence_dev,
                   float* atom_confidence_dev, void* stream) {
  if (!e || !pos_dev || !confidence_dev) return fail(CBD_ERR_ARG, "null argument");
  return conf_score_impl(1, &e, &B, &pos_dev, crop_beyond, &confidence_dev, &atom_confidence_dev, reinterpret_cast<hipStream_t>(stream));
}

int cbd_conf_score_multi(int32_t n, cbd_conf_engine* const* engines, const int32_t* B, const float* const* pos_dev, float crop_beyond,
                         float* const* confidence_dev, float* const* atom_confidence_dev, void* stream) {
  if (!engines || !B || !pos_dev || !confidence_dev) return fail(CBD_ERR_ARG, "null argument");
  return conf_score_impl(n, engines, B, pos_dev, crop_beyond, confidence_dev, atom_confidence_dev, reinterpret_cast<hipStream_t>(stream));
}

int cbd_conf_check(cbd_conf_engine* e) {
  if (!e || !e->complex_ready) return fail(CBD_ERR_STATE, "no confidence call to check");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->last_stream));
  int flag = 0;
  HIPCHK(hipMemcpy(&flag, e->overflow_dev, sizeof(int), hipMemcpyDeviceToHost));
  if (flag) HIPCHK(hipMemset(e->overflow_dev, 0, sizeof(int)));
  if (flag) return fail(CBD_ERR_CAPACITY, "a ligand atom has more than %d receptor atoms within lig_max_radius", e->cd.la_cap);
  return 0;
}

int cbd_conf_set_option(cbd_conf_engine* e, const char* name, int64_t value) {
  if (!e || !name) return fail(CBD_ERR_ARG, "null argument");
  if (!strcmp(name, "debug")) { e->keep_debug = value != 0; return 0; }
  return fail(CBD_ERR_ARG, "unknown option '%s'", name);
}

int64_t cbd_conf_debug_fetch(cbd_conf_engine* e, const char* name, float* out, int64_t capacity) {
  if (!e || !name) return fail(CBD_ERR_ARG, "null argument");
  auto it = e->dbg.find(name);
  if (it == e->dbg.end()) return fail(CBD_ERR_ARG, "no debug tensor '%s' (enable the 'debug' option before cbd_conf_score)", name);
  const int64_t n = (int64_t)it->second.size();
  if (out && capacity >= n) memcpy(out, it->second.data(), n * sizeof(float));
  return n;
}

int cbd_conf_last_edge_counts(cbd_conf_engine* e, int64_t counts[9]) {
  if (!e || !e->complex_ready) return fail(CBD_ERR_STATE, "no complex");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipStreamSynchronize(e->last_stream));
  int h[CONF_MAX_GROUPS];
  HIPCHK(hipMemcpy(h, e->cd.total, sizeof h, hipMemcpyDeviceToHost));
  for (int g = 0; g < CONF_MAX_GROUPS; ++g) counts[g] = h[g];
  return 0;
}

int cbd_conf_kernel_timing(cbd_conf_engine* e, int32_t enable, int32_t reset, double* avg_ms, int64_t* n_launches, double* total_ms) {
  if (!e) return fail(CBD_ERR_ARG, "null engine");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->ev_used > 0) {
    HIPCHK(hipDeviceSynchronize());
    for (size_t i = 0; i < e->ev_used; ++i) {
      float ms = 0.f;
      HIPCHK(hipEventElapsedTime(&ms, e->ev_pool[i].first, e->ev_pool[i].second));
      e->t_total_ms += ms;
      ++e->t_n;
    }
    e->ev_used = 0;
  }
  if (avg_ms) *avg_ms = e->t_n ? e->t_total_ms / (double)e->t_n : 0.0;
  if (n_launches) *n_launches = e->t_n;
  if (total_ms) *total_ms = e->t_total_ms;
  if (reset) { e->t_total_ms = 0; e->t_n = 0; }
  e->timing = enable != 0;
  return 0;
}

int64_t cbd_conf_stream_floats(int32_t in_level, int32_t out_level) {
  return (int64_t)fctp_stream_floats(fctp_shape(in_level, out_level).ntiles);
}

int cbd_conf_pack_stream(int32_t in_level, int32_t out_level, const float* w1, const float* b1, const float* w2, const float* b2, float* out) {
  if (!w1 || !b1 || !w2 || !b2 || !out) return fail(CBD_ERR_ARG, "null argument");
  if (in_level < 0 || in_level > 3 || out_level < 1 || out_level > 3) return fail(CBD_ERR_ARG, "bad irreps level");
  const std::vector<float> st = pack_fctp_stream(in_level, out_level, w1, b1, w2, b2);
  memcpy(out, st.data(), st.size() * sizeof(float));
  return 0;
}

}  // extern "C"
