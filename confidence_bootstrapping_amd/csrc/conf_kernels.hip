// Graph construction, featurisation and output heads of the all-atom confidence model for gfx950 (MI355X).
// HBM/latency-bound helpers around fctp_conv.hip; every kernel cites the reference lines it replaces.
#include "conf_kernels.h"
#include "device_util.h"

namespace cbd {

// ---------------------------------------------------------------------------------------------------------------
// crop_beyond (reference utils/utils.py:395-399): residue kept iff any ligand atom of the pose is closer than the cutoff
__global__ void conf_keep_kernel(ConfStatic cs, ConfDyn cd, int B, float crop2) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * cs.Nr) return;
  const int b = idx / cs.Nr, r = idx % cs.Nr;
  const float* P = cd.pos + (size_t)b * cs.Nl * 3;
  const float rx = cs.rec_pos[3 * r], ry = cs.rec_pos[3 * r + 1], rz = cs.rec_pos[3 * r + 2];
  int keep = 0;
  for (int a = 0; a < cs.Nl; ++a) keep |= dist2_nofma(P[3 * a], P[3 * a + 1], P[3 * a + 2], rx, ry, rz) < crop2 ? 1 : 0;
  cd.keep_res[idx] = keep;
}

// ---------------------------------------------------------------------------------------------------------------
// Edges aggregated by LIGAND atoms, one wave per (pose, atom): lig-lig (bonds + radius graph,
// all_atom_score_model.py:530-553), lig->residue (radius on cutoff-scaled coordinates, :590-597) and lig->atom
// (radius 5 A, :608-609), all restricted to the residues / atoms the crop kept.  COUNT pass, scan, FILL pass.
template <bool FILL>
__global__ __launch_bounds__(64) void conf_graph_lig_kernel(ConfStatic cs, ConfDyn cd, int B, float lig_r2, int lig_cap, float cross_cut) {
  const int lane = lane_id();
  const int node = blockIdx.x;
  const int nL = B * cs.Nl;
  const int b = node / cs.Nl, a = node % cs.Nl;
  const float* P = cd.pos + (size_t)b * cs.Nl * 3;
  const float px = P[3 * a], py = P[3 * a + 1], pz = P[3 * a + 2];
  const int* keep = cd.keep_res + (size_t)b * cs.Nr;
  // ---- lig-lig
  {
    const int nb0 = cs.bond_row[a], nb1 = cs.bond_row[a + 1];
    int base = 0;
    if (FILL) {
      base = cd.start[G_LL][node];
      for (int k = nb0 + lane; k < nb1; k += 64) {
        const int e = base + (k - nb0), d = cs.bond_dst[k];
        float ux, uy, uz, n;
        unit_vec(P[3 * d] - px, P[3 * d + 1] - py, P[3 * d + 2] - pz, ux, uy, uz, n);
        cd.src[G_LL][e] = node; cd.dst[G_LL][e] = b * cs.Nl + d; cd.aidx[G_LL][e] = e;
        reinterpret_cast<f32x4*>(cd.ll_vec)[e] = f32x4{ux, uy, uz, 0.f};
        cd.ll_dist[e] = n;
        reinterpret_cast<f32x4*>(cd.ll_bond4)[e] = reinterpret_cast<const f32x4*>(cs.bond_attr)[k];
      }
      base += nb1 - nb0;
    }
    // radius_graph rows are [neighbour; centre] and the layer aggregates into row 0: atom a receives an edge from every
    // centre y whose capped scan (first lig_cap+1 in-radius atoms in index order, self included) contains a
    int kept = 0;
    for (int c0 = 0; c0 < cs.Nl; c0 += 64) {
      const int d = c0 + lane;
      bool ok = false;
      if (d < cs.Nl && d != a) {
        const float yx = P[3 * d], yy = P[3 * d + 1], yz = P[3 * d + 2];
        if (dist2_nofma(px, py, pz, yx, yy, yz) < lig_r2) {
          int rank = 0;
          for (int x = 0; x < a; ++x) rank += dist2_nofma(P[3 * x], P[3 * x + 1], P[3 * x + 2], yx, yy, yz) < lig_r2 ? 1 : 0;
          ok = rank < lig_cap + 1;
        }
      }
      const unsigned long long mk = __ballot(ok);
      if (FILL && ok) {
        const int e = base + kept + popc_below(mk, lane);
        float ux, uy, uz, n;
        unit_vec(P[3 * d] - px, P[3 * d + 1] - py, P[3 * d + 2] - pz, ux, uy, uz, n);
        cd.src[G_LL][e] = node; cd.dst[G_LL][e] = b * cs.Nl + d; cd.aidx[G_LL][e] = e;
        reinterpret_cast<f32x4*>(cd.ll_vec)[e] = f32x4{ux, uy, uz, 0.f};
        cd.ll_dist[e] = n;
        reinterpret_cast<f32x4*>(cd.ll_bond4)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      kept += __popcll(mk);
    }
    if (!FILL && lane == 0) cd.cnt[G_LL][node] = (nb1 - nb0) + kept;
  }
  // ---- lig -> residue
  {
    const float lp[3] = {px, py, pz};
    int n_e = 0;
    const int base = FILL ? cd.start[G_LR][node] : 0;
    for (int c0 = 0; c0 < cs.Nr; c0 += 64) {
      const int r = c0 + lane;
      bool in = false;
      if (r < cs.Nr) in = keep[r] != 0 && cross_pair_in(lp, cs.rec_pos + 3 * r, cross_cut);
      const unsigned long long m = __ballot(in);
      if (FILL && r < cs.Nr) {
        int eid = -1;
        if (in) {
          eid = base + n_e + popc_below(m, lane);
          float ux, uy, uz, n;
          unit_vec(cs.rec_pos[3 * r] - px, cs.rec_pos[3 * r + 1] - py, cs.rec_pos[3 * r + 2] - pz, ux, uy, uz, n);
          cd.src[G_LR][eid] = node; cd.dst[G_LR][eid] = nL + b * cs.Nr + r; cd.aidx[G_LR][eid] = eid;
          reinterpret_cast<f32x4*>(cd.lr_vec)[eid] = f32x4{ux, uy, uz, 0.f};
          cd.lr_dist[eid] = n;
        }
        cd.lr_pair[(size_t)node * cs.Nr + r] = eid;
      }
      n_e += __popcll(m);
    }
    if (!FILL && lane == 0) cd.cnt[G_LR][node] = n_e;
  }
  // ---- lig -> receptor atom
  {
    int n_e = 0;
    const int base = FILL ? cd.start[G_LA][node] : 0;
    for (int c0 = 0; c0 < cs.Na; c0 += 64) {
      const int k = c0 + lane;
      bool in = false;
      if (k < cs.Na)
        in = keep[cs.atom_res[k]] != 0 &&
             dist2_nofma(cs.atom_pos[3 * k], cs.atom_pos[3 * k + 1], cs.atom_pos[3 * k + 2], px, py, pz) < lig_r2;
      const unsigned long long m = __ballot(in);
      if (FILL && k < cs.Na) {
        int eid = -1;
        const int slot = n_e + popc_below(m, lane);
        if (in && slot < cd.la_cap) {
          eid = base + slot;
          float ux, uy, uz, n;
          unit_vec(cs.atom_pos[3 * k] - px, cs.atom_pos[3 * k + 1] - py, cs.atom_pos[3 * k + 2] - pz, ux, uy, uz, n);
          cd.src[G_LA][eid] = node; cd.dst[G_LA][eid] = nL + B * cs.Nr + b * cs.Na + k; cd.aidx[G_LA][eid] = eid;
          reinterpret_cast<f32x4*>(cd.la_vec)[eid] = f32x4{ux, uy, uz, 0.f};
          cd.la_dist[eid] = n;
        }
        cd.la_pair[(size_t)node * cs.Na + k] = eid;
      }
      n_e += __popcll(m);
    }
    if (!FILL && lane == 0) {
      if (n_e > cd.la_cap) { *cd.overflow = 1; n_e = cd.la_cap; }
      cd.cnt[G_LA][node] = n_e;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Edges aggregated by RESIDUES (res-res, res->lig = flipped lr, res->atom = flipped ar) and by receptor ATOMS
// (atom-atom, atom->lig = flipped la, atom->res), one wave per (pose, node); the stored receptor graphs are filtered
// to the kept nodes like torch_geometric.utils.subgraph does (utils/utils.py:409-417).  Runs after the ligand FILL pass
// (reads the pair maps).
template <bool FILL>
__global__ __launch_bounds__(64) void conf_graph_rec_atom_kernel(ConfStatic cs, ConfDyn cd, int B) {
  const int lane = lane_id();
  const int nL = B * cs.Nl, nR = B * cs.Nr;
  if ((int)blockIdx.x < nR) {
    const int rn = blockIdx.x;
    const int b = rn / cs.Nr, r = rn % cs.Nr;
    const int* keep = cd.keep_res + (size_t)b * cs.Nr;
    const bool kept = keep[r] != 0;
    const int me = nL + rn;
    {  // res-res
      int n_e = 0;
      const int base = FILL ? cd.start[G_RR][rn] : 0;
      const int j0 = cs.rr_ptr[r], j1 = cs.rr_ptr[r + 1];
      for (int c0 = j0; c0 < j1; c0 += 64) {
        const int jj = c0 + lane;
        bool ok = false;
        int d = 0;
        if (jj < j1) { d = cs.rr_dst[jj]; ok = kept && keep[d] != 0; }
        const unsigned long long m = __ballot(ok);
        if (FILL && ok) {
          const int e = base + n_e + popc_below(m, lane);
          cd.src[G_RR][e] = me; cd.dst[G_RR][e] = nL + b * cs.Nr + d; cd.aidx[G_RR][e] = cs.rr_eid[jj];
        }
        n_e += __popcll(m);
      }
      if (!FILL && lane == 0) cd.cnt[G_RR][rn] = n_e;
    }
    {  // res -> lig
      int n_e = 0;
      const int base = FILL ? cd.start[G_RL][rn] : 0;
      for (int c0 = 0; c0 < cs.Nl; c0 += 64) {
        const int a = c0 + lane;
        int eid = -1;
        if (a < cs.Nl) eid = cd.lr_pair[(size_t)(b * cs.Nl + a) * cs.Nr + r];
        const unsigned long long m = __ballot(eid >= 0);
        if (FILL && eid >= 0) {
          const int e = base + n_e + popc_below(m, lane);
          cd.src[G_RL][e] = me; cd.dst[G_RL][e] = b * cs.Nl + a; cd.aidx[G_RL][e] = eid;
        }
        n_e += __popcll(m);
      }
      if (!FILL && lane == 0) cd.cnt[G_RL][rn] = n_e;
    }
    {  // res -> atom
      const int j0 = cs.ra_ptr[r], j1 = cs.ra_ptr[r + 1];
      const int n_e = kept ? j1 - j0 : 0;
      if (FILL) {
        const int base = cd.start[G_RA][rn];
        for (int jj = lane; jj < n_e; jj += 64) {
          const int k = cs.ra_atom[j0 + jj];
          cd.src[G_RA][base + jj] = me; cd.dst[G_RA][base + jj] = nL + nR + b * cs.Na + k; cd.aidx[G_RA][base + jj] = k;
        }
      } else if (lane == 0) cd.cnt[G_RA][rn] = n_e;
    }
  } else {
    const int an = blockIdx.x - nR;
    const int b = an / cs.Na, k = an % cs.Na;
    const int* keep = cd.keep_res + (size_t)b * cs.Nr;
    const int res = cs.atom_res[k];
    const bool kept = keep[res] != 0;
    const int me = nL + nR + an;
    {  // atom-atom
      int n_e = 0;
      const int base = FILL ? cd.start[G_AA][an] : 0;
      const int j0 = cs.aa_ptr[k], j1 = cs.aa_ptr[k + 1];
      for (int c0 = j0; c0 < j1; c0 += 64) {
        const int jj = c0 + lane;
        bool ok = false;
        int d = 0;
        if (jj < j1) { d = cs.aa_dst[jj]; ok = kept && keep[cs.atom_res[d]] != 0; }
        const unsigned long long m = __ballot(ok);
        if (FILL && ok) {
          const int e = base + n_e + popc_below(m, lane);
          cd.src[G_AA][e] = me; cd.dst[G_AA][e] = nL + nR + b * cs.Na + d; cd.aidx[G_AA][e] = cs.aa_eid[jj];
        }
        n_e += __popcll(m);
      }
      if (!FILL && lane == 0) cd.cnt[G_AA][an] = n_e;
    }
    {  // atom -> lig
      int n_e = 0;
      const int base = FILL ? cd.start[G_AL][an] : 0;
      if (kept) {
        for (int c0 = 0; c0 < cs.Nl; c0 += 64) {
          const int a = c0 + lane;
          int eid = -1;
          if (a < cs.Nl) eid = cd.la_pair[(size_t)(b * cs.Nl + a) * cs.Na + k];
          const unsigned long long m = __ballot(eid >= 0);
          if (FILL && eid >= 0) {
            const int e = base + n_e + popc_below(m, lane);
            cd.src[G_AL][e] = me; cd.dst[G_AL][e] = b * cs.Nl + a; cd.aidx[G_AL][e] = eid;
          }
          n_e += __popcll(m);
        }
      }
      if (!FILL && lane == 0) cd.cnt[G_AL][an] = n_e;
    }
    {  // atom -> its residue
      if (FILL) {
        if (kept && lane == 0) {
          const int e = cd.start[G_AR][an];
          cd.src[G_AR][e] = me; cd.dst[G_AR][e] = nL + b * cs.Nr + res; cd.aidx[G_AR][e] = k;
        }
      } else if (lane == 0) cd.cnt[G_AR][an] = kept ? 1 : 0;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// exclusive scan of the per-node counts of one edge group per block (blockIdx.x + g0 = group)
struct ScanArgs { int* cnt[CONF_MAX_GROUPS]; int* start[CONF_MAX_GROUPS]; int n[CONF_MAX_GROUPS]; int* total; int g0; };
__global__ __launch_bounds__(1024) void conf_scan_kernel(ScanArgs sa) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int g = sa.g0 + blockIdx.x;
  const int n = sa.n[g];
  const int* cnt = sa.cnt[g];
  int* start = sa.start[g];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int v = i < n ? cnt[i] : 0;
    int x = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(x, off, 64);
      if (lane >= off) x += y;
    }
    if (lane == 63) wsum[wid] = x;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += wsum[w];
    const int carry = carry_s;
    if (i < n) start[i] = carry + woff + x - v;
    __syncthreads();
    if (tid == 1023) carry_s = carry + woff + x;
    __syncthreads();
  }
  if (tid == 0) sa.total[g] = carry_s;
}

// ---------------------------------------------------------------------------------------------------------------
// Edge embedding MLPs (lig_edge / lr / la / rec_edge / atom_edge / ar_edge _embedding, all_atom_score_model.py:86-96)
// on GaussianSmearing(dist) (+ bond one-hot): one thread per edge.  The constant sigma-embedding inputs (t = 0) and the
// constant rec_sigma_emb addend are folded into b0 / b1 by the host.
struct EdgeMlpArgs { ConfEdgeMlp m; };
__global__ __launch_bounds__(256) void conf_edge_mlp_kernel(EdgeMlpArgs A, const float* __restrict__ dist, const float* __restrict__ bond4,
                                                            const int* __restrict__ count, int cap, float* __restrict__ attr) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = count ? *count : cap;
  if (e >= n) return;
  const float d = dist[e];
  float h[CNS];
#pragma unroll
  for (int o = 0; o < CNS; ++o) h[o] = A.m.b0[o];
  for (int k = 0; k < 32; ++k) {
    const float t = d - A.m.offset[k];
    const float gk = expf(A.m.coeff * (t * t));
#pragma unroll
    for (int o = 0; o < CNS; ++o) h[o] = fmaf(A.m.WgT[k * CNS + o], gk, h[o]);
  }
  if (A.m.WbT) {
    const f32x4 bb = reinterpret_cast<const f32x4*>(bond4)[e];
    const float bv[4] = {bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int o = 0; o < CNS; ++o) h[o] = fmaf(A.m.WbT[c * CNS + o], bv[c], h[o]);
  }
  float out[CNS];
#pragma unroll
  for (int o = 0; o < CNS; ++o) { h[o] = fmaxf(h[o], 0.f); out[o] = A.m.b1[o]; }
#pragma unroll
  for (int k = 0; k < CNS; ++k)
#pragma unroll
    for (int o = 0; o < CNS; ++o) out[o] = fmaf(A.m.W1T[k * CNS + o], h[k], out[o]);
  f32x4* dstp = reinterpret_cast<f32x4*>(attr + (size_t)e * CNS);
#pragma unroll
  for (int q = 0; q < CNS / 4; ++q) dstp[q] = f32x4{out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]};
}

// unit vector + length of the stored receptor edges (pose independent; build_rec/atom/cross_rec_conv_graph :556-584,623-633)
__global__ void conf_static_geom_kernel(const float* __restrict__ pos_src, const float* __restrict__ pos_dst, const int* __restrict__ src,
                                        const int* __restrict__ dst, int n, float* __restrict__ vec, float* __restrict__ dist) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int s = src[e], d = dst[e];
  float ux, uy, uz, nn;
  unit_vec(pos_dst[3 * d] - pos_src[3 * s], pos_dst[3 * d + 1] - pos_src[3 * s + 1], pos_dst[3 * d + 2] - pos_src[3 * s + 2], ux, uy, uz, nn);
  reinterpret_cast<f32x4*>(vec)[e] = f32x4{ux, uy, uz, 0.f};
  dist[e] = nn;
}

// AtomEncoder (models/score_model.py:18-41): sum of categorical embeddings, then (optionally) Linear over
// [embedding | extra features].  One 64-thread block per node; constant extra inputs are folded into `bias` by the host.
__global__ __launch_bounds__(64) void conf_node_embed_kernel(const float* __restrict__ x, int x_stride, int n_cat, const int* __restrict__ table_off,
                                                             const float* __restrict__ tables, const float* __restrict__ W, int in_extra,
                                                             const float* __restrict__ bias, int n, float* __restrict__ out) {
  __shared__ float emb[CNS];
  __shared__ float part[64];
  const int node = blockIdx.x, tid = threadIdx.x;
  const float* xr = x + (size_t)node * x_stride;
  if (tid < CNS) {
    float s = 0.f;
    for (int c = 0; c < n_cat; ++c) s += tables[(size_t)table_off[c] + (size_t)((int)xr[c]) * CNS + tid];
    emb[tid] = s;
  }
  __syncthreads();
  if (!W) {
    if (tid < CNS) out[(size_t)node * CNS + tid] = emb[tid] + (bias ? bias[tid] : 0.f);
    return;
  }
  const int K = CNS + in_extra;
  for (int o = 0; o < CNS; ++o) {
    const float* w = W + (size_t)o * K;
    float s = 0.f;
    for (int k = tid; k < K; k += 64) s = fmaf(w[k], k < CNS ? emb[k] : xr[n_cat + (k - CNS)], s);
    part[tid] = s;
    __syncthreads();
    if (tid == 0) {
      float t = bias[o];
      for (int i = 0; i < 64; ++i) t += part[i];
      out[(size_t)node * CNS + o] = t;
    }
    __syncthreads();
  }
}

// initial joint node features of a batch of B poses: embedded scalars in columns 0..23, zeros elsewhere
// (F.pad of the ligand rows, all_atom_score_model.py:355; receptor/atom rows are already 24 wide with no embedding layers)
__global__ void conf_node_init_kernel(const float* __restrict__ lig_base, const float* __restrict__ rec_base, const float* __restrict__ atom_base,
                                      int B, int Nl, int Nr, int Na, float* __restrict__ node) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * (Nl + Nr + Na) * CN_STRIDE;
  if (idx >= total) return;
  const int i = (int)(idx / CN_STRIDE), c = (int)(idx % CN_STRIDE);
  float v = 0.f;
  if (c < CNS) {
    const int nL = B * Nl, nR = B * Nr;
    if (i < nL) v = lig_base[(size_t)(i % Nl) * CNS + c];
    else if (i < nL + nR) v = rec_base[(size_t)((i - nL) % Nr) * CNS + c];
    else v = atom_base[(size_t)((i - nL - nR) % Na) * CNS + c];
  }
  node[idx] = v;
}

// atom_confidence_predictor on [0e | 0o] of every ligand atom, mean over the pose, confidence_predictor
// (all_atom_score_model.py:436-446).  One block per pose.
__device__ __forceinline__ void head_layer(const float* W, const float* s, const float* t, const float* x, int in_dim, float* y, int tid) {
  if (tid < CNS) {
    float a = 0.f;
    for (int k = 0; k < in_dim; ++k) a = fmaf(W[tid * in_dim + k], x[k], a);
    y[tid] = fmaxf(a * s[tid] + t[tid], 0.f);
  }
}
__global__ __launch_bounds__(64) void conf_heads_kernel(ConfHead ah, ConfHead ch, const float* __restrict__ node, int Nl,
                                                        float* __restrict__ atom_conf, float* __restrict__ conf) {
  __shared__ float x[2 * CNS], h0[CNS], h1[CNS], acc[CNS];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < CNS) acc[tid] = 0.f;
  for (int a = 0; a < Nl; ++a) {
    const float* row = node + (size_t)(b * Nl + a) * CN_STRIDE;
    if (tid < CNS) { x[tid] = row[tid]; x[CNS + tid] = row[CC_0O + tid]; }
    __syncthreads();
    head_layer(ah.W0, ah.s0, ah.t0, x, 2 * CNS, h0, tid);
    __syncthreads();
    head_layer(ah.W1, ah.s1, ah.t1, h0, CNS, h1, tid);
    __syncthreads();
    if (tid < 1 + CNS) {
      float o = ah.b2[tid];
      for (int k = 0; k < CNS; ++k) o = fmaf(ah.W2[tid * CNS + k], h1[k], o);
      if (tid == 0) atom_conf[b * Nl + a] = o;
      else acc[tid - 1] += o;
    }
    __syncthreads();
  }
  if (tid < CNS) x[tid] = acc[tid] / (float)Nl;
  __syncthreads();
  head_layer(ch.W0, ch.s0, ch.t0, x, CNS, h0, tid);
  __syncthreads();
  head_layer(ch.W1, ch.s1, ch.t1, h0, CNS, h1, tid);
  __syncthreads();
  if (tid == 0) {
    float o = ch.b2[0];
    for (int k = 0; k < CNS; ++k) o = fmaf(ch.W2[k], h1[k], o);
    conf[b] = o;
  }
}

// ------------------------------------------------------------------------------------------------ host launchers
hipError_t conf_launch_keep(const ConfStatic& cs, const ConfDyn& cd, int B, float crop2, hipStream_t s) {
  const int n = B * cs.Nr;
  hipLaunchKernelGGL(conf_keep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, cs, cd, B, crop2);
  return hipGetLastError();
}

hipError_t conf_launch_graph_lig(bool fill, const ConfStatic& cs, const ConfDyn& cd, int B, float lig_r2, int lig_cap, float cross_cut, hipStream_t s) {
  const int n = B * cs.Nl;
  if (fill) hipLaunchKernelGGL(conf_graph_lig_kernel<true>, dim3(n), dim3(64), 0, s, cs, cd, B, lig_r2, lig_cap, cross_cut);
  else hipLaunchKernelGGL(conf_graph_lig_kernel<false>, dim3(n), dim3(64), 0, s, cs, cd, B, lig_r2, lig_cap, cross_cut);
  return hipGetLastError();
}

hipError_t conf_launch_graph_rec_atom(bool fill, const ConfStatic& cs, const ConfDyn& cd, int B, hipStream_t s) {
  const int n = B * (cs.Nr + cs.Na);
  if (fill) hipLaunchKernelGGL(conf_graph_rec_atom_kernel<true>, dim3(n), dim3(64), 0, s, cs, cd, B);
  else hipLaunchKernelGGL(conf_graph_rec_atom_kernel<false>, dim3(n), dim3(64), 0, s, cs, cd, B);
  return hipGetLastError();
}

hipError_t conf_launch_scan(const ConfDyn& cd, int g0, int g1, const int* n_nodes, hipStream_t s) {
  ScanArgs sa;
  for (int g = 0; g < CONF_MAX_GROUPS; ++g) { sa.cnt[g] = cd.cnt[g]; sa.start[g] = cd.start[g]; sa.n[g] = n_nodes[g]; }
  sa.total = cd.total;
  sa.g0 = g0;
  hipLaunchKernelGGL(conf_scan_kernel, dim3(g1 - g0), dim3(1024), 0, s, sa);
  return hipGetLastError();
}

hipError_t conf_launch_edge_mlp(const ConfEdgeMlp& m, const float* dist, const float* bond4, const int* count, int cap, float* attr,
                                hipStream_t s) {
  if (cap <= 0) return hipSuccess;
  EdgeMlpArgs A{m};
  hipLaunchKernelGGL(conf_edge_mlp_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, A, dist, bond4, count, cap, attr);
  return hipGetLastError();
}

hipError_t conf_launch_static_geom(const float* pos_src, const float* pos_dst, const int* src, const int* dst, int n, float* vec,
                                   float* dist, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(conf_static_geom_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pos_src, pos_dst, src, dst, n, vec, dist);
  return hipGetLastError();
}

hipError_t conf_launch_node_embed(const float* x, int x_stride, int n_cat, const int* table_off, const float* tables, const float* W,
                                  int in_extra, const float* bias, int n, float* out, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(conf_node_embed_kernel, dim3(n), dim3(64), 0, s, x, x_stride, n_cat, table_off, tables, W, in_extra, bias, n, out);
  return hipGetLastError();
}

hipError_t conf_launch_node_init(const float* lig_base, const float* rec_base, const float* atom_base, int B, int Nl, int Nr, int Na,
                                 float* node, hipStream_t s) {
  const size_t total = (size_t)B * (Nl + Nr + Na) * CN_STRIDE;
  hipLaunchKernelGGL(conf_node_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, lig_base, rec_base, atom_base, B, Nl, Nr, Na, node);
  return hipGetLastError();
}

hipError_t conf_launch_heads(const ConfHead& atom_head, const ConfHead& conf_head, const float* node, int B, int Nl,
                             float* atom_conf, float* conf, hipStream_t s) {
  hipLaunchKernelGGL(conf_heads_kernel, dim3(B), dim3(64), 0, s, atom_head, conf_head, node, Nl, atom_conf, conf);
  return hipGetLastError();
}

}  // namespace cbd
