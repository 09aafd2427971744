// Graph construction, edge featurisation, score heads and pose update for gfx950 (MI355X).
// These are the HBM/latency-bound parts of one reverse-diffusion step; the matrix-core work lives in tp_conv.hip.
// Each kernel cites the reference lines it replaces.  64-wide wavefronts throughout (ballot = 64-bit).
#include "kernels.h"
#include "device_util.h"

namespace cbd {

// Which pose batch does this workgroup belong to?  (Multi::off is a prefix sum over the batches of this launch; everything is
// wave-uniform, so the descriptor fields are fetched with scalar loads.)
CBD_DEV const PoseBatch& locate(const Multi& m, int& local) {
  int k = 0;
#pragma unroll
  for (int i = 1; i < MAX_COSCHED; ++i) k = (i < m.n && (int)blockIdx.x >= m.off[i]) ? i : k;
  local = blockIdx.x - m.off[k];
  return *m.d[k];
}

// descriptor upload: by-value kernel argument -> device memory (stream ordered, capturable, no host buffer to keep alive)
__global__ __launch_bounds__(64) void set_desc_kernel(PoseBatch v, PoseBatch* dst) {
  const int* src = reinterpret_cast<const int*>(&v);
  int* d = reinterpret_cast<int*>(dst);
  for (int i = threadIdx.x; i < (int)(sizeof(PoseBatch) / sizeof(int)); i += 64) d[i] = src[i];
}
hipError_t launch_set_desc(const PoseBatch& v, PoseBatch* dst, hipStream_t s) {
  static_assert(sizeof(PoseBatch) % sizeof(int) == 0 && sizeof(PoseBatch) <= 3584, "PoseBatch is passed by value");
  hipLaunchKernelGGL(set_desc_kernel, dim3(1), dim3(64), 0, s, v, dst);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Graph construction.  One wave per node.  Blocks [0, B*Nl) handle ligand nodes (ligand radius graph,
// score_model.py:502-507, and ligand->receptor cross edges, :564-573); blocks [B*Nl, B*Nl + B*Nr) handle
// receptor nodes (the flipped cross edges, :356-357).  COUNT pass, single-block scan, FILL pass.
template <bool FILL>
__global__ __launch_bounds__(64) void graph_kernel(Multi mm, float lig_r2, int lig_cap, float cutoff) {
  int node;
  const PoseBatch& PB = locate(mm, node);
  const GraphStatic gs = PB.gs;   // by value: kept in scalar registers, not re-read after every store
  const GraphDyn gd = PB.gd;
  const int B = PB.B;
  const int lane = lane_id();
  const int nL = B * gs.Nl;
  const int roff = gs.rec_off;
  if (node < nL) {
    const int b = node / gs.Nl, a = node % gs.Nl;
    const float* P = gd.pos + (size_t)b * gs.Nl * 3;
    const float px = P[3 * a], py = P[3 * a + 1], pz = P[3 * a + 2];
    // ---- ligand-ligand: bonds first, then radius-graph edges
    const int nb0 = gs.bond_row[a], nb1 = gs.bond_row[a + 1];
    int base = 0;
    if (FILL) {
      base = gd.start_ll[node];
      for (int k = nb0 + lane; k < nb1; k += 64) {
        const int e = base + (k - nb0), d = gs.bond_dst[k];
        float ux, uy, uz, n;
        unit_vec(P[3 * d] - px, P[3 * d + 1] - py, P[3 * d + 2] - pz, ux, uy, uz, n);
        gd.ll_src[e] = node; gd.ll_dst[e] = b * gs.Nl + d; gd.ll_aidx[e] = e;
        reinterpret_cast<f32x4*>(gd.ll_vec)[e] = f32x4{ux, uy, uz, 0.f};
        gd.ll_dist[e] = n;
        reinterpret_cast<f32x4*>(gd.ll_bond4)[e] = reinterpret_cast<const f32x4*>(gs.bond_attr)[k];
      }
      base += nb1 - nb0;
    }
    // radius_graph(pos, r, batch) (score_model.py:502) returns [neighbour; centre] rows and the layer aggregates into row 0:
    // node a receives an edge from every centre y that has a among the first lig_cap+1 in-radius atoms of ITS scan (index
    // order, self included, torch_cluster radius with max_num_neighbors+1).  rank_y(a) = #{x < a : |x - y| < r}.
    int kept = 0;
    for (int c0 = 0; c0 < gs.Nl; c0 += 64) {
      const int d = c0 + lane;   // candidate centre y
      bool keep = false;
      if (d < gs.Nl && d != a) {
        const float yx = P[3 * d], yy = P[3 * d + 1], yz = P[3 * d + 2];
        if (dist2_nofma(px, py, pz, yx, yy, yz) < lig_r2) {
          int rank = 0;
          for (int x = 0; x < a; ++x) rank += dist2_nofma(P[3 * x], P[3 * x + 1], P[3 * x + 2], yx, yy, yz) < lig_r2 ? 1 : 0;
          keep = rank < lig_cap + 1;
        }
      }
      const unsigned long long mk = __ballot(keep);
      if (FILL && keep) {
        const int e = base + kept + popc_below(mk, lane);
        float ux, uy, uz, n;
        unit_vec(P[3 * d] - px, P[3 * d + 1] - py, P[3 * d + 2] - pz, ux, uy, uz, n);
        gd.ll_src[e] = node; gd.ll_dst[e] = b * gs.Nl + d; gd.ll_aidx[e] = e;
        reinterpret_cast<f32x4*>(gd.ll_vec)[e] = f32x4{ux, uy, uz, 0.f};
        gd.ll_dist[e] = n;
        reinterpret_cast<f32x4*>(gd.ll_bond4)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      kept += __popcll(mk);
    }
    if (!FILL && lane == 0) gd.cnt_ll[node] = (nb1 - nb0) + kept;
    // ---- ligand -> receptor cross edges (cap 10000 never binds for Nr <= 10000)
    const float lp[3] = {px, py, pz};
    int nlr = 0;
    const int lbase = FILL ? gd.start_lr[node] : 0;
    for (int c0 = 0; c0 < gs.Nr; c0 += 64) {
      const int r = c0 + lane;
      bool in = false;
      if (r < gs.Nr) in = cross_pair_in(lp, gs.rec_pos + 3 * r, cutoff);
      const unsigned long long m = __ballot(in);
      if (FILL && r < gs.Nr) {
        int eid = -1;
        if (in) {
          eid = lbase + nlr + popc_below(m, lane);
          float ux, uy, uz, n;
          unit_vec(gs.rec_pos[3 * r] - px, gs.rec_pos[3 * r + 1] - py, gs.rec_pos[3 * r + 2] - pz, ux, uy, uz, n);
          gd.lr_src[eid] = node; gd.lr_dst[eid] = roff + b * gs.Nr + r; gd.lr_aidx[eid] = eid;
          reinterpret_cast<f32x4*>(gd.lr_vec)[eid] = f32x4{ux, uy, uz, 0.f};
          gd.lr_dist[eid] = n;
        }
        gd.pair_eid[(size_t)node * gs.Nr + r] = eid;
      }
      nlr += __popcll(m);
    }
    if (!FILL && lane == 0) gd.cnt_lr[node] = nlr;
  } else {
    // ---- receptor node: flipped cross edges (aggregating node = residue, features read from ligand atoms)
    const int rn = node - nL;
    const int b = rn / gs.Nr, r = rn % gs.Nr;
    const float* P = gd.pos + (size_t)b * gs.Nl * 3;
    const float* rp = gs.rec_pos + 3 * r;
    int nrl = 0;
    const int rbase = FILL ? gd.start_rl[rn] : 0;
    for (int c0 = 0; c0 < gs.Nl; c0 += 64) {
      const int a = c0 + lane;
      bool in = false;
      if (a < gs.Nl) in = cross_pair_in(P + 3 * a, rp, cutoff);
      const unsigned long long m = __ballot(in);
      if (FILL && in) {
        const int e = rbase + nrl + popc_below(m, lane);
        const int lr = gd.pair_eid[(size_t)(b * gs.Nl + a) * gs.Nr + r];
        const f32x4 vv = reinterpret_cast<const f32x4*>(gd.lr_vec)[lr];
        gd.rl_src[e] = roff + rn; gd.rl_dst[e] = b * gs.Nl + a; gd.rl_aidx[e] = lr;
        reinterpret_cast<f32x4*>(gd.rl_vec)[e] = f32x4{-vv.x, -vv.y, -vv.z, 0.f};   // sh(-edge_vec), score_model.py:582
      }
      nrl += __popcll(m);
    }
    if (!FILL && lane == 0) gd.cnt_rl[rn] = nrl;
  }
}

// m: off = prefix sums of B * (Nl + Nr) workgroups
hipError_t launch_graph_count(const Multi& m, float lig_r, int lig_cap, float cutoff, hipStream_t s) {
  if (m.off[m.n] <= 0) return hipSuccess;
  hipLaunchKernelGGL((graph_kernel<false>), dim3(m.off[m.n]), dim3(64), 0, s, m, lig_r * lig_r, lig_cap, cutoff);
  return hipGetLastError();
}
__global__ __launch_bounds__(64) void graph_fill_rec(Multi mm, float cutoff) {
  int rn;
  const PoseBatch& PB = locate(mm, rn);
  const GraphStatic gs = PB.gs;   // by value: kept in scalar registers, not re-read after every store
  const GraphDyn gd = PB.gd;
  const int lane = lane_id();
  const int roff = gs.rec_off;
  const int b = rn / gs.Nr, r = rn % gs.Nr;
  const float* P = gd.pos + (size_t)b * gs.Nl * 3;
  const float* rp = gs.rec_pos + 3 * r;
  int nrl = 0;
  const int rbase = gd.start_rl[rn];
  for (int c0 = 0; c0 < gs.Nl; c0 += 64) {
    const int a = c0 + lane;
    bool in = false;
    if (a < gs.Nl) in = cross_pair_in(P + 3 * a, rp, cutoff);
    const unsigned long long m = __ballot(in);
    if (in) {
      const int e = rbase + nrl + popc_below(m, lane);
      const int lr = gd.pair_eid[(size_t)(b * gs.Nl + a) * gs.Nr + r];
      const f32x4 vv = reinterpret_cast<const f32x4*>(gd.lr_vec)[lr];
      gd.rl_src[e] = roff + rn; gd.rl_dst[e] = b * gs.Nl + a; gd.rl_aidx[e] = lr;
      reinterpret_cast<f32x4*>(gd.rl_vec)[e] = f32x4{-vv.x, -vv.y, -vv.z, 0.f};   // sh(-edge_vec), score_model.py:582
    }
    nrl += __popcll(m);
  }
}

// The fill pass runs as two launches: receptor-node blocks read pair_eid / lr_vec written by the ligand-node blocks.
// m_lig: off = prefix sums of B * Nl workgroups (ligand nodes only); m_rec: of B * Nr
hipError_t launch_graph_fill(const Multi& m_lig, const Multi& m_rec, float lig_r, int lig_cap, float cutoff, hipStream_t s) {
  if (m_lig.off[m_lig.n] <= 0) return hipSuccess;
  hipLaunchKernelGGL((graph_kernel<true>), dim3(m_lig.off[m_lig.n]), dim3(64), 0, s, m_lig, lig_r * lig_r, lig_cap, cutoff);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(graph_fill_rec, dim3(m_rec.off[m_rec.n]), dim3(64), 0, s, m_rec, cutoff);
  return hipGetLastError();
}

// Exclusive scans of the three per-node count arrays, one 1024-thread workgroup per array (wave-shuffle scan over
// coalesced 1024-entry chunks), edge totals -> counts[], algorithmic work accounting -> stats[] (bench.py).
__global__ __launch_bounds__(1024) void graph_scan_kernel(Multi mm) {
  __shared__ int wsum[16];
  __shared__ int carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  int which;
  const PoseBatch& PB = locate(mm, which);
  const GraphStatic gs = PB.gs;   // by value: kept in scalar registers, not re-read after every store
  const GraphDyn gd = PB.gd;
  const int B = PB.B;
  unsigned long long* const stats = PB.stats;
  const int nL = B * gs.Nl, nR = B * gs.Nr;
  const int* cnt = which == 0 ? gd.cnt_ll : which == 1 ? gd.cnt_lr : gd.cnt_rl;
  int* start = which == 0 ? gd.start_ll : which == 1 ? gd.start_lr : gd.start_rl;
  const int n = which == 2 ? nR : nL;
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
  if (tid == 0) {
    const unsigned long long total = (unsigned long long)carry_s;
    gd.counts[which == 0 ? 0 : which == 1 ? 1 : 3] = (int)total;
    if (which == 2) gd.counts[2] = B * gs.Err;
    if (which == 0) gd.counts[4] = 0;   // torsion-edge counter of this forward pass (bond_nb_kernel adds to it later in the stream)
    if (stats) {   // edge-layer visits: 3 ligand embedding layers visit ll; 4 joint layers visit all groups, the last one ll + lr
      if (which == 0) { atomicAdd(&stats[0], total); atomicAdd(&stats[1], 5ull * total); atomicAdd(&stats[2], 1ull); }
      if (which == 1) atomicAdd(&stats[1], 5ull * total);
      if (which == 2) {
        atomicAdd(&stats[1], 4ull * total + 4ull * (unsigned long long)(B * gs.Err));
        // credited but not executed: the layer-0 receptor->receptor messages are computed once per complex, not once per sample
        atomicAdd(&stats[3], (unsigned long long)((B - 1) * gs.Err));
      }
    }
  }
}

// m: 3 workgroups per batch
hipError_t launch_graph_scan(const Multi& m, hipStream_t s) {
  hipLaunchKernelGGL(graph_scan_kernel, dim3(m.off[m.n]), dim3(1024), 0, s, m);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Edge embedding MLP (GaussianSmearing + Linear/ReLU/Linear):
// lig_edge_embedding / cross_edge_embedding / rec_edge_embedding / final_edge_embedding,
// score_model.py:111,114,123,259-264 applied at :286,311,352,660 on the features built at :504-518,534-535,578-580,658.
// One lane per edge: all weight addresses are wave-uniform (scalar loads), the 2 x 32x32 products are straight-line
// FMAs on register arrays, no cross-lane traffic; each lane writes its own 128-B row.
__global__ __launch_bounds__(256) void edge_mlp_kernel(EdgeMlpArgs a) {
  // segment of this workgroup (every segment is padded to whole 256-edge workgroups)
  int sg = 0, blk = blockIdx.x;
  for (int i = 0; i < a.n; ++i) {
    const int nb = (a.seg[i].cap + 255) / 256;
    if (blk >= nb && i + 1 < a.n) { blk -= nb; sg = i + 1; } else break;
  }
  const EdgeSeg& S = a.seg[sg];
  const EdgeMlp& m = S.m;
  const float* __restrict__ dist = S.dist;
  const float* __restrict__ bond4 = S.bond4;
  float* __restrict__ out = S.out;
  const int n = S.count ? min(*S.count, S.cap) : S.cap;
  const int e = blk * 256 + threadIdx.x;
  if (e >= n) return;
  const float d = dist[e];
  float h[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) h[o] = m.part[o];
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    const float t = d - m.offset[k];
    const float gk = expf(m.coeff * (t * t));
#pragma unroll
    for (int o = 0; o < 32; ++o) h[o] = fmaf(m.WgT[k * 32 + o], gk, h[o]);
  }
  if (m.WbT && bond4) {
    const f32x4 b = reinterpret_cast<const f32x4*>(bond4)[e];
    const float bb[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int o = 0; o < 32; ++o) h[o] = fmaf(m.WbT[c * 32 + o], bb[c], h[o]);
  }
  float r[32];
#pragma unroll
  for (int o = 0; o < 32; ++o) { h[o] = fmaxf(h[o], 0.f); r[o] = m.b1[o]; }
#pragma unroll
  for (int k = 0; k < 32; ++k)
#pragma unroll
    for (int o = 0; o < 32; ++o) r[o] = fmaf(m.W1T[k * 32 + o], h[k], r[o]);
  f32x4* po = reinterpret_cast<f32x4*>(out + (size_t)e * 32);
#pragma unroll
  for (int q = 0; q < 8; ++q) po[q] = f32x4{r[4 * q], r[4 * q + 1], r[4 * q + 2], r[4 * q + 3]};
}

hipError_t launch_edge_mlp(const EdgeMlpArgs& a, hipStream_t s) {
  int grid = 0;
  for (int i = 0; i < a.n; ++i) grid += (a.seg[i].cap + 255) / 256;
  if (grid <= 0) return hipSuccess;
  hipLaunchKernelGGL(edge_mlp_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Per-step constant vectors: everything that depends on the diffusion time only (the reference recomputes the
// sinusoidal embedding per node/edge, score_model.py:323-326,499,510,579,645; here its linear images are
// computed once per step).
CBD_DEV float dot32(const float* w, const float* x) {
  float s = 0.f;
#pragma unroll 8
  for (int k = 0; k < 32; ++k) s = fmaf(w[k], x[k], s);
  return s;
}

// `se2` = [sigma_emb | sigma_emb_t] of the step (cbd_step): the receptor side embeds the translation time, everything on the ligand
// side and the two magnitude heads the common time t -- the same vector unless the model has an asyncronous noise schedule
__global__ __launch_bounds__(64) void step_prep_kernel(StepWeights w, StepVectors v, const float* __restrict__ se2) {
  __shared__ float hid[32];
  const float* const se_rec = se2;
  const float* const se = se2 + 32;
  const int o = threadIdx.x;
  if (o < 32) hid[o] = fmaxf(dot32(w.rec_sig_w0 + o * 32, se_rec) + w.rec_sig_b0[o], 0.f);
  __syncthreads();
  if (o < 32) {
    v.rec_sigma_emb[o] = dot32(w.rec_sig_w1 + o * 32, hid) + w.rec_sig_b1[o];
    v.ll_part[o] = dot32(w.lig_edge_w0 + o * 68 + 4, se) + w.lig_edge_b0[o];          // input = [bond4 | sigma_emb | gauss]
    v.lr_part[o] = dot32(w.cross_w0 + o * 64, se) + w.cross_b0[o];                    // input = [sigma_emb | gauss]
    v.center_part[o] = dot32(w.center_w0 + o * 64 + 32, se) + w.center_b0[o];         // input = [gauss | sigma_emb]
    v.lig_node_c[o] = dot32(w.lig_node_w + o * 64 + 32, se) + w.lig_node_b[o];        // input = [emb sum | sigma_emb]
    v.tr_part[o] = dot32(w.tr_w0 + o * 33 + 1, se) + w.tr_b0[o];                      // input = [norm | sigma_emb]
    v.rot_part[o] = dot32(w.rot_w0 + o * 33 + 1, se) + w.rot_b0[o];
  }
}

hipError_t launch_step_prep(const StepWeights& w, const StepVectors& v, const float* sigma_emb_dev, hipStream_t s) {
  hipLaunchKernelGGL(step_prep_kernel, dim3(1), dim3(64), 0, s, w, v, sigma_emb_dev);
  return hipGetLastError();
}

// X[xi][b*Nl + a][c] = lig_static32[a][c] + lig_node_c[c] (c < 32), 0 elsewhere        (AtomEncoder, score_model.py:285)
__global__ void lig_node_init_kernel(Multi mm, const float* __restrict__ c32, int xi) {
  int blk;
  const PoseBatch& PB = locate(mm, blk);
  const int Nl = PB.gs.Nl;
  const int idx = blk * blockDim.x + threadIdx.x;
  const int n = idx / NODE_STRIDE, c = idx % NODE_STRIDE;
  if (n >= PB.B * Nl) return;
  PB.X[xi][(size_t)n * NODE_STRIDE + c] = c < 32 ? PB.lig_static32[(n % Nl) * 32 + c] + c32[c] : 0.f;
}
hipError_t launch_lig_node_init(const Multi& m, const float* c32, int xi, hipStream_t s) {
  if (m.off[m.n] <= 0) return hipSuccess;
  hipLaunchKernelGGL(lig_node_init_kernel, dim3(m.off[m.n]), dim3(256), 0, s, m, c32, xi);
  return hipGetLastError();
}

// Everything of a batch that depends on the diffusion time only, one launch: receptor rows
//   X[xi][rec_off + b*Nr + r] = rec_static[r] (+ rec_sigma_emb on the 32 scalars)          (score_model.py:324-325)
// followed by the receptor edge attributes rr_attr_t[e][c] = rr_attr0[e][c] + rec_sigma_emb[c]   (score_model.py:534-535).
// Workgroups of a batch: ceil(B*Nr*NODE_STRIDE / 256) for the rows, then ceil(Err*32 / 256) for the attributes.
__global__ void rec_time_init_kernel(Multi mm, const float* __restrict__ se, int xi) {
  int blk;
  const PoseBatch& PB = locate(mm, blk);
  const int Nr = PB.gs.Nr;
  const int row_blocks = (PB.B * Nr * NODE_STRIDE + 255) / 256;
  if (blk < row_blocks) {
    const int idx = blk * 256 + threadIdx.x;
    const int n = idx / NODE_STRIDE, c = idx % NODE_STRIDE;
    if (n >= PB.B * Nr) return;
    const float v = PB.rec_static[(size_t)(n % Nr) * NODE_STRIDE + c];
    PB.X[xi][(size_t)(PB.gs.rec_off + n) * NODE_STRIDE + c] = c < 32 ? v + se[c] : v;
  } else {
    const int idx = (blk - row_blocks) * 256 + threadIdx.x;
    if (idx < PB.gs.Err * 32) PB.rr_attr_t[idx] = PB.rr_attr0[idx] + se[idx & 31];
  }
}
hipError_t launch_rec_time_init(const Multi& m, const float* se, int xi, hipStream_t s) {
  if (m.off[m.n] <= 0) return hipSuccess;
  hipLaunchKernelGGL(rec_time_init_kernel, dim3(m.off[m.n]), dim3(256), 0, s, m, se, xi);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Centre convolution -> translation / rotation scores (score_model.py:393-420, 635-648).
// final_conv.tp is e3nn FullyConnectedTensorProduct(74 x (0e+1o) -> 2x1o + 2x1e); its six instructions reduce to the
// closed forms below (prototype + check against the Wigner-3j einsum: tests/test_kernel_math.py::final_conv_tp).
// Stage 1: one workgroup per (sample, atom) edge -> 12-value message.  Stage 2: one wave per sample: mean over atoms in
// index order (deterministic), BatchNorm, magnitude MLPs.
__global__ __launch_bounds__(128) void center_msg_kernel(CenterHead h, StepVectors sv, Multi mm, int xi) {
  __shared__ float s_in[64], s_hid[64], s_w[124], s_c[3], s_g[32], s_eh[32];
  int edge;   // (sample, atom) edge of this workgroup within its batch
  const PoseBatch& PB = locate(mm, edge);
  const int Nl = PB.gs.Nl;
  const float* __restrict__ pos = PB.gd.pos;
  const float* __restrict__ node = PB.X[xi];
  float* __restrict__ msg = PB.center_msg;
  const int b = edge / Nl, a = edge % Nl, tid = threadIdx.x;
  const float* P = pos + (size_t)b * Nl * 3;
  if (tid < 3) {
    float c = 0.f;
    for (int k = 0; k < Nl; ++k) c += P[3 * k + tid];
    s_c[tid] = c / (float)Nl;
  }
  __syncthreads();
  const float pw_o = sqrtf(3.f / 44.f), pw_e = sqrtf(3.f / 18.f);
  const float is3 = 0.57735026918962576f, is6 = 0.40824829046386302f, s3 = 1.7320508075688772f;
  const float* x = node + (size_t)(b * Nl + a) * NODE_STRIDE;
  float ux, uy, uz, d;
  unit_vec(P[3 * a] - s_c[0], P[3 * a + 1] - s_c[1], P[3 * a + 2] - s_c[2], ux, uy, uz, d);
  // centre edge embedding: [gauss(d) | sigma_emb] -> 32 -> 32
  if (tid < 32) { const float t = d - h.offset[tid]; s_g[tid] = expf(h.coeff * (t * t)); }
  __syncthreads();
  if (tid < 32) {
    float v = sv.center_part[tid];
    for (int k = 0; k < 32; ++k) v = fmaf(h.ce_WgT[k * 32 + tid], s_g[k], v);
    s_eh[tid] = fmaxf(v, 0.f);
  }
  __syncthreads();
  if (tid < 32) {
    float v = h.ce_b1[tid];
    for (int k = 0; k < 32; ++k) v = fmaf(h.ce_W1T[k * 32 + tid], s_eh[k], v);
    s_in[tid] = v;
  } else if (tid < 64) {
    s_in[tid] = x[tid - 32];   // lig_node_attr[atom, :ns]  (fixed_center_conv, score_model.py:397)
  }
  __syncthreads();
  if (tid < 64) {
    float v = h.fc_b0[tid];
    for (int k = 0; k < 64; ++k) v = fmaf(h.fc_w0[tid * 64 + k], s_in[k], v);
    s_hid[tid] = fmaxf(v, 0.f);
  }
  __syncthreads();
  if (tid < 124) {
    float v = h.fc_b1[tid];
    for (int k = 0; k < 64; ++k) v = fmaf(h.fc_w1[tid * 64 + k], s_hid[k], v);
    s_w[tid] = v;
  }
  __syncthreads();
  if (tid < 12) {
    const int blk = tid / 6, wv = (tid % 6) / 3, k = tid % 3;   // blk 0: 2x1o, 1: 2x1e ; wv = multiplicity ; k = component
    const float sh[3] = {s3 * ux, s3 * uy, s3 * uz};
    float r = 0.f;
    if (blk == 0) {
      float t0 = 0.f;
      for (int u = 0; u < 32; ++u) t0 = fmaf(s_w[u * 2 + wv], x[u], t0);
      r += pw_o * is3 * t0 * sh[k];
      for (int u = 0; u < 6; ++u) {
        r += pw_o * is3 * s_w[64 + u * 2 + wv] * x[COL_1O + 3 * u + k];
        const float* e = x + COL_1E + 3 * u;
        const float cr = k == 0 ? e[1] * sh[2] - e[2] * sh[1] : k == 1 ? e[2] * sh[0] - e[0] * sh[2] : e[0] * sh[1] - e[1] * sh[0];
        r += pw_o * is6 * s_w[100 + u * 2 + wv] * cr;
      }
    } else {
      for (int u = 0; u < 6; ++u) {
        const float* o = x + COL_1O + 3 * u;
        const float cr = k == 0 ? o[1] * sh[2] - o[2] * sh[1] : k == 1 ? o[2] * sh[0] - o[0] * sh[2] : o[0] * sh[1] - o[1] * sh[0];
        r += pw_e * is6 * s_w[76 + u * 2 + wv] * cr;
        r += pw_e * is3 * s_w[88 + u * 2 + wv] * x[COL_1E + 3 * u + k];
        r += pw_e * is3 * s_w[112 + u * 2 + wv] * x[COL_0O + u] * sh[k];
      }
    }
    msg[(size_t)edge * 12 + tid] = r;
  }
}

__global__ __launch_bounds__(64) void center_final_kernel(CenterHead h, StepVectors sv, Multi mm, float tr_sigma, float rot_norm) {
  __shared__ float s_acc[12];
  int b;
  const PoseBatch& PB = locate(mm, b);
  const int Nl = PB.gs.Nl;
  const float* __restrict__ msg = PB.center_msg;
  float* __restrict__ tr_out = PB.tr_out;
  float* __restrict__ rot_out = PB.rot_out;
  float* __restrict__ dbg = PB.dbg_global;
  const int tid = threadIdx.x;
  if (tid < 12) {
    float a = 0.f;
    for (int k = 0; k < Nl; ++k) a += msg[((size_t)b * Nl + k) * 12 + tid];
    const float mean = a / (float)Nl;
    if (dbg) dbg[b * 12 + tid] = mean;
    s_acc[tid] = mean * h.bn_scale[tid / 3];   // e3nn BatchNorm on 2x1o+2x1e: scale per multiplicity channel, no shift
  }
  __syncthreads();
  // tr = g[0:3] + g[6:9], rot = g[3:6] + g[9:12]; magnitude re-scaling MLPs (score_model.py:402-420)
  const int which = tid >> 5, o = tid & 31;   // 0: tr, 1: rot
  const float vx = s_acc[3 * which] + s_acc[6 + 3 * which], vy = s_acc[3 * which + 1] + s_acc[7 + 3 * which],
              vz = s_acc[3 * which + 2] + s_acc[8 + 3 * which];
  const float nrm = sqrtf(vx * vx + vy * vy + vz * vz);
  const float* w0n = which ? h.rot_w0n : h.tr_w0n;
  const float* w1 = which ? h.rot_w1 : h.tr_w1;
  const float* part = which ? sv.rot_part : sv.tr_part;
  float hv = fmaxf(fmaf(w0n[o], nrm, part[o]), 0.f) * w1[o];
  for (int off = 16; off > 0; off >>= 1) hv += __shfl_xor(hv, off);
  const float mag = hv + (which ? h.rot_b1[0] : h.tr_b1[0]);
  if (o < 3) {
    const float comp = o == 0 ? vx : o == 1 ? vy : vz;
    const float val = comp / nrm * mag;
    if (which == 0) tr_out[b * 3 + o] = val / tr_sigma;
    else rot_out[b * 3 + o] = val * rot_norm;
  }
}

// m_atoms: B * Nl workgroups per batch; m_samples: B per batch
hipError_t launch_center_head(const CenterHead& h, const StepVectors& v, const Multi& m_atoms, const Multi& m_samples, int xi,
                              float tr_sigma, float rot_norm, hipStream_t s) {
  hipLaunchKernelGGL(center_msg_kernel, dim3(m_atoms.off[m_atoms.n]), dim3(128), 0, s, h, v, m_atoms, xi);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(center_final_kernel, dim3(m_samples.off[m_samples.n]), dim3(64), 0, s, h, v, m_samples, tr_sigma, rot_norm);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Torsion head (score_model.py:431-448, 650-664).
// tor_bond_conv.tp has two live paths (6x1o x T1 -> 32x0e, 6x1e x T1 -> 32x0o) where T1 is the 1o block of
// FullTensorProduct(sh(edge), Y2(bond)) = (3/sqrt2)(b b^T - I/3)(sqrt3 v)  (tests/test_kernel_math.py::tor_t1).
// Stage 1: one wave per (sample, bond): radius(lig_pos, bond_pos, 5) -> first `cap` atoms in index order.
// Stage 2 (tp_conv.hip::bond_conv_kernel): one wave per (sample, bond) on the matrix cores: edge MLPs + tensor product for
// its <= 32 neighbour slots, mean over slots, BatchNorm, tor_final_layer.
constexpr int TOR_SLOTS = 32;

__global__ __launch_bounds__(64) void bond_nb_kernel(Multi mm, float lig_r2, int cap) {
  int bond;
  const PoseBatch& PB = locate(mm, bond);
  const GraphStatic gs = PB.gs;
  const float* __restrict__ pos = PB.gd.pos;
  int* __restrict__ nb = PB.tor_nb;
  int* __restrict__ nb_cnt = PB.tor_nb_cnt;
  int* __restrict__ tor_edge_count = PB.gd.counts + 4;
  const int tid = threadIdx.x;
  const int b = bond / gs.R, rho = bond % gs.R, Nl = gs.Nl;
  const float* P = pos + (size_t)b * Nl * 3;
  const int u = gs.rot_u[rho], v = gs.rot_v[rho];
  const float bx = (P[3 * u] + P[3 * v]) / 2, by = (P[3 * u + 1] + P[3 * v + 1]) / 2, bz = (P[3 * u + 2] + P[3 * v + 2]) / 2;
  int n = 0;
  for (int c0 = 0; c0 < Nl && n < cap; c0 += 64) {
    const int a = c0 + tid;
    bool in = false;
    if (a < Nl) in = dist2_nofma(P[3 * a], P[3 * a + 1], P[3 * a + 2], bx, by, bz) < lig_r2;
    const unsigned long long m = __ballot(in);
    const int slot = n + popc_below(m, tid);
    if (in && slot < cap && slot < TOR_SLOTS) nb[(size_t)bond * TOR_SLOTS + slot] = a;
    n += __popcll(m);
  }
  if (tid == 0) {
    const int ne = min(n, min(cap, TOR_SLOTS));
    nb_cnt[bond] = ne;
    atomicAdd(tor_edge_count, ne);
  }
}

// m: B * R workgroups per batch
hipError_t launch_bond_nb(const Multi& m, float lig_r, int cap, hipStream_t s) {
  if (m.off[m.n] <= 0) return hipSuccess;
  hipLaunchKernelGGL(bond_nb_kernel, dim3(m.off[m.n]), dim3(64), 0, s, m, lig_r * lig_r, cap);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Reverse-SDE perturbation (utils/sampling.py:119-141) + modify_conformer_batch (utils/diffusion_utils.py:60-78):
// rigid update about the centroid, R sequential torsion rotations (utils/torsion.py:75-90, order dependent),
// Kabsch re-alignment of the flexible pose onto the rigid one (utils/geometry.py:246-276; closed form via Horn's
// quaternion + fp64 Jacobi, tests/test_kernel_math.py::horn_rotation).  One wave per sample, one lane per atom.
CBD_DEV void axis_angle_to_matrix(float ax, float ay, float az, float (&R)[9]) {
  // via quaternion, incl. the |angle| < 1e-6 series branch (utils/geometry.py:39-86)
  const float ang = sqrtf(ax * ax + ay * ay + az * az);
  const float half = 0.5f * ang;
  const float k = fabsf(ang) < 1e-6f ? 0.5f - (ang * ang) / 48.f : sinf(half) / ang;
  const float r = cosf(half), i = ax * k, j = ay * k, kk = az * k;
  const float two_s = 2.0f / (r * r + i * i + j * j + kk * kk);
  R[0] = 1 - two_s * (j * j + kk * kk); R[1] = two_s * (i * j - kk * r);     R[2] = two_s * (i * kk + j * r);
  R[3] = two_s * (i * j + kk * r);     R[4] = 1 - two_s * (i * i + kk * kk); R[5] = two_s * (j * kk - i * r);
  R[6] = two_s * (i * kk - j * r);     R[7] = two_s * (j * kk + i * r);     R[8] = 1 - two_s * (i * i + j * j);
}

CBD_DEV float wave_sum(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
CBD_DEV double wave_sum_d(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(64) void pose_update_kernel(Multi mm, int step, SdeCoefs cf, int use_coefs, int with_torsion) {
  extern __shared__ float sp[];   // [Nl][3] flexible pose, [Nl][3] rigid pose
  int b;
  const PoseBatch& PB = locate(mm, b);
  const GraphStatic gs = PB.gs;
  const int lane = lane_id();
  const int Nl = gs.Nl, R = gs.R, B = PB.B;
  float* __restrict__ pos = PB.gd.pos;
  const float* __restrict__ tr = PB.tr_out;
  const float* __restrict__ rot = PB.rot_out;
  const float* __restrict__ tor = with_torsion ? PB.tor_out : nullptr;
  // noise rows of this step; a term whose coefficient is 0 is skipped (no_final_step_noise / ODE, utils/sampling.py:119-141)
  const float* __restrict__ z_tr = (use_coefs && PB.z_tr && cf.tr_n != 0.f) ? PB.z_tr + (size_t)step * B * 3 : nullptr;
  const float* __restrict__ z_rot = (use_coefs && PB.z_rot && cf.rot_n != 0.f) ? PB.z_rot + (size_t)step * B * 3 : nullptr;
  const float* __restrict__ z_tor = (use_coefs && PB.z_tor && cf.tor_n != 0.f) ? PB.z_tor + (size_t)step * B * R : nullptr;
  float* flex = sp;
  float* rigid = sp + 3 * Nl;
  float* P = pos + (size_t)b * Nl * 3;
  // perturbations
  float trp[3], rotp[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (use_coefs) {
      trp[c] = cf.tr_s * tr[b * 3 + c] + (z_tr ? cf.tr_n * z_tr[b * 3 + c] : 0.f);
      rotp[c] = cf.rot_s * rot[b * 3 + c] + (z_rot ? cf.rot_n * z_rot[b * 3 + c] : 0.f);
    } else {
      trp[c] = tr[b * 3 + c];
      rotp[c] = rot[b * 3 + c];
    }
  }
  // centroid
  float cx = 0.f, cy = 0.f, cz = 0.f;
  for (int a = lane; a < Nl; a += 64) { cx += P[3 * a]; cy += P[3 * a + 1]; cz += P[3 * a + 2]; }
  cx = wave_sum(cx) / (float)Nl; cy = wave_sum(cy) / (float)Nl; cz = wave_sum(cz) / (float)Nl;
  float Rm[9];
  axis_angle_to_matrix(rotp[0], rotp[1], rotp[2], Rm);
  for (int a = lane; a < Nl; a += 64) {
    const float x = P[3 * a] - cx, y = P[3 * a + 1] - cy, z = P[3 * a + 2] - cz;
    const float nx = Rm[0] * x + Rm[1] * y + Rm[2] * z + trp[0] + cx;
    const float ny = Rm[3] * x + Rm[4] * y + Rm[5] * z + trp[1] + cy;
    const float nz = Rm[6] * x + Rm[7] * y + Rm[8] * z + trp[2] + cz;
    rigid[3 * a] = nx; rigid[3 * a + 1] = ny; rigid[3 * a + 2] = nz;
    flex[3 * a] = nx; flex[3 * a + 1] = ny; flex[3 * a + 2] = nz;
  }
  __syncthreads();
  if (tor == nullptr || R == 0) {
    for (int a = lane; a < Nl; a += 64) { P[3 * a] = rigid[3 * a]; P[3 * a + 1] = rigid[3 * a + 1]; P[3 * a + 2] = rigid[3 * a + 2]; }
    return;
  }
  // sequential torsions on the already-updated coordinates
  for (int rho = 0; rho < R; ++rho) {
    float th = use_coefs ? cf.tor_s * tor[b * R + rho] + (z_tor ? cf.tor_n * z_tor[b * R + rho] : 0.f) : tor[b * R + rho];
    const int u = gs.rot_u[rho], v = gs.rot_v[rho];
    const float vx = flex[3 * v], vy = flex[3 * v + 1], vz = flex[3 * v + 2];
    float ax = flex[3 * u] - vx, ay = flex[3 * u + 1] - vy, az = flex[3 * u + 2] - vz;
    const float n = sqrtf(ax * ax + ay * ay + az * az);
    ax = ax / n * th; ay = ay / n * th; az = az / n * th;
    float Q[9];
    axis_angle_to_matrix(ax, ay, az, Q);
    __syncthreads();
    for (int a = lane; a < Nl; a += 64) {
      if (gs.mask_rotate[rho * Nl + a]) {
        const float x = flex[3 * a] - vx, y = flex[3 * a + 1] - vy, z = flex[3 * a + 2] - vz;
        flex[3 * a] = Q[0] * x + Q[1] * y + Q[2] * z + vx;
        flex[3 * a + 1] = Q[3] * x + Q[4] * y + Q[5] * z + vy;
        flex[3 * a + 2] = Q[6] * x + Q[7] * y + Q[8] * z + vz;
      }
    }
    __syncthreads();
  }
  // Kabsch: R, t minimising |R flex + t - rigid|
  float fa[3] = {0, 0, 0}, fb[3] = {0, 0, 0};
  for (int a = lane; a < Nl; a += 64)
    for (int c = 0; c < 3; ++c) { fa[c] += flex[3 * a + c]; fb[c] += rigid[3 * a + c]; }
  for (int c = 0; c < 3; ++c) { fa[c] = wave_sum(fa[c]) / (float)Nl; fb[c] = wave_sum(fb[c]) / (float)Nl; }
  double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int a = lane; a < Nl; a += 64) {
    const float am[3] = {flex[3 * a] - fa[0], flex[3 * a + 1] - fa[1], flex[3 * a + 2] - fa[2]};
    const float bm[3] = {rigid[3 * a] - fb[0], rigid[3 * a + 1] - fb[1], rigid[3 * a + 2] - fb[2]};
    for (int i = 0; i < 3; ++i)
      for (int k = 0; k < 3; ++k) S[3 * i + k] += (double)(am[i] * bm[k]);
  }
  for (int i = 0; i < 9; ++i) S[i] = wave_sum_d(S[i]);
  double N[4][4] = {{S[0] + S[4] + S[8], S[5] - S[7], S[6] - S[2], S[1] - S[3]},
                    {S[5] - S[7], S[0] - S[4] - S[8], S[1] + S[3], S[6] + S[2]},
                    {S[6] - S[2], S[1] + S[3], -S[0] + S[4] - S[8], S[5] + S[7]},
                    {S[1] - S[3], S[6] + S[2], S[5] + S[7], -S[0] - S[4] + S[8]}};
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 12; ++sweep) {
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        const double apq = N[p][q];
        if (fabs(apq) < 1e-280) continue;
        const double th = (N[q][q] - N[p][p]) / (2.0 * apq);
        const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {   // N <- N J
          const double nkp = N[k][p], nkq = N[k][q];
          N[k][p] = c * nkp - s * nkq; N[k][q] = s * nkp + c * nkq;
        }
        for (int k = 0; k < 4; ++k) {   // N <- J^T N
          const double npk = N[p][k], nqk = N[q][k];
          N[p][k] = c * npk - s * nqk; N[q][k] = s * npk + c * nqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k][p], vkq = V[k][q];
          V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
        }
      }
  }
  int best = 0;
  for (int k = 1; k < 4; ++k) if (N[k][k] > N[best][best]) best = k;
  const double w = V[0][best], x = V[1][best], y = V[2][best], z = V[3][best];
  const float Rk[9] = {(float)(w * w + x * x - y * y - z * z), (float)(2 * (x * y - w * z)), (float)(2 * (x * z + w * y)),
                       (float)(2 * (x * y + w * z)), (float)(w * w - x * x + y * y - z * z), (float)(2 * (y * z - w * x)),
                       (float)(2 * (x * z - w * y)), (float)(2 * (y * z + w * x)), (float)(w * w - x * x - y * y + z * z)};
  // t = -R ca + cb ; aligned = flex R^T + t
  const float tx = fb[0] - (Rk[0] * fa[0] + Rk[1] * fa[1] + Rk[2] * fa[2]);
  const float ty = fb[1] - (Rk[3] * fa[0] + Rk[4] * fa[1] + Rk[5] * fa[2]);
  const float tz = fb[2] - (Rk[6] * fa[0] + Rk[7] * fa[1] + Rk[8] * fa[2]);
  for (int a = lane; a < Nl; a += 64) {
    const float fx = flex[3 * a], fy = flex[3 * a + 1], fz = flex[3 * a + 2];
    P[3 * a] = Rk[0] * fx + Rk[1] * fy + Rk[2] * fz + tx;
    P[3 * a + 1] = Rk[3] * fx + Rk[4] * fy + Rk[5] * fz + ty;
    P[3 * a + 2] = Rk[6] * fx + Rk[7] * fy + Rk[8] * fz + tz;
  }
}

// m: B workgroups per batch; max_nl: the largest Nl among the batches (dynamic LDS)
hipError_t launch_pose_update(const Multi& m, int step, const SdeCoefs& coefs, int use_coefs, int with_torsion, int max_nl, hipStream_t s) {
  hipLaunchKernelGGL(pose_update_kernel, dim3(m.off[m.n]), dim3(64), max_nl * 6 * sizeof(float), s, m, step, coefs, use_coefs, with_torsion);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Receptor node encoder: embedding(residue type) then Linear over [embedding | LM features]  (score_model.py:33-41,310).
__global__ __launch_bounds__(64) void rec_node_embed_kernel(const float* __restrict__ rec_x, int Nr, int lm_dim,
                                                            const float* __restrict__ emb, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ node) {
  const int r = blockIdx.x, lane = lane_id();
  const float* x = rec_x + (size_t)r * (1 + lm_dim);
  const int type = (int)x[0];
  const int in_dim = 32 + lm_dim;
  for (int c = lane; c < NODE_STRIDE; c += 64) {
    float v = 0.f;
    if (c < 32) {
      if (lm_dim > 0) {
        v = bias[c];
        const float* wr = w + (size_t)c * in_dim;
        for (int k = 0; k < 32; ++k) v = fmaf(wr[k], emb[type * 32 + k], v);
        for (int k = 0; k < lm_dim; ++k) v = fmaf(wr[32 + k], x[1 + k], v);
      } else {
        v = emb[type * 32 + c];
      }
    }
    node[(size_t)r * NODE_STRIDE + c] = v;
  }
}
hipError_t launch_rec_node_embed(const float* rec_x, int Nr, int lm_dim, const float* emb_table, const float* w, const float* b,
                                 float* node, hipStream_t s) {
  hipLaunchKernelGGL(rec_node_embed_kernel, dim3(Nr), dim3(64), 0, s, rec_x, Nr, lm_dim, emb_table, w, b, node);
  return hipGetLastError();
}

// vec4[e] = unit(pos[dst] - pos[src]), dist[e] = |.|      (receptor kNN edges, score_model.py:531-536)
__global__ void edge_geom_kernel(const float* __restrict__ pos, const int* __restrict__ src, const int* __restrict__ dst, int n,
                                 float* __restrict__ vec4, float* __restrict__ dist) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int s = src[e], d = dst[e];
  float ux, uy, uz, nn;
  unit_vec(pos[3 * d] - pos[3 * s], pos[3 * d + 1] - pos[3 * s + 1], pos[3 * d + 2] - pos[3 * s + 2], ux, uy, uz, nn);
  reinterpret_cast<f32x4*>(vec4)[e] = f32x4{ux, uy, uz, 0.f};
  dist[e] = nn;
}
hipError_t launch_edge_geom(const float* pos, const int* src, const int* dst, int n, float* vec4, float* dist, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(edge_geom_kernel, dim3((n + 255) / 256), dim3(256), 0, s, pos, src, dst, n, vec4, dist);
  return hipGetLastError();
}

__global__ void fill_i32_kernel(int* p, int v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
hipError_t launch_fill_i32(int* p, int v, int n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, p, v, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// Symmetry-corrected RMSD (reference utils/molecules_utils.py:3-18 -> spyrmsd/rmsd.py:116-203 with center=False,
// minimize=False): for pose b the minimum over graph isomorphisms k of sum_i |ref[idx_ref[k][i]] - pos[b][idx_pos[k][i]]|^2,
// rmsd = sqrt(min / n).  One wave per pose, lanes over atoms, fp64 accumulation; the first minimum wins (strict <).
__global__ __launch_bounds__(64) void symm_rmsd_kernel(int N, int K, const float* __restrict__ pos, const float* __restrict__ ref,
                                                       const int* __restrict__ idx_ref, const int* __restrict__ idx_pos,
                                                       float* __restrict__ out, int* __restrict__ argmin) {
  const int b = blockIdx.x, lane = lane_id();
  const float* P = pos + (size_t)b * N * 3;
  double best = 1.0e300;
  int best_k = 0;
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
    for (int i = lane; i < N; i += 64) {
      const int ir = idx_ref[(size_t)k * N + i], ip = idx_pos[(size_t)k * N + i];
      const double dx = (double)ref[3 * ir] - (double)P[3 * ip], dy = (double)ref[3 * ir + 1] - (double)P[3 * ip + 1],
                   dz = (double)ref[3 * ir + 2] - (double)P[3 * ip + 2];
      s += dx * dx + dy * dy + dz * dz;
    }
    s = wave_sum_d(s);
    if (s < best) { best = s; best_k = k; }
  }
  if (lane == 0) {
    out[b] = (float)sqrt(best / (double)N);
    if (argmin) argmin[b] = best_k;
  }
}

hipError_t launch_symm_rmsd(int B, int N, int K, const float* pos, const float* ref, const int* idx_ref, const int* idx_pos, float* out,
                            int* argmin, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(symm_rmsd_kernel, dim3(B), dim3(64), 0, s, N, K, pos, ref, idx_ref, idx_pos, out, argmin);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------------------
// Per-node part of the first Linear of a conv layer's FCBlocks (see ConvGroup::psrc): P[row][c] = sum_{k<32} WT[k][c] x[row][k].
// The 32 scalars of a row are the same address for 32 consecutive lanes, WT (12 KB per job) stays in L1/L2.
constexpr int PROJ_ROWS = 32;   // rows per 256-thread block: thread (rg = tid / 32, c = tid % 32) owns rows 4 rg .. 4 rg + 3, columns c, c + 32, c + 64
__global__ __launch_bounds__(256) void node_proj_kernel(ProjArgs a) {
  int t = blockIdx.x;
  int jb = -1, row0 = 0;
  for (int j = 0; j < a.n_jobs; ++j) {
    const int nb = (a.job[j].n + PROJ_ROWS - 1) / PROJ_ROWS;
    if (jb < 0) {
      if (t < nb) { jb = j; row0 = t * PROJ_ROWS; }
      else t -= nb;
    }
  }
  if (jb < 0) return;
  const ProjJob J = a.job[jb];
  const int c = threadIdx.x & 31, r0 = row0 + 4 * (threadIdx.x >> 5);
  float acc[4][3];
#pragma unroll
  for (int r = 0; r < 4; ++r) acc[r][0] = acc[r][1] = acc[r][2] = 0.f;
  const float* xr[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) xr[r] = J.node_in + (size_t)(J.lo + (r0 + r < J.n ? r0 + r : J.n - 1)) * NODE_STRIDE;   // clamped: no divergent loads
#pragma unroll
  for (int k4 = 0; k4 < NS / 4; ++k4) {
    f32x4 x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = reinterpret_cast<const f32x4*>(xr[r])[k4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const float* w = J.WT + (4 * k4 + kk) * KDIM + c;
      const float w0 = w[0], w1 = w[32], w2 = w[64];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float xv = x[r][kk];
        acc[r][0] = fmaf(xv, w0, acc[r][0]); acc[r][1] = fmaf(xv, w1, acc[r][1]); acc[r][2] = fmaf(xv, w2, acc[r][2]);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    if (r0 + r < J.n) {
      float* o = J.P + (size_t)(J.lo + r0 + r) * KDIM + c;
      o[0] = acc[r][0]; o[32] = acc[r][1]; o[64] = acc[r][2];
    }
}

hipError_t launch_node_proj(const ProjArgs& a, hipStream_t s) {
  int grid = 0;
  for (int j = 0; j < a.n_jobs; ++j) grid += (a.job[j].n + PROJ_ROWS - 1) / PROJ_ROWS;
  if (grid == 0) return hipSuccess;
  hipLaunchKernelGGL(node_proj_kernel, dim3(grid), dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace cbd
