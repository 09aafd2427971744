"""CPU (PyTorch fp32) restatement of the reference score model's forward pass for the shipped
`workdir/pretrained_score/model_parameters.yml` architecture (sh_lmax=1, ns=32, nv=6, 3+3 embedding
layers, 5 interaction layers, reduce_pseudoscalars, embed_also_ligand, dynamic_max_cross,
fixed_center_conv, scale_by_sigma, eval mode).

TEST INFRASTRUCTURE ONLY (oracle) -- see oracle/e3nn_ref.py header for the import rule.  It is also
the timed "port" CPU baseline of bench.py (materialised [E, W] per-edge weights, FasterTensorProduct
arithmetic as in the reference, index_add mean).

Follows (file:line in /root/reference):
  TensorProductScoreModel.forward          models/score_model.py:333-449
  embedding / ligand_embedding             models/score_model.py:282-331
  graph builders                           models/score_model.py:492-539,564-587,635-664
  GaussianSmearing / AtomEncoder           models/score_model.py:667-677 / 18-41
  TensorProductConvLayer.forward           models/tensor_layers.py:195-217
  FasterTensorProduct.forward              models/tensor_layers.py:66-117
  FCBlock                                  models/layers.py:8-15
  sinusoidal_embedding, t_to_sigma         utils/diffusion_utils.py:99-110, 21-32
  so3.score_norm / torus.score_norm        utils/so3.py:90-94 / utils/torus.py:78-82

Written as one functional pass over explicit tensors (no PyG containers) that records every
intermediate the GPU parity tests compare against.  The reference's own wiring is pinned separately by
oracle/make_golden.py, which drives the reference's TensorProductScoreModel class with these primitives
as shims and stores the result in tests/golden/.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import e3nn_ref as e3
from . import graph_ref as gr

NS, NV = 32, 6
LIG_FEATURE_DIMS = [119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2]  # datasets/process_mols.py:95-112
REC_FEATURE_DIMS = [38]                                                   # datasets/process_mols.py:121-123
IRREP_SEQ = ["32x0e", "32x0e+6x1o", "32x0e+6x1o+6x1e", "32x0e+6x1o+6x1e+6x0o"]  # tensor_layers.py:21-26


@dataclass
class ScoreConfig:
    """Subset of model_parameters.yml that shapes the forward pass (utils/utils.py:239-283 mapping)."""
    ns: int = 32
    nv: int = 6
    sigma_embed_dim: int = 32
    embedding_scale: float = 1000.0
    lig_max_radius: float = 5.0      # args.max_radius
    rec_max_radius: float = 30.0     # ctor default
    cross_max_distance: float = 80.0
    center_max_distance: float = 30.0
    distance_embed_dim: int = 32
    cross_distance_embed_dim: int = 32
    num_conv_layers: int = 5
    num_prot_emb_layers: int = 3
    tr_sigma_min: float = 0.1
    tr_sigma_max: float = 19.0
    rot_sigma_min: float = 0.06
    rot_sigma_max: float = 3.1
    tor_sigma_min: float = 0.0314
    tor_sigma_max: float = 3.14
    no_torsion: bool = False
    lig_radius_cap: int = 32         # torch_cluster default max_num_neighbors
    bond_radius_cap: int = 32


@dataclass
class ComplexData:
    """One protein-ligand complex in the reference's graph schema, un-batched."""
    lig_x: torch.Tensor            # [Nl,16] int64
    lig_bond_index: torch.Tensor   # [2, 2*bonds] int64, each bond twice, consecutive
    lig_bond_attr: torch.Tensor    # [2*bonds, 4] f32 one-hot
    edge_mask: torch.Tensor        # [2*bonds] bool (rotatable, one direction)
    mask_rotate: np.ndarray        # [R, Nl] bool
    rec_x: torch.Tensor            # [Nr, 1+1280] f32, col 0 = residue type
    rec_pos: torch.Tensor          # [Nr,3] f32
    rec_edge_index: torch.Tensor   # [2, Err] int64
    extra: dict = field(default_factory=dict)

    @property
    def Nl(self):
        return self.lig_x.shape[0]

    @property
    def Nr(self):
        return self.rec_x.shape[0]

    @property
    def R(self):
        return int(self.edge_mask.sum())


# ----------------------------------------------------------------------------- small pieces
def t_to_sigma(t_tr, t_rot, t_tor, cfg: ScoreConfig):
    return (cfg.tr_sigma_min ** (1 - t_tr) * cfg.tr_sigma_max ** t_tr,
            cfg.rot_sigma_min ** (1 - t_rot) * cfg.rot_sigma_max ** t_rot,
            cfg.tor_sigma_min ** (1 - t_tor) * cfg.tor_sigma_max ** t_tor)


def sinusoidal_embedding(timesteps: torch.Tensor, dim: int, max_positions=10000):
    half = dim // 2
    k = math.log(max_positions) / (half - 1)
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * -k)
    arg = timesteps.float()[:, None] * freqs[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)


def gaussian_smearing(dist, start, stop, n):
    offset = torch.linspace(start, stop, n)
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
    d = dist.view(-1, 1) - offset.view(1, -1)
    return torch.exp(coeff * torch.pow(d, 2))


def mlp2(w, prefix, x, i0="0", i1="3", act=torch.relu):
    """nn.Sequential(Linear, ReLU, Dropout, Linear) in eval mode."""
    h = F.linear(x, w[f"{prefix}.{i0}.weight"], w.get(f"{prefix}.{i0}.bias"))
    h = act(h)
    return F.linear(h, w[f"{prefix}.{i1}.weight"], w.get(f"{prefix}.{i1}.bias"))


def atom_encoder(w, prefix, x, n_cat):
    emb = 0
    for i in range(n_cat):
        emb = emb + F.embedding(x[:, i].long(), w[f"{prefix}.atom_embedding_list.{i}.weight"])
    extra = x[:, n_cat:]
    if extra.shape[1] > 0:
        emb = F.linear(torch.cat([emb, extra.to(emb.dtype)], dim=1),
                       w[f"{prefix}.additional_features_embedder.weight"],
                       w[f"{prefix}.additional_features_embedder.bias"])
    return emb


def so3_score_norm(table: np.ndarray, eps: np.ndarray):
    MIN_EPS, MAX_EPS, N_EPS = 0.0005, 4, 2000
    idx = (np.log10(eps) - np.log10(MIN_EPS)) / (np.log10(MAX_EPS) - np.log10(MIN_EPS)) * N_EPS
    idx = np.clip(np.around(idx).astype(int), a_min=0, a_max=N_EPS - 1)
    return torch.from_numpy(np.asarray(table)[idx]).float()


def torus_score_norm(table: np.ndarray, sigma: np.ndarray):
    SIGMA_MIN, SIGMA_MAX, SIGMA_N = 3e-3, 2, 5000
    s = np.log(sigma / np.pi)
    s = (s - np.log(SIGMA_MIN)) / (np.log(SIGMA_MAX) - np.log(SIGMA_MIN)) * SIGMA_N
    s = np.round(np.clip(s, 0, SIGMA_N)).astype(int)
    return np.asarray(table)[s]


# ----------------------------------------------------------------------------- tensor product conv
def irreps_muls(irreps: str):
    muls = {"0e": 0, "1o": 0, "1e": 0, "0o": 0}
    for m, ir in e3.Irreps(irreps):
        muls[str(ir)] = m
    return muls


def faster_tp_weight_shapes(in_irreps: str, out_irreps: str):
    i, o = irreps_muls(in_irreps), irreps_muls(out_irreps)
    return {"0e": (i["0e"] + i["1o"], o["0e"]),
            "1o": (i["0e"] + i["1o"] + i["1e"], o["1o"]),
            "1e": (i["1o"] + i["1e"] + i["0o"], o["1e"]),
            "0o": (i["1e"] + i["0o"], o["0o"])}


def faster_tp_weight_numel(in_irreps, out_irreps):
    return sum(a * b for a, b in faster_tp_weight_shapes(in_irreps, out_irreps).values())


def faster_tensor_product(x, sh, weight, in_irreps: str, out_irreps: str):
    """lmax=1 tensor product with per-edge weights (reference FasterTensorProduct)."""
    im = irreps_muls(in_irreps)
    E = x.shape[0]
    parts, off = {}, 0
    for key in ("0e", "1o", "1e", "0o"):
        m = im[key]
        if m == 0:
            continue
        if key[0] == "1":
            parts[key] = x[:, off:off + 3 * m].reshape(E, m, 3)
            off += 3 * m
        else:
            parts[key] = x[:, off:off + m]
            off += m
    s, v = sh[:, 0], sh[:, 1:]
    mids = {"0e": [], "1o": [], "1e": [], "0o": []}
    if "0e" in parts:
        mids["0e"].append(parts["0e"] * s[:, None])
        mids["1o"].append(parts["0e"][:, :, None] * v[:, None, :])
    if "1o" in parts:
        mids["0e"].append((parts["1o"] * v[:, None, :]).sum(-1) / np.sqrt(3))
        mids["1o"].append(parts["1o"] * s[:, None, None])
        mids["1e"].append(torch.linalg.cross(parts["1o"], v[:, None, :].expand_as(parts["1o"]), dim=-1) / np.sqrt(2))
    if "1e" in parts:
        mids["1o"].append(torch.linalg.cross(parts["1e"], v[:, None, :].expand_as(parts["1e"]), dim=-1) / np.sqrt(2))
        mids["1e"].append(parts["1e"] * s[:, None, None])
        mids["0o"].append((parts["1e"] * v[:, None, :]).sum(-1) / np.sqrt(3))
    if "0o" in parts:
        mids["1e"].append(parts["0o"][:, :, None] * v[:, None, :])
        mids["0o"].append(parts["0o"] * s[:, None])
    shapes = faster_tp_weight_shapes(in_irreps, out_irreps)
    out, start = {}, 0
    for key in ("0e", "1o", "1e", "0o"):
        fin, mo = shapes[key]
        wk = weight[:, start:start + fin * mo].reshape(E, fin, mo) / np.sqrt(fin) if fin * mo > 0 else None
        start += fin * mo
        if mo == 0 or not mids[key]:
            continue
        if key[0] == "0":
            mid = torch.cat(mids[key], dim=-1)                       # [E, fin]
            out[key] = torch.matmul(mid[:, None, :], wk).squeeze(1)  # [E, mo]
        else:
            mid = torch.cat(mids[key], dim=-2)                       # [E, fin, 3]
            out[key] = (mid[:, :, None, :] * wk[:, :, :, None]).sum(1).reshape(E, -1)
    om = irreps_muls(out_irreps)
    return torch.cat([out[k] for k in ("0e", "1o", "1e", "0o") if om[k] > 0], dim=-1)


def bn_eval(w, prefix, x, irreps: str, eps=1e-5):
    outs, ix, iw, ib = [], 0, 0, 0
    for m, ir in e3.Irreps(irreps):
        d = ir.dim
        f = x[:, ix:ix + m * d].reshape(-1, m, d)
        ix += m * d
        scalar = (ir.l == 0 and ir.p == 1)
        if scalar:
            f = f - w[f"{prefix}.running_mean"][ib:ib + m].reshape(1, m, 1)
        f = f * ((w[f"{prefix}.running_var"][iw:iw + m] + eps).pow(-0.5) * w[f"{prefix}.weight"][iw:iw + m]).reshape(1, m, 1)
        if scalar:
            f = f + w[f"{prefix}.bias"][ib:ib + m].reshape(1, m, 1)
            ib += m
        iw += m
        outs.append(f.reshape(-1, m * d))
    return torch.cat(outs, dim=-1)


def fc_block(w, prefix, x):
    return mlp2(w, prefix, x)


def tp_conv_faster(w, prefix, node_attr, edge_index, edge_attr_groups, edge_sh, in_irreps, out_irreps,
                   n_groups, out_nodes=None, residual=True, trace=None):
    """TensorProductConvLayer.forward with FasterTensorProduct, edge_groups in {1, n}."""
    out_dim = e3.Irreps(out_irreps).dim
    if edge_index.shape[1] == 0:
        out = torch.zeros(node_attr.shape[0], out_dim, dtype=node_attr.dtype)
    else:
        src, dst = edge_index[0], edge_index[1]
        if n_groups == 1:
            tpw = fc_block(w, f"{prefix}.fc", edge_attr_groups)
        else:
            tpw = torch.cat([fc_block(w, f"{prefix}.fc.{g}", edge_attr_groups[g]) for g in range(n_groups)], dim=0)
        msg = faster_tensor_product(node_attr[dst], edge_sh, tpw, in_irreps, out_irreps)
        n_out = out_nodes or node_attr.shape[0]
        out = gr.scatter(msg, src, dim=0, dim_size=n_out, reduce="mean")
        if trace is not None:
            trace["mean"] = out
        out = bn_eval(w, f"{prefix}.batch_norm", out, out_irreps)
    if residual:
        out = out + F.pad(node_attr, (0, out.shape[-1] - node_attr.shape[-1]))
    return out


# ----------------------------------------------------------------------------- full forward
@torch.no_grad()
def receptor_embedding(w: Dict[str, torch.Tensor], cx: ComplexData, cfg: ScoreConfig):
    """Time-independent part of embedding() (score_model.py:297-320), for ONE copy of the receptor."""
    src, dst = cx.rec_edge_index[0].long(), cx.rec_edge_index[1].long()
    vec = cx.rec_pos[dst] - cx.rec_pos[src]
    edge_attr = gaussian_smearing(vec.norm(dim=-1), 0.0, cfg.rec_max_radius, cfg.distance_embed_dim)
    edge_sh = e3.sh_l1(vec)
    node = atom_encoder(w, "rec_node_embedding", cx.rec_x, len(REC_FEATURE_DIMS))
    edge_attr = mlp2(w, "rec_edge_embedding", edge_attr)
    for l in range(cfg.num_prot_emb_layers):
        ea = torch.cat([edge_attr, node[src, :NS], node[dst, :NS]], -1)
        node = tp_conv_faster(w, f"rec_emb_layers.{l}", node, cx.rec_edge_index.long(), ea, edge_sh,
                              IRREP_SEQ[min(l, 3)], IRREP_SEQ[min(l + 1, 3)], 1)
    return node, edge_attr, edge_sh


@torch.no_grad()
def score_forward(w: Dict[str, torch.Tensor], cx: ComplexData, pos: torch.Tensor, t_tr: float, t_rot: float,
                  t_tor: float, cfg: ScoreConfig, so3_table, torus_table, rec_cache=None,
                  keep_intermediates: bool = True, t_common=None):
    """Score model forward for B poses `pos [B, Nl, 3]` of one complex at diffusion time (t_tr, t_rot, t_tor).
    Returns dict with tr_pred [B,3], rot_pred [B,3], tor_pred [B*R] and intermediates.
    `t_common`: the common time of a model with asyncronous_noise_schedule -- embedded by the ligand nodes and the graph-level heads
    (score_model.py:408,460,497) while the receptor keeps t_tr (score_model.py:323); None = an ordinary model."""
    B, Nl, Nr, R = pos.shape[0], cx.Nl, cx.Nr, cx.R
    T: Dict[str, torch.Tensor] = {}
    f32 = torch.float32
    # set_time (diffusion_utils.py:150-179) multiplies ones() by the schedule value -> fp32 tensors, and
    # t_to_sigma is then evaluated on those fp32 tensors (score_model.py:338)
    ct = [float(t) * torch.ones(1) for t in (t_tr, t_rot, t_tor)]
    tr_sigma_t, rot_sigma_t, tor_sigma_t = t_to_sigma(ct[0], ct[1], ct[2], cfg)
    T["tr_sigma"], T["rot_sigma"], T["tor_sigma"] = tr_sigma_t[0], rot_sigma_t[0], tor_sigma_t[0]
    t_emb_rec = sinusoidal_embedding(cfg.embedding_scale * ct[0], cfg.sigma_embed_dim)  # [1,32]: complex_t['tr'], receptor side
    t_emb_one = t_emb_rec if t_common is None else sinusoidal_embedding(cfg.embedding_scale * (float(t_common) * torch.ones(1)), cfg.sigma_embed_dim)
    T["sigma_emb"] = t_emb_one[0]

    lig_pos = pos.reshape(B * Nl, 3).to(f32)
    lig_batch = torch.arange(B).repeat_interleave(Nl)
    rec_batch = torch.arange(B).repeat_interleave(Nr)
    rec_pos = cx.rec_pos.repeat(B, 1)

    # ---------------- receptor embedding (cached, identical for all B copies) + sigma embedding
    if rec_cache is None:
        rec_cache = receptor_embedding(w, cx, cfg)
    rec_node0, rec_edge_attr0, rec_edge_sh0 = rec_cache
    T["rec_node_static"] = rec_node0
    rec_sigma_emb = mlp2(w, "rec_sigma_embedding", t_emb_rec)  # [1,32]
    T["rec_sigma_emb"] = rec_sigma_emb[0]
    rec_node = rec_node0.repeat(B, 1)
    rec_node[:, :NS] = rec_node[:, :NS] + rec_sigma_emb
    rec_edge_attr = rec_edge_attr0.repeat(B, 1) + rec_sigma_emb
    rec_edge_sh = rec_edge_sh0.repeat(B, 1)
    Err = cx.rec_edge_index.shape[1]
    rec_edge_index = torch.cat([cx.rec_edge_index.long() + b * Nr for b in range(B)], dim=1)

    # ---------------- ligand graph + embedding (score_model.py:492-522, 282-295)
    node_sigma_emb = t_emb_one.expand(B * Nl, -1)
    nb = cx.lig_bond_index.shape[1]
    bond_index = torch.cat([cx.lig_bond_index.long() + b * Nl for b in range(B)], dim=1)
    radius_edges = gr.radius_graph(lig_pos, cfg.lig_max_radius, lig_batch, max_num_neighbors=cfg.lig_radius_cap)
    lig_edge_index = torch.cat([bond_index, radius_edges], 1)
    lig_edge_attr = torch.cat([cx.lig_bond_attr.repeat(B, 1), torch.zeros(radius_edges.shape[1], 4)], 0)
    lig_edge_attr = torch.cat([lig_edge_attr, node_sigma_emb[lig_edge_index[0]]], 1)
    lsrc, ldst = lig_edge_index
    lvec = lig_pos[ldst] - lig_pos[lsrc]
    lig_edge_attr = torch.cat([lig_edge_attr, gaussian_smearing(lvec.norm(dim=-1), 0.0, cfg.lig_max_radius,
                                                                cfg.distance_embed_dim)], 1)
    lig_edge_sh = e3.sh_l1(lvec)
    lig_node_in = torch.cat([cx.lig_x.repeat(B, 1).to(f32), node_sigma_emb], 1)
    lig_node = atom_encoder(w, "lig_node_embedding", lig_node_in, len(LIG_FEATURE_DIMS))
    lig_edge_attr = mlp2(w, "lig_edge_embedding", lig_edge_attr)
    T["lig_node_emb0"] = lig_node
    T["lig_edge_index"] = lig_edge_index
    T["lig_edge_attr"] = lig_edge_attr
    for l in range(cfg.num_prot_emb_layers):
        ea = torch.cat([lig_edge_attr, lig_node[lsrc, :NS], lig_node[ldst, :NS]], -1)
        lig_node = tp_conv_faster(w, f"lig_emb_layers.{l}", lig_node, lig_edge_index, ea, lig_edge_sh,
                                  IRREP_SEQ[min(l, 3)], IRREP_SEQ[min(l + 1, 3)], 1)
        T[f"lig_emb_{l}"] = lig_node

    # ---------------- cross graph (score_model.py:345-352, 564-587)
    cutoff = (tr_sigma_t * 3 + 20).expand(B).unsqueeze(1)
    T["cross_cutoff"] = cutoff[0, 0]
    lr = gr.radius(rec_pos / cutoff[rec_batch], lig_pos / cutoff[lig_batch], 1, rec_batch, lig_batch,
                   max_num_neighbors=10000)
    csrc, cdst = lr[0], lr[1]
    cvec = rec_pos[cdst] - lig_pos[csrc]
    lr_edge_attr = torch.cat([node_sigma_emb[csrc], gaussian_smearing(cvec.norm(dim=-1), 0.0, cfg.cross_max_distance,
                                                                      cfg.cross_distance_embed_dim)], 1)
    lr_edge_sh = e3.sh_l1(cvec)
    rl_edge_sh = e3.sh_l1(-cvec)
    lr_edge_attr = mlp2(w, "cross_edge_embedding", lr_edge_attr)
    T["lr_edge_index"] = lr.clone()
    T["lr_edge_attr"] = lr_edge_attr

    # ---------------- joint graph + interaction layers (score_model.py:354-376)
    nL = B * Nl
    node = torch.cat([lig_node, rec_node], 0)
    lr_j = torch.stack([csrc, cdst + nL], 0)
    edge_index = torch.cat([lig_edge_index, lr_j, rec_edge_index + nL, torch.flip(lr_j, dims=[0])], 1)
    edge_attr = torch.cat([lig_edge_attr, lr_edge_attr, rec_edge_attr, lr_edge_attr], 0)
    edge_sh = torch.cat([lig_edge_sh, lr_edge_sh, rec_edge_sh, rl_edge_sh], 0)
    s1 = lig_edge_index.shape[1]
    s2 = s1 + lr_j.shape[1]
    s3 = s2 + rec_edge_index.shape[1]
    T["node_in"] = node
    irr = IRREP_SEQ[3]
    for l in range(cfg.num_conv_layers):
        if l < cfg.num_conv_layers - 1:
            ea = torch.cat([edge_attr, node[edge_index[0], :NS], node[edge_index[1], :NS]], -1)
            node = tp_conv_faster(w, f"conv_layers.{l}", node, edge_index, [ea[:s1], ea[s1:s2], ea[s2:s3], ea[s3:]],
                                  edge_sh, irr, irr, 4)
        else:
            ea = torch.cat([edge_attr[:s2], node[edge_index[0, :s2], :NS], node[edge_index[1, :s2], :NS]], -1)
            node = tp_conv_faster(w, f"conv_layers.{l}", node, edge_index[:, :s2], [ea[:s1], ea[s1:s2]],
                                  edge_sh[:s2], irr, irr, 2)
        T[f"conv_{l}"] = node
    lig_node = node[:nL]

    # ---------------- centre convolution -> tr / rot (score_model.py:393-420, 635-648)
    center = torch.zeros(B, 3).index_add_(0, lig_batch, lig_pos) / torch.bincount(lig_batch).unsqueeze(1)
    c_index = torch.stack([lig_batch, torch.arange(nL)], 0)
    c_vec = lig_pos[c_index[1]] - center[c_index[0]]
    c_attr = torch.cat([gaussian_smearing(c_vec.norm(dim=-1), 0.0, cfg.center_max_distance, cfg.distance_embed_dim),
                        node_sigma_emb], 1)
    c_sh = e3.sh_l1(c_vec)
    c_attr = mlp2(w, "center_edge_embedding", c_attr)
    c_attr = torch.cat([c_attr, lig_node[c_index[1], :NS]], -1)   # fixed_center_conv=True
    tp = e3.FullyConnectedTensorProduct(irr, "1x0e+1x1o", "2x1o+2x1e")
    tpw = fc_block(w, "final_conv.fc", c_attr)
    msg = tp(lig_node[c_index[1]], c_sh, tpw)
    gp = gr.scatter(msg, c_index[0], dim=0, dim_size=B, reduce="mean")
    T["center_mean"] = gp
    gp = bn_eval(w, "final_conv.batch_norm", gp, "2x1o+2x1e")
    T["global_pred"] = gp
    tr_pred = gp[:, :3] + gp[:, 6:9]
    rot_pred = gp[:, 3:6] + gp[:, 9:]
    graph_sigma_emb = t_emb_one.expand(B, -1)
    tr_norm = torch.linalg.vector_norm(tr_pred, dim=1).unsqueeze(1)
    tr_pred = tr_pred / tr_norm * mlp2(w, "tr_final_layer", torch.cat([tr_norm, graph_sigma_emb], 1), "0", "3")
    rot_norm = torch.linalg.vector_norm(rot_pred, dim=1).unsqueeze(1)
    rot_pred = rot_pred / rot_norm * mlp2(w, "rot_final_layer", torch.cat([rot_norm, graph_sigma_emb], 1), "0", "3")
    tr_pred = tr_pred / tr_sigma_t.expand(B).unsqueeze(1)
    T["so3_norm"] = so3_score_norm(so3_table, rot_sigma_t.numpy())[0]
    rot_pred = rot_pred * so3_score_norm(so3_table, rot_sigma_t.expand(B).numpy()).unsqueeze(1)
    T["tr_pred"], T["rot_pred"] = tr_pred, rot_pred

    if cfg.no_torsion or R == 0:
        T["tor_pred"] = torch.empty(0)
        return T

    # ---------------- torsion head (score_model.py:431-448, 650-664)
    mask = cx.edge_mask.repeat(B)
    bonds = bond_index[:, mask]
    bond_pos = (lig_pos[bonds[0]] + lig_pos[bonds[1]]) / 2
    bond_batch = lig_batch[bonds[0]]
    t_index = gr.radius(lig_pos, bond_pos, cfg.lig_max_radius, lig_batch, bond_batch,
                        max_num_neighbors=cfg.bond_radius_cap)
    t_vec = lig_pos[t_index[1]] - bond_pos[t_index[0]]
    t_attr = gaussian_smearing(t_vec.norm(dim=-1), 0.0, cfg.lig_max_radius, cfg.distance_embed_dim)
    t_attr = mlp2(w, "final_edge_embedding", t_attr)
    t_sh = e3.sh_l1(t_vec)
    bond_vec = lig_pos[bonds[1]] - lig_pos[bonds[0]]
    bond_attr = lig_node[bonds[0]] + lig_node[bonds[1]]
    bond_sh = e3.sh_l2(bond_vec)
    ftp = e3.FullTensorProduct("1x0e+1x1o", "2e")
    t_sh_full = ftp(t_sh, bond_sh[t_index[0]])
    T["tor_edge_index"] = t_index
    T["tor_edge_sh"] = t_sh_full
    t_attr = torch.cat([t_attr, lig_node[t_index[1], :NS], bond_attr[t_index[0], :NS]], -1)
    tp2 = e3.FullyConnectedTensorProduct(irr, ftp.irreps_out, "32x0o+32x0e")
    tpw2 = fc_block(w, "tor_bond_conv.fc", t_attr)
    msg2 = tp2(lig_node[t_index[1]], t_sh_full, tpw2)
    tor = gr.scatter(msg2, t_index[0], dim=0, dim_size=B * R, reduce="mean")
    T["tor_mean"] = tor
    tor = bn_eval(w, "tor_bond_conv.batch_norm", tor, "32x0o+32x0e")
    T["tor_feat"] = tor
    tor = F.linear(torch.tanh(F.linear(tor, w["tor_final_layer.0.weight"])), w["tor_final_layer.3.weight"]).squeeze(1)
    edge_sigma = tor_sigma_t.expand(B * R).numpy()
    tor_norm = torch.sqrt(torch.tensor(torus_score_norm(torus_table, edge_sigma)).float())
    T["torus_norm_sqrt"] = tor_norm[0]
    tor = tor * tor_norm
    T["tor_pred"] = tor
    return T
