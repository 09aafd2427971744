"""CPU (PyTorch fp32) restatement of the reference's all-atom CONFIDENCE model forward pass for the shipped
`workdir/pretrained_confidence/model_parameters.yml` architecture (all_atoms, ns=24, nv=6, sh_lmax=2 -> e3nn
FullyConnectedTensorProduct layers, 5 interaction layers with 9 edge groups (3 in the last), no embedding layers,
atom_confidence head, crop_beyond=20, dynamic_max_cross, eval mode), evaluated at t = 0 as utils/sampling.py does.

TEST INFRASTRUCTURE ONLY (oracle) -- see oracle/e3nn_ref.py header for the import rule.

Follows (file:line in /root/reference):
  crop_beyond                                   utils/utils.py:395-420         (called per pose, utils/sampling.py:245-250)
  set_time(.., 0, 0, 0, 0, ..)                  utils/sampling.py:253, utils/diffusion_utils.py:150-179
  TensorProductScoreModel.forward (all-atom)    models/all_atom_score_model.py:363-454
  embedding()                                   models/all_atom_score_model.py:284-361 (num_prot_emb_layers = 0)
  graph builders                                models/all_atom_score_model.py:516-626
  TensorProductConvLayer.forward                models/tensor_layers.py:195-217 (faster=False: e3nn FCTP)
  heads                                         models/all_atom_score_model.py:436-446
The reference's own wiring is pinned by oracle/make_golden.py (g8_confidence.npz), which drives the reference's class.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F

from . import e3nn_ref as e3
from . import graph_ref as gr
from .score_ref import sinusoidal_embedding, gaussian_smearing, mlp2, atom_encoder

REC_ATOM_FEATURE_DIMS = [38, 119, 23, 38]   # datasets/process_mols.py:114-119
SH_IRREPS = "1x0e+1x1o+1x2e"


@dataclass
class ConfConfig:
    ns: int = 24
    nv: int = 6
    sigma_embed_dim: int = 32
    embedding_scale: float = 10000.0
    lig_max_radius: float = 5.0        # args.max_radius
    rec_max_radius: float = 30.0       # ctor default
    cross_max_distance: float = 80.0
    distance_embed_dim: int = 32
    cross_distance_embed_dim: int = 32
    num_conv_layers: int = 5
    crop_beyond: float = 20.0
    lig_radius_cap: int = 32

    @property
    def irrep_seq(self):              # models/tensor_layers.py:12-27 without reduce_pseudoscalars
        ns, nv = self.ns, self.nv
        return [f"{ns}x0e", f"{ns}x0e+{nv}x1o", f"{ns}x0e+{nv}x1o+{nv}x1e", f"{ns}x0e+{nv}x1o+{nv}x1e+{ns}x0o"]


@dataclass
class AllAtomComplex:
    """One complex in the all-atom schema (datasets/process_mols.py:448-526), un-batched and un-cropped."""
    lig_x: torch.Tensor            # [Nl,16] int64
    lig_bond_index: torch.Tensor   # [2, 2*bonds]
    lig_bond_attr: torch.Tensor    # [2*bonds, 4]
    rec_x: torch.Tensor            # [Nr, 1+1280]
    rec_pos: torch.Tensor          # [Nr,3]
    rec_edge_index: torch.Tensor   # [2, Err]   (rec_contact)
    atom_x: torch.Tensor           # [Na,4]
    atom_pos: torch.Tensor         # [Na,3]
    atom_edge_index: torch.Tensor  # [2, Eaa]   (atom_contact)
    atom_res: torch.Tensor         # [Na] residue of every atom (atom_rec_contact row 1)
    extra: dict = field(default_factory=dict)


def subgraph(keep: torch.Tensor, edge_index: torch.Tensor):
    """torch_geometric.utils.subgraph(mask, edge_index, relabel_nodes=True)[0]: edges with both ends kept, in their
    original order, node ids renumbered by rank among the kept nodes."""
    m = keep[edge_index[0]] & keep[edge_index[1]]
    remap = torch.cumsum(keep.long(), 0) - 1
    return remap[edge_index[:, m]], m


def crop(cx: AllAtomComplex, lig_pos: torch.Tensor, cutoff: float):
    """utils/utils.py:395-420 for one pose.  Returns index tensors into the un-cropped complex + relabelled edges."""
    d2 = torch.sum((lig_pos.unsqueeze(0) - cx.rec_pos.unsqueeze(1)) ** 2, -1)
    res_keep = torch.any(d2 < cutoff ** 2, dim=1)
    atom_keep = res_keep[cx.atom_res]
    rec_remap = torch.cumsum(res_keep.long(), 0) - 1
    rr, _ = subgraph(res_keep, cx.rec_edge_index)
    aa, _ = subgraph(atom_keep, cx.atom_edge_index)
    new_res = rec_remap[cx.atom_res][atom_keep]
    ar = torch.stack([torch.arange(len(new_res)), new_res])
    return dict(res_idx=torch.nonzero(res_keep).squeeze(1), atom_idx=torch.nonzero(atom_keep).squeeze(1), rr=rr, aa=aa, ar=ar)


def fctp_conv_layer(w, prefix, node_attr, edge_index, edge_attr_groups, edge_sh, in_irreps, out_irreps, n_groups):
    """TensorProductConvLayer.forward with e3nn FullyConnectedTensorProduct, residual, BatchNorm (eval), mean reduce."""
    tp = e3.FullyConnectedTensorProduct(in_irreps, SH_IRREPS, out_irreps, shared_weights=False)
    src, dst = edge_index
    ws = [mlp2(w, f"{prefix}.fc.{g}" if n_groups > 1 else f"{prefix}.fc", edge_attr_groups[g]) for g in range(n_groups)]
    weight = torch.cat(ws, dim=0)
    assert weight.shape[1] == tp.weight_numel
    msg = tp(node_attr[dst], edge_sh, weight)
    out = gr.scatter(msg, src, dim=0, dim_size=node_attr.shape[0], reduce="mean")
    bn = e3.BatchNorm(out_irreps)
    bn.weight.data, bn.bias.data = w[f"{prefix}.batch_norm.weight"], w[f"{prefix}.batch_norm.bias"]
    bn.running_mean, bn.running_var = w[f"{prefix}.batch_norm.running_mean"], w[f"{prefix}.batch_norm.running_var"]
    bn.eval()
    out = bn(out)
    return out + F.pad(node_attr, (0, out.shape[-1] - node_attr.shape[-1])), msg


def bn1d(w, prefix, x, eps=1e-5):
    return (x - w[f"{prefix}.running_mean"]) / torch.sqrt(w[f"{prefix}.running_var"] + eps) * w[f"{prefix}.weight"] + w[f"{prefix}.bias"]


def head(w, prefix, x):
    """Linear-BN1d-ReLU-Dropout-Linear-BN1d-ReLU-Dropout-Linear (eval), keys 0,1,4,5,8."""
    h = torch.relu(bn1d(w, f"{prefix}.1", F.linear(x, w[f"{prefix}.0.weight"], w[f"{prefix}.0.bias"])))
    h = torch.relu(bn1d(w, f"{prefix}.5", F.linear(h, w[f"{prefix}.4.weight"], w[f"{prefix}.4.bias"])))
    return F.linear(h, w[f"{prefix}.8.weight"], w[f"{prefix}.8.bias"])


@torch.no_grad()
def confidence_forward(w, cx: AllAtomComplex, pos: torch.Tensor, cfg: ConfConfig = ConfConfig(), record=False):
    """pos [B,Nl,3] -> {'confidence' [B], 'atom_confidence' [B*Nl,1]} (+ intermediates when record=True)."""
    B, Nl = pos.shape[0], pos.shape[1]
    ns = cfg.ns
    rec = {}
    crops = [crop(cx, pos[b], cfg.crop_beyond) for b in range(B)]
    n_res = [len(c["res_idx"]) for c in crops]
    n_atom = [len(c["atom_idx"]) for c in crops]
    r_off = np.concatenate([[0], np.cumsum(n_res)]).astype(int)
    a_off = np.concatenate([[0], np.cumsum(n_atom)]).astype(int)
    rec_x = torch.cat([cx.rec_x[c["res_idx"]] for c in crops])
    rec_pos = torch.cat([cx.rec_pos[c["res_idx"]] for c in crops])
    rec_batch = torch.cat([torch.full((n,), b, dtype=torch.long) for b, n in enumerate(n_res)])
    atom_x = torch.cat([cx.atom_x[c["atom_idx"]] for c in crops])
    atom_pos = torch.cat([cx.atom_pos[c["atom_idx"]] for c in crops])
    atom_batch = torch.cat([torch.full((n,), b, dtype=torch.long) for b, n in enumerate(n_atom)])
    rr = torch.cat([c["rr"] + r_off[b] for b, c in enumerate(crops)], dim=1)
    aa = torch.cat([c["aa"] + a_off[b] for b, c in enumerate(crops)], dim=1)
    ar = torch.cat([c["ar"] + torch.tensor([[a_off[b]], [r_off[b]]]) for b, c in enumerate(crops)], dim=1)
    lig_pos = pos.reshape(B * Nl, 3)
    lig_batch = torch.arange(B).repeat_interleave(Nl)
    lig_x = cx.lig_x.repeat(B, 1)
    bonds = torch.cat([cx.lig_bond_index + b * Nl for b in range(B)], dim=1)
    bond_attr = cx.lig_bond_attr.repeat(B, 1)

    def sh(vec):
        return e3.spherical_harmonics(SH_IRREPS, vec, normalize=True, normalization="component")

    # ---- time embedding at t = 0 (confidence_mode: sigmas are the raw times, all_atom_score_model.py:379-382)
    t0 = torch.zeros(B)
    sig_emb_graph = sinusoidal_embedding(cfg.embedding_scale * t0, cfg.sigma_embed_dim)
    rec_sigma_emb = mlp2(w, "rec_sigma_embedding", sig_emb_graph)                                  # [B, ns]

    # ---- receptor / atom embedding (embedding(): 284-343)
    rec_node = atom_encoder(w, "rec_node_embedding", rec_x, 1)
    rr_vec = rec_pos[rr[1]] - rec_pos[rr[0]]
    rr_attr = mlp2(w, "rec_edge_embedding", gaussian_smearing(rr_vec.norm(dim=-1), 0.0, cfg.rec_max_radius, cfg.distance_embed_dim))
    atom_node = atom_encoder(w, "atom_node_embedding", atom_x, 4)
    aa_vec = atom_pos[aa[1]] - atom_pos[aa[0]]
    aa_attr = mlp2(w, "atom_edge_embedding", gaussian_smearing(aa_vec.norm(dim=-1), 0.0, cfg.lig_max_radius, cfg.distance_embed_dim))
    ar_vec = rec_pos[ar[1]] - atom_pos[ar[0]]
    ar_attr = mlp2(w, "ar_edge_embedding", gaussian_smearing(ar_vec.norm(dim=-1), 0.0, cfg.rec_max_radius, cfg.distance_embed_dim))
    rr_sh, aa_sh, ar_sh = sh(rr_vec), sh(aa_vec), sh(ar_vec)
    rec_node = rec_node + 0
    rec_node[:, :ns] = rec_node[:, :ns] + rec_sigma_emb[rec_batch]
    rr_attr = rr_attr + rec_sigma_emb[rec_batch[rr[0]]]
    atom_node[:, :ns] = atom_node[:, :ns] + rec_sigma_emb[atom_batch]
    aa_attr = aa_attr + rec_sigma_emb[atom_batch[aa[0]]]
    ar_attr = ar_attr + rec_sigma_emb[atom_batch[ar[0]]]

    # ---- ligand graph (build_lig_conv_graph: 516-554)
    node_sig = sinusoidal_embedding(cfg.embedding_scale * torch.zeros(B * Nl), cfg.sigma_embed_dim)
    radius_edges = gr.radius_graph(lig_pos, cfg.lig_max_radius, lig_batch, max_num_neighbors=cfg.lig_radius_cap)
    ll = torch.cat([bonds, radius_edges], 1).long()
    ll_attr = torch.cat([bond_attr, torch.zeros(radius_edges.shape[1], 4)], 0)
    ll_attr = torch.cat([ll_attr, node_sig[ll[0]]], 1)
    ll_vec = lig_pos[ll[1]] - lig_pos[ll[0]]
    ll_attr = torch.cat([ll_attr, gaussian_smearing(ll_vec.norm(dim=-1), 0.0, cfg.lig_max_radius, cfg.distance_embed_dim)], 1)
    ll_attr = mlp2(w, "lig_edge_embedding", ll_attr)
    ll_sh = sh(ll_vec)
    lig_node = atom_encoder(w, "lig_node_embedding", torch.cat([lig_x.float(), node_sig], 1), 16)
    lig_node = F.pad(lig_node, (0, rec_node.shape[-1] - lig_node.shape[-1]))

    # ---- cross graphs (build_cross_lig_conv_graph: 586-621); dynamic_max_cross with sigma = t = 0 -> 20 A
    cutoff = (t0 * 3 + 20).unsqueeze(1)
    lr = gr.radius(rec_pos / cutoff[rec_batch], lig_pos / cutoff[lig_batch], 1, rec_batch, lig_batch, max_num_neighbors=10000)
    lr_vec = rec_pos[lr[1]] - lig_pos[lr[0]]
    lr_attr = torch.cat([node_sig[lr[0]], gaussian_smearing(lr_vec.norm(dim=-1), 0.0, cfg.cross_max_distance, cfg.cross_distance_embed_dim)], 1)
    lr_attr = mlp2(w, "lr_edge_embedding", lr_attr)
    lr_sh = sh(lr_vec)
    la = gr.radius(atom_pos, lig_pos, cfg.lig_max_radius, atom_batch, lig_batch, max_num_neighbors=10000)
    la_vec = atom_pos[la[1]] - lig_pos[la[0]]
    la_attr = torch.cat([node_sig[la[0]], gaussian_smearing(la_vec.norm(dim=-1), 0.0, cfg.lig_max_radius, cfg.distance_embed_dim)], 1)
    la_attr = mlp2(w, "la_edge_embedding", la_attr)
    la_sh = sh(la_vec)

    # ---- joint graph [lig; rec; atom], 9 edge groups (forward: 396-421)
    n_lig, n_rec = B * Nl, rec_node.shape[0]
    node_attr = torch.cat([lig_node, rec_node, atom_node], 0)
    rr_j = rr + n_lig
    aa_j = aa + n_lig + n_rec
    lr_j = torch.stack([lr[0], lr[1] + n_lig])
    la_j = torch.stack([la[0], la[1] + n_lig + n_rec])
    ar_j = torch.stack([ar[0] + n_lig + n_rec, ar[1] + n_lig])
    groups_index = [ll, lr_j, la_j, rr_j, lr_j.flip(0), ar_j.flip(0), aa_j, la_j.flip(0), ar_j]
    groups_attr = [ll_attr, lr_attr, la_attr, rr_attr, lr_attr, ar_attr, aa_attr, la_attr, ar_attr]
    groups_sh = [ll_sh, lr_sh, la_sh, rr_sh, lr_sh, ar_sh, aa_sh, la_sh, ar_sh]
    if record:
        rec.update(n_res=np.array(n_res), n_atom=np.array(n_atom), node_attr0=node_attr.clone(),
                   edge_counts=np.array([g.shape[1] for g in groups_index]),
                   **{f"attr_{k}": a for k, a in zip(("ll", "lr", "la", "rr", "ar", "aa"), (ll_attr, lr_attr, la_attr, rr_attr, ar_attr, aa_attr))})
    seq = cfg.irrep_seq
    for l in range(cfg.num_conv_layers):
        ng = 9 if l < cfg.num_conv_layers - 1 else 3
        ei = torch.cat(groups_index[:ng], 1)
        esh = torch.cat(groups_sh[:ng], 0)
        attrs = [torch.cat([groups_attr[g], node_attr[groups_index[g][0], :ns], node_attr[groups_index[g][1], :ns]], -1) for g in range(ng)]
        node_attr, _ = fctp_conv_layer(w, f"conv_layers.{l}", node_attr, ei, attrs, esh, seq[min(l, 3)], seq[min(l + 1, 3)], ng)
        if record:
            rec[f"node_attr{l + 1}"] = node_attr.clone()
    lig_out = node_attr[:n_lig]
    scalar = torch.cat([lig_out[:, :ns], lig_out[:, -ns:]], dim=1)
    scalar = head(w, "atom_confidence_predictor", scalar)
    atom_conf = scalar[:, :1]
    scalar = scalar[:, 1:]
    conf = head(w, "confidence_predictor", gr.scatter_mean(scalar, lig_batch, dim=0, dim_size=B)).squeeze(-1)
    rec.update(confidence=conf, atom_confidence=atom_conf)
    return rec
