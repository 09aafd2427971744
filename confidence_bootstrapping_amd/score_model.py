"""`TensorProductScoreModel` with the reference's constructor signature and `state_dict` layout
(reference models/score_model.py:44-280, models/tensor_layers.py:120-193), whose forward pass runs on the
MI355X engine (csrc/, hand-written HIP) instead of PyTorch/e3nn/torch_scatter/torch_cluster ops.

The module tree below exists to (a) own the parameters under the exact checkpoint key names so that
`load_state_dict(torch.load('best_ema_inference_epoch_model.pt'), strict=True)` works unchanged
(inference.py:298-309) and (b) hand them to the engine.  `model.eval()`: forward = the fused inference engine (no autograd).
`model.train()`: forward = the differentiable fine-tuning path (train_forward.py: the tensor-product layers on the HIP
forward/backward kernels of csrc/tp_train.hip, the small ops around them as autograd-visible torch ops on the same GPU).
There is NO CPU / pure-PyTorch fallback for either: if the HIP library is missing or the tensors are not on the GPU, it raises.

Supported architecture = the shipped `workdir/pretrained_score/model_parameters.yml` family:
sh_lmax=1, use_second_order_repr=False (=> FasterTensorProduct layers), reduce_pseudoscalars=True,
embed_also_ligand=True, differentiate_convolutions=True, tp_weights_layers=2, batch_norm,
lm_embedding_type in {None,'precomputed'}.  Anything else raises NotImplementedError at construction.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
from torch import nn

LIG_FEATURE_DIMS = ([119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2], 0)  # datasets/process_mols.py:95-112
REC_RESIDUE_FEATURE_DIMS = ([38], 0)                                           # datasets/process_mols.py:121-123


def get_irrep_seq(ns, nv, use_second_order_repr, reduce_pseudoscalars):
    """Irreps of the node features after 0,1,2,>=3 layers (reference models/tensor_layers.py:12-27)."""
    if use_second_order_repr:
        raise NotImplementedError("second-order representations are outside the MI355X hot path")
    last = nv if reduce_pseudoscalars else ns
    return [f"{ns}x0e", f"{ns}x0e + {nv}x1o", f"{ns}x0e + {nv}x1o + {nv}x1e",
            f"{ns}x0e + {nv}x1o + {nv}x1e + {last}x0o"]


def parse_irreps(s: str):
    """'32x0e + 6x1o' -> [(32, 0, +1), (6, 1, -1)]"""
    out = []
    for term in s.split("+"):
        term = term.strip()
        mul, ir = term.split("x")
        out.append((int(mul), int(ir[:-1]), 1 if ir[-1] == "e" else -1))
    return out


def irreps_dim(s: str) -> int:
    return sum(m * (2 * l + 1) for m, l, _ in parse_irreps(s))


def faster_tp_weight_numel(in_irreps: str, out_irreps: str) -> int:
    """Weight count of the lmax=1 tensor product (reference models/tensor_layers.py:53-64)."""
    def muls(s):
        d = {"0e": 0, "1o": 0, "1e": 0, "0o": 0}
        for m, l, p in parse_irreps(s):
            d[f"{l}{'e' if p == 1 else 'o'}"] = m
        return d
    i, o = muls(in_irreps), muls(out_irreps)
    return ((i["0e"] + i["1o"]) * o["0e"] + (i["0e"] + i["1o"] + i["1e"]) * o["1o"]
            + (i["1o"] + i["1e"] + i["0o"]) * o["1e"] + (i["1e"] + i["0o"]) * o["0o"])


class AtomEncoder(nn.Module):
    """Sum of categorical embeddings + Linear over [embedding, scalar/sigma/LM features]
    (reference models/score_model.py:18-41).  Parameter container; evaluated inside the engine."""

    def __init__(self, emb_dim, feature_dims, sigma_embed_dim, lm_embedding_dim=0):
        super().__init__()
        self.atom_embedding_list = nn.ModuleList()
        self.num_categorical_features = len(feature_dims[0])
        self.additional_features_dim = feature_dims[1] + sigma_embed_dim + lm_embedding_dim
        for dim in feature_dims[0]:
            emb = nn.Embedding(dim, emb_dim)
            nn.init.xavier_uniform_(emb.weight.data)
            self.atom_embedding_list.append(emb)
        if self.additional_features_dim > 0:
            self.additional_features_embedder = nn.Linear(self.additional_features_dim + emb_dim, emb_dim)


class GaussianSmearing(nn.Module):
    """exp(coeff (d - mu_k)^2) distance expansion (reference models/score_model.py:667-677)."""

    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)


class IrrepsBatchNorm(nn.Module):
    """Parameter container with e3nn.nn.BatchNorm's state_dict layout: weight[sum mul], bias[#0e],
    running_mean[#0e], running_var[sum mul]; no num_batches_tracked."""

    def __init__(self, irreps: str):
        super().__init__()
        ir = parse_irreps(irreps)
        nf = sum(m for m, _, _ in ir)
        nsc = sum(m for m, l, p in ir if l == 0 and p == 1)
        self.irreps = irreps
        self.weight = nn.Parameter(torch.ones(nf))
        self.bias = nn.Parameter(torch.zeros(nsc))
        self.register_buffer("running_mean", torch.zeros(nsc))
        self.register_buffer("running_var", torch.ones(nf))


def FCBlock(in_dim, hidden_dim, out_dim, dropout):
    """Linear-ReLU-Dropout-Linear; keys '0' and '3' (reference models/layers.py:8-15, layers=2)."""
    return nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.ReLU(), nn.Dropout(dropout), nn.Linear(hidden_dim, out_dim))


class TensorProductConvLayer(nn.Module):
    """Parameter container for one message-passing layer (reference models/tensor_layers.py:120-193)."""

    def __init__(self, in_irreps, sh_irreps, out_irreps, n_edge_features, residual=True, batch_norm=True,
                 dropout=0.0, hidden_features=None, weight_numel=None, edge_groups=1):
        super().__init__()
        self.in_irreps, self.out_irreps, self.sh_irreps = in_irreps, out_irreps, sh_irreps
        self.residual, self.edge_groups = residual, edge_groups
        hidden_features = hidden_features or n_edge_features
        self.weight_numel = weight_numel
        if edge_groups == 1:
            self.fc = FCBlock(n_edge_features, hidden_features, weight_numel, dropout)
        else:
            self.fc = nn.ModuleList([FCBlock(n_edge_features, hidden_features, weight_numel, dropout)
                                     for _ in range(edge_groups)])
        self.batch_norm = IrrepsBatchNorm(out_irreps) if batch_norm else None


def weights_version(module):
    """Identity of a module's weights for the engine cache: autograd version counters of parameters AND buffers (in-place
    optimiser / BatchNorm updates bump them) plus a checksum over a handful of tensors, which also catches writes through
    `.data` (the idiom of torch-ema style `copy_to` / `restore`), which leave `_version` untouched."""
    ts = list(module.parameters()) + [b for b in module.buffers() if b.is_floating_point()]
    v = sum(int(t._version) for t in ts)
    pick = [t for t in (ts[:2] + ts[len(ts) // 2:len(ts) // 2 + 2] + ts[-2:]) if t.numel() > 0]
    with torch.no_grad():
        chk = float(torch.stack([t.detach().double().sum() + t.detach().double().abs().sum() for t in pick]).sum()) if pick else 0.0
    return (v, len(ts), chk)


class TensorProductScoreModel(nn.Module):
    def __init__(self, t_to_sigma, device, timestep_emb_func, in_lig_edge_features=4, sigma_embed_dim=32, sh_lmax=2,
                 ns=16, nv=4, num_conv_layers=2, lig_max_radius=5, rec_max_radius=30, cross_max_distance=250,
                 center_max_distance=30, distance_embed_dim=32, cross_distance_embed_dim=32, no_torsion=False,
                 scale_by_sigma=True, norm_by_sigma=True, use_second_order_repr=False, batch_norm=True,
                 dynamic_max_cross=False, dropout=0.0, smooth_edges=False, odd_parity=False,
                 separate_noise_schedule=False, lm_embedding_type=None, confidence_mode=False,
                 confidence_dropout=0, confidence_no_batchnorm=False,
                 asyncronous_noise_schedule=False, affinity_prediction=False, parallel=1,
                 parallel_aggregators="mean max min std", num_confidence_outputs=1, atom_num_confidence_outputs=1,
                 fixed_center_conv=False, no_aminoacid_identities=False, include_miscellaneous_atoms=False,
                 differentiate_convolutions=True, tp_weights_layers=2, num_prot_emb_layers=0,
                 reduce_pseudoscalars=False, embed_also_ligand=False, atom_confidence=False, sidechain_pred=False,
                 depthwise_convolution=False, embedding_scale=None):
        super().__init__()
        unsupported = {
            "sh_lmax != 1": sh_lmax != 1, "use_second_order_repr": use_second_order_repr,
            "confidence_mode": confidence_mode, "separate_noise_schedule": separate_noise_schedule,
            "smooth_edges": smooth_edges,
            "odd_parity": odd_parity, "include_miscellaneous_atoms": include_miscellaneous_atoms,
            "sidechain_pred": sidechain_pred, "depthwise_convolution": depthwise_convolution,
            "not differentiate_convolutions": not differentiate_convolutions, "tp_weights_layers != 2": tp_weights_layers != 2,
            "not embed_also_ligand": not embed_also_ligand, "not batch_norm": not batch_norm,
            "not reduce_pseudoscalars": not reduce_pseudoscalars, "not scale_by_sigma": not scale_by_sigma,
            "not dynamic_max_cross": not dynamic_max_cross, "not fixed_center_conv": not fixed_center_conv,
            "no_aminoacid_identities": no_aminoacid_identities, "parallel != 1": parallel != 1,
            "ns != 32 or nv != 6": (ns, nv) != (32, 6), "num_prot_emb_layers != 3": num_prot_emb_layers != 3,
            "lm_embedding_type": lm_embedding_type not in (None, "precomputed"),
            "embed dims != 32": (sigma_embed_dim, distance_embed_dim, cross_distance_embed_dim) != (32, 32, 32),
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError("MI355X engine covers the shipped pretrained_score architecture only; "
                                      "unsupported options: " + ", ".join(bad))
        self.t_to_sigma = t_to_sigma
        self.device = device
        self.timestep_emb_func = timestep_emb_func
        # asyncronous_noise_schedule (score_model.py:85): no parameters of its own -- the ligand side and the tr / rot magnitude heads
        # embed the common time complex_t['t'] instead of complex_t['tr'] (engine.make_steps: cbd_step.sigma_emb_t)
        self.asyncronous_noise_schedule = bool(asyncronous_noise_schedule)
        self.in_lig_edge_features = in_lig_edge_features
        self.sigma_embed_dim = sigma_embed_dim
        self.lig_max_radius, self.rec_max_radius = lig_max_radius, rec_max_radius
        self.cross_max_distance, self.center_max_distance = cross_max_distance, center_max_distance
        self.distance_embed_dim, self.cross_distance_embed_dim = distance_embed_dim, cross_distance_embed_dim
        self.dynamic_max_cross = dynamic_max_cross
        self.ns, self.nv = ns, nv
        self.scale_by_sigma, self.no_torsion = scale_by_sigma, no_torsion
        self.confidence_mode = False
        self.num_conv_layers, self.num_prot_emb_layers = num_conv_layers, num_prot_emb_layers
        self.fixed_center_conv = fixed_center_conv
        self.reduce_pseudoscalars = reduce_pseudoscalars
        self.lm_embedding_type = lm_embedding_type
        self.embedding_scale = embedding_scale
        lm_dim = 1280 if lm_embedding_type == "precomputed" else 0
        sh = "1x0e + 1x1o"

        self.lig_node_embedding = AtomEncoder(ns, LIG_FEATURE_DIMS, sigma_embed_dim)
        self.lig_edge_embedding = nn.Sequential(nn.Linear(in_lig_edge_features + sigma_embed_dim + distance_embed_dim, ns),
                                                nn.ReLU(), nn.Dropout(dropout), nn.Linear(ns, ns))
        self.rec_node_embedding = AtomEncoder(ns, REC_RESIDUE_FEATURE_DIMS, 0, lm_embedding_dim=lm_dim)
        self.rec_edge_embedding = nn.Sequential(nn.Linear(distance_embed_dim, ns), nn.ReLU(), nn.Dropout(dropout), nn.Linear(ns, ns))
        self.rec_sigma_embedding = nn.Sequential(nn.Linear(sigma_embed_dim, ns), nn.ReLU(), nn.Dropout(dropout), nn.Linear(ns, ns))
        self.cross_edge_embedding = nn.Sequential(nn.Linear(sigma_embed_dim + cross_distance_embed_dim, ns), nn.ReLU(),
                                                  nn.Dropout(dropout), nn.Linear(ns, ns))
        self.lig_distance_expansion = GaussianSmearing(0.0, lig_max_radius, distance_embed_dim)
        self.rec_distance_expansion = GaussianSmearing(0.0, rec_max_radius, distance_embed_dim)
        self.cross_distance_expansion = GaussianSmearing(0.0, cross_max_distance, cross_distance_embed_dim)

        seq = get_irrep_seq(ns, nv, use_second_order_repr, reduce_pseudoscalars)

        def conv(i, groups):
            a, b = seq[min(i, len(seq) - 1)], seq[min(i + 1, len(seq) - 1)]
            return TensorProductConvLayer(a, sh, b, 3 * ns, hidden_features=3 * ns, residual=True, batch_norm=batch_norm,
                                          dropout=dropout, weight_numel=faster_tp_weight_numel(a, b), edge_groups=groups)

        self.rec_emb_layers = nn.ModuleList([conv(i, 1) for i in range(num_prot_emb_layers)])
        self.embed_also_ligand = embed_also_ligand
        self.lig_emb_layers = nn.ModuleList([conv(i, 1) for i in range(num_prot_emb_layers)])
        last = num_prot_emb_layers + num_conv_layers - 1
        self.conv_layers = nn.ModuleList([conv(i, 2 if i == last else 4)
                                          for i in range(num_prot_emb_layers, num_prot_emb_layers + num_conv_layers)])

        self.center_distance_expansion = GaussianSmearing(0.0, center_max_distance, distance_embed_dim)
        self.center_edge_embedding = nn.Sequential(nn.Linear(distance_embed_dim + sigma_embed_dim, ns), nn.ReLU(),
                                                   nn.Dropout(dropout), nn.Linear(ns, ns))
        # e3nn FullyConnectedTensorProduct weight counts for these irreps: 124 and 384
        self.final_conv = TensorProductConvLayer(self.conv_layers[-1].out_irreps, sh, "2x1o + 2x1e", 2 * ns, residual=False,
                                                 dropout=dropout, batch_norm=batch_norm,
                                                 weight_numel=ns * 2 + 5 * nv * 2)
        self.tr_final_layer = nn.Sequential(nn.Linear(1 + sigma_embed_dim, ns), nn.Dropout(dropout), nn.ReLU(), nn.Linear(ns, 1))
        self.rot_final_layer = nn.Sequential(nn.Linear(1 + sigma_embed_dim, ns), nn.Dropout(dropout), nn.ReLU(), nn.Linear(ns, 1))
        if not no_torsion:
            self.final_edge_embedding = nn.Sequential(nn.Linear(distance_embed_dim, ns), nn.ReLU(), nn.Dropout(dropout), nn.Linear(ns, ns))
            self.tor_bond_conv = TensorProductConvLayer(self.conv_layers[-1].out_irreps, "1x1o + 1x2o + 1x2e + 1x3o",
                                                        f"{ns}x0o + {ns}x0e", 3 * ns, residual=False, dropout=dropout,
                                                        batch_norm=batch_norm, weight_numel=2 * nv * ns)
            self.tor_final_layer = nn.Sequential(nn.Linear(2 * ns, ns, bias=False), nn.Tanh(), nn.Dropout(dropout),
                                                 nn.Linear(ns, 1, bias=False))
        self._engine = None
        self._engine_key = None

    # ------------------------------------------------------------------ checkpoint compatibility
    _IGNORED_PREFIXES = ("final_conv.tp.", "tor_bond_conv.tp.", "final_tp_tor.")

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Persistent buffers real e3nn modules add (output_mask, _w3j_*) are not parameters of this model: their arithmetic
        is hard-wired in the engine.  The Wigner-3j constants among them are authoritative for the checkpoint:
        they are compared with the baked ones and a mismatch raises (e3nn_constants.check_w3j_buffers); the rest is dropped."""
        from .e3nn_constants import check_w3j_buffers
        check_w3j_buffers(state_dict)
        sd = {k: v for k, v in state_dict.items() if not k.startswith(self._IGNORED_PREFIXES)}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._engine_key = None  # weights changed -> re-upload
        return out

    # ------------------------------------------------------------------ engine plumbing
    def engine(self):
        """The per-device DockEngine holding this model's weights (created lazily, refreshed after
        load_state_dict / .to())."""
        from .engine import DockEngine
        if self.training:
            raise RuntimeError("the MI355X engine implements the eval-mode forward pass only; call model.eval()")
        dev = next(self.parameters()).device
        key = (str(dev), self._weights_version())
        if self._engine is None or self._engine_key != key:
            self._engine = DockEngine.from_model(self, dev)
            self._engine_key = key
        return self._engine

    def co_engines(self, n: int, main):
        """`n` further engines on the device of `main` that use ITS device-resident weights (cbd_share_weights) and hold their own
        complex / workspace: the partners of `main` in cbd_sample_multi (sampling(co_schedule=...))."""
        from .engine import DockEngine
        if getattr(self, "_co_main", None) is not main:
            self._co, self._co_main = [], main
        while len(self._co) < n:
            e = DockEngine(main.device, max_batch=main.max_batch, lm_embedding_dim=main.cfg.lm_embedding_dim,
                           no_torsion=bool(main.cfg.no_torsion), lig_max_radius=main.cfg.lig_max_radius,
                           rec_max_radius=main.cfg.rec_max_radius, cross_max_distance=main.cfg.cross_max_distance,
                           center_max_distance=main.cfg.center_max_distance)
            e.share_weights_from(main)
            self._co.append(e)
        return self._co[:n]

    def engine_pool(self, n_streams: int = 1, max_batch: int = 64):
        """n-stream engine pool used by sampling() (see engine.DockEnginePool)."""
        from .engine import DockEnginePool
        if self.training:
            raise RuntimeError("the MI355X engine implements the eval-mode forward pass only; call model.eval()")
        dev = next(self.parameters()).device
        key = (str(dev), self._weights_version(), n_streams, max_batch)
        if getattr(self, "_pool", None) is None or self._pool_key != key:
            self._pool = DockEnginePool.from_model(self, dev, n=n_streams, max_batch=max_batch)
            self._pool_key = key
        return self._pool

    def _weights_version(self):
        return weights_version(self)

    def invalidate_engine(self):
        """Force the next engine() / engine_pool() call to re-upload the weights (after out-of-band updates of parameters or
        BatchNorm buffers that autograd's version counters do not see)."""
        self._engine_key = None
        self._pool_key = None

    def train(self, mode: bool = True):
        if self.training and not mode:
            self.invalidate_engine()      # leaving training mode: parameters and running statistics have (very likely) moved
        return super().train(mode)

    def forward(self, data):
        """Same contract as the reference forward (models/score_model.py:333-449):
        returns (tr_pred [B,3], rot_pred [B,3], tor_pred [B*R], None).  eval mode: the fused inference engine (no autograd);
        training mode: the differentiable path of train_forward.py (HIP tensor-product op + autograd) for fine-tuning."""
        if self.training:
            return self.forward_train(data)
        from .engine import score_batch
        return score_batch(self, data)

    def forward_train(self, data):
        """Differentiable forward (list of HeteroData or a collated Batch with per-graph complex_t), any mode."""
        from .train_forward import forward as _fwd
        return _fwd(self, data)
