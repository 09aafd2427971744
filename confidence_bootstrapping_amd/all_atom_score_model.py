"""All-atom `TensorProductScoreModel` in CONFIDENCE mode with the reference's constructor signature and `state_dict`
layout (reference models/all_atom_score_model.py:22-282), whose forward pass runs on the MI355X engine.

Like score_model.TensorProductScoreModel this is a parameter container + engine handle: it owns the parameters under
the checkpoint's key names (264 entries, 3 883 676 parameters for the shipped
workdir/pretrained_confidence/model_parameters.yml) so `load_state_dict(..., strict=True)` of
`best_ema_inference_epoch_model.pt`-style checkpoints works, and hands them to the HIP engine.  There is NO PyTorch
fallback: `forward` raises if the HIP library is missing.

Supported architecture = the shipped confidence yml: confidence_mode, all atoms, sh_lmax=2 (e3nn
FullyConnectedTensorProduct layers), ns=24, nv=6, 5 interaction layers (9 edge groups, 3 in the last), no embedding
layers, atom_confidence head, BatchNorm everywhere, eval mode, lm_embedding_type in {None,'precomputed'}.
"""
from __future__ import annotations

import torch
from torch import nn

from .score_model import (AtomEncoder, GaussianSmearing, TensorProductConvLayer, LIG_FEATURE_DIMS, REC_RESIDUE_FEATURE_DIMS,
                          get_irrep_seq, parse_irreps)

REC_ATOM_FEATURE_DIMS = ([38, 119, 23, 38], 0)   # datasets/process_mols.py:114-119


def fctp_weight_numel(in_irreps: str, sh_irreps: str, out_irreps: str) -> int:
    """weight_numel of o3.FullyConnectedTensorProduct(in, sh, out): sum over (in, sh, out) triples allowed by the
    selection rules |l1-l2| <= l3 <= l1+l2, p3 = p1 p2 of mul_in * mul_sh * mul_out."""
    n = 0
    for m1, l1, p1 in parse_irreps(in_irreps):
        for m2, l2, p2 in parse_irreps(sh_irreps):
            for m3, l3, p3 in parse_irreps(out_irreps):
                if abs(l1 - l2) <= l3 <= l1 + l2 and p3 == p1 * p2:
                    n += m1 * m2 * m3
    return n


def _confidence_head(in_dim, ns, out_dim, dropout):
    return nn.Sequential(nn.Linear(in_dim, ns), nn.BatchNorm1d(ns), nn.ReLU(), nn.Dropout(dropout),
                         nn.Linear(ns, ns), nn.BatchNorm1d(ns), nn.ReLU(), nn.Dropout(dropout), nn.Linear(ns, out_dim))


class TensorProductScoreModel(nn.Module):
    def __init__(self, t_to_sigma, device, timestep_emb_func, in_lig_edge_features=4, sigma_embed_dim=32, sh_lmax=2,
                 ns=16, nv=4, num_conv_layers=2, lig_max_radius=5, rec_max_radius=30, cross_max_distance=250,
                 center_max_distance=30, distance_embed_dim=32, cross_distance_embed_dim=32, no_torsion=False,
                 scale_by_sigma=True, norm_by_sigma=True, use_second_order_repr=False, batch_norm=True,
                 dynamic_max_cross=False, dropout=0.0, smooth_edges=False, odd_parity=False,
                 separate_noise_schedule=False, lm_embedding_type=False, confidence_mode=False,
                 confidence_dropout=0, confidence_no_batchnorm=False,
                 asyncronous_noise_schedule=False, affinity_prediction=False, parallel=1,
                 parallel_aggregators="mean max min std", num_confidence_outputs=1, atom_num_confidence_outputs=1,
                 fixed_center_conv=False, no_aminoacid_identities=False, include_miscellaneous_atoms=False,
                 differentiate_convolutions=True, tp_weights_layers=2, num_prot_emb_layers=0,
                 reduce_pseudoscalars=False, embed_also_ligand=False, atom_confidence=False, sidechain_pred=False,
                 depthwise_convolution=False, crop_beyond=None, embedding_scale=None):
        super().__init__()
        unsupported = {
            "not confidence_mode": not confidence_mode, "sh_lmax != 2": sh_lmax != 2,
            "use_second_order_repr": use_second_order_repr, "separate_noise_schedule": separate_noise_schedule,
            "asyncronous_noise_schedule": asyncronous_noise_schedule, "smooth_edges": smooth_edges, "odd_parity": odd_parity,
            "include_miscellaneous_atoms": include_miscellaneous_atoms, "sidechain_pred": sidechain_pred,
            "depthwise_convolution": depthwise_convolution, "not differentiate_convolutions": not differentiate_convolutions,
            "tp_weights_layers != 2": tp_weights_layers != 2, "embed_also_ligand": embed_also_ligand,
            "num_prot_emb_layers != 0": num_prot_emb_layers != 0, "not batch_norm": not batch_norm,
            "reduce_pseudoscalars": reduce_pseudoscalars, "not dynamic_max_cross": not dynamic_max_cross,
            "no_aminoacid_identities": no_aminoacid_identities, "parallel != 1": parallel != 1,
            "affinity_prediction": affinity_prediction, "confidence_no_batchnorm": confidence_no_batchnorm,
            "not atom_confidence": not atom_confidence, "confidence outputs != 1": (num_confidence_outputs, atom_num_confidence_outputs) != (1, 1),
            "ns != 24 or nv != 6": (ns, nv) != (24, 6), "num_conv_layers != 5": num_conv_layers != 5,
            "lm_embedding_type": lm_embedding_type not in (None, "precomputed"),
            "embed dims != 32": (sigma_embed_dim, distance_embed_dim, cross_distance_embed_dim) != (32, 32, 32),
        }
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError("MI355X engine covers the shipped pretrained_confidence architecture only; "
                                      "unsupported options: " + ", ".join(bad))
        self.t_to_sigma, self.device, self.timestep_emb_func = t_to_sigma, device, timestep_emb_func
        self.in_lig_edge_features, self.sigma_embed_dim = in_lig_edge_features, sigma_embed_dim
        self.lig_max_radius, self.rec_max_radius, self.cross_max_distance = lig_max_radius, rec_max_radius, cross_max_distance
        self.dynamic_max_cross = dynamic_max_cross
        self.distance_embed_dim, self.cross_distance_embed_dim = distance_embed_dim, cross_distance_embed_dim
        self.ns, self.nv = ns, nv
        self.confidence_mode, self.atom_confidence = True, True
        self.num_conv_layers, self.num_prot_emb_layers = num_conv_layers, 0
        self.lm_embedding_type = lm_embedding_type
        self.crop_beyond = crop_beyond
        self.embedding_scale = embedding_scale
        lm_dim = 1280 if lm_embedding_type == "precomputed" else 0
        sh = "1x0e + 1x1o + 1x2e"

        def edge_mlp(in_dim):
            return nn.Sequential(nn.Linear(in_dim, ns), nn.ReLU(), nn.Dropout(dropout), nn.Linear(ns, ns))

        self.lig_node_embedding = AtomEncoder(ns, LIG_FEATURE_DIMS, sigma_embed_dim)
        self.lig_edge_embedding = edge_mlp(in_lig_edge_features + sigma_embed_dim + distance_embed_dim)
        self.rec_sigma_embedding = edge_mlp(sigma_embed_dim)
        self.rec_node_embedding = AtomEncoder(ns, REC_RESIDUE_FEATURE_DIMS, 0, lm_embedding_dim=lm_dim)
        self.rec_edge_embedding = edge_mlp(distance_embed_dim)
        self.atom_node_embedding = AtomEncoder(ns, REC_ATOM_FEATURE_DIMS, 0)
        self.atom_edge_embedding = edge_mlp(distance_embed_dim)
        self.lr_edge_embedding = edge_mlp(sigma_embed_dim + cross_distance_embed_dim)
        self.ar_edge_embedding = edge_mlp(distance_embed_dim)
        self.la_edge_embedding = edge_mlp(sigma_embed_dim + cross_distance_embed_dim)
        self.lig_distance_expansion = GaussianSmearing(0.0, lig_max_radius, distance_embed_dim)
        self.rec_distance_expansion = GaussianSmearing(0.0, rec_max_radius, distance_embed_dim)
        self.cross_distance_expansion = GaussianSmearing(0.0, cross_max_distance, cross_distance_embed_dim)

        seq = get_irrep_seq(ns, nv, use_second_order_repr, reduce_pseudoscalars)
        self.rec_emb_layers = nn.ModuleList()
        self.embed_also_ligand = False
        layers = []
        for i in range(num_conv_layers):
            a, b = seq[min(i, len(seq) - 1)], seq[min(i + 1, len(seq) - 1)]
            layers.append(TensorProductConvLayer(a, sh, b, 3 * ns, hidden_features=3 * ns, residual=True, batch_norm=batch_norm,
                                                 dropout=dropout, weight_numel=fctp_weight_numel(a, sh, b),
                                                 edge_groups=3 if i == num_conv_layers - 1 else 9))
        self.conv_layers = nn.ModuleList(layers)
        self.atom_confidence_predictor = _confidence_head(2 * ns, ns, atom_num_confidence_outputs + ns, confidence_dropout)
        self.confidence_predictor = _confidence_head(ns, ns, num_confidence_outputs, confidence_dropout)
        self._engine = None
        self._engine_key = None

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Persistent buffers real e3nn modules add under `conv_layers.N.tp.` are dropped after their
        Wigner-3j constants have been checked against the ones hard-wired in fctp_conv.hip (a mismatch raises)."""
        from .e3nn_constants import check_w3j_buffers
        check_w3j_buffers(state_dict)
        sd = {k: v for k, v in state_dict.items() if ".tp." not in k}
        out = super().load_state_dict(sd, strict=strict, **kw)
        self._engine_key = None
        return out

    def _weights_version(self):
        from .score_model import weights_version
        return weights_version(self)

    def invalidate_engine(self):
        self._engine_key = None

    def engine(self, max_batch: int = 64):
        from .engine import ConfidenceEngine
        if self.training:
            raise RuntimeError("the MI355X engine implements the eval-mode forward pass only; call model.eval()")
        dev = next(self.parameters()).device
        key = (str(dev), self._weights_version(), max_batch)
        if self._engine is None or self._engine_key != key:
            self._engine = ConfidenceEngine.from_model(self, dev, max_batch=max_batch)
            self._engine_key = key
        return self._engine

    def co_engines(self, n: int, main):
        """`n` further confidence engines on the device of `main` (own weights copy, own complex / workspace): the partners of `main`
        in cbd_conf_score_multi -- sampling() scores the final poses of a co-scheduled group of complexes in one set of launches."""
        from .engine import ConfidenceEngine
        if getattr(self, "_co_main", None) is not main:
            self._co, self._co_main = [], main
        while len(self._co) < n:
            self._co.append(ConfidenceEngine.from_model(self, main.device, max_batch=main.max_batch))
        return self._co[:n]

    def forward(self, data):
        """Same contract as the reference forward in confidence mode (models/all_atom_score_model.py:363-454):
        returns (confidence [B], atom_confidence [B*Nl, 1]).  `data` is a Batch of poses of ONE complex carrying the
        all-atom stores; cropping to `crop_beyond` (if set on the model args) is applied by the caller in the reference
        (utils/sampling.py:245-250) and inside the engine here -- pass the un-cropped complex."""
        from .engine import confidence_batch
        return confidence_batch(self, data)
