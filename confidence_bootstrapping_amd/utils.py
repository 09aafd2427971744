"""`get_model` with the reference's signature and yml -> constructor mapping (reference
utils/utils.py:175-288), plus helpers to read `model_parameters.yml` the way inference.py:246-267 does."""
from __future__ import annotations

import os
from argparse import Namespace
from functools import partial

import torch
import yaml

from .diffusion_utils import get_timestep_embedding, t_to_sigma as t_to_sigma_compl
from .score_model import TensorProductScoreModel

_DEFAULT_YML = os.path.join(os.path.dirname(__file__), "data", "pretrained_score_model_parameters.yml")
_CONFIDENCE_YML = os.path.join(os.path.dirname(__file__), "data", "pretrained_confidence_model_parameters.yml")


def load_model_args(path: str = None) -> Namespace:
    """yaml.full_load -> Namespace with the back-compat defaults inference.py:246-267 applies."""
    with open(path or _DEFAULT_YML) as f:
        args = Namespace(**yaml.full_load(f))
    for k, v in (("not_fixed_knn_radius_graph", True), ("not_knn_only_graph", True), ("include_miscellaneous_atoms", False),
                 ("triple_training", False), ("train_multiplicity", 1), ("old_score_model", False)):
        if not hasattr(args, k):
            setattr(args, k, v)
    return args


def _has(args, k):
    return k in args and getattr(args, k) is not None


def get_model(args, device, t_to_sigma, no_parallel=False, confidence_mode=False, old=False):
    if old:
        raise NotImplementedError("old score-model variants are outside the MI355X hot path")
    all_atoms = "all_atoms" in args and args.all_atoms
    if all_atoms != bool(confidence_mode):
        raise NotImplementedError("the MI355X build covers the coarse-grained score model and the all-atom confidence model")
    emb_type = args.embedding_type if "embedding_type" in args else "sinusoidal"
    emb_scale = args.embedding_scale if "embedding_type" in args else 10000
    timestep_emb_func = get_timestep_embedding(emb_type, args.sigma_embed_dim, emb_scale)
    lm = None
    if any(_has(args, k) for k in ("moad_esm_embeddings_path", "pdbbind_esm_embeddings_path",
                                   "pdbsidechain_esm_embeddings_path", "esm_embeddings_path")):
        lm = "precomputed"
    if _has(args, "esm_embeddings_model"):
        lm = args.esm_embeddings_model
    g = lambda k, d: getattr(args, k) if k in args else d
    if all_atoms:
        from .all_atom_score_model import TensorProductScoreModel as AAScoreModel
        n_out = lambda k: len(getattr(args, k)) + 1 if k in args and isinstance(getattr(args, k), list) else 1
        model = AAScoreModel(
            t_to_sigma=t_to_sigma, device=device, no_torsion=args.no_torsion, timestep_emb_func=timestep_emb_func,
            num_conv_layers=args.num_conv_layers, lig_max_radius=args.max_radius, scale_by_sigma=args.scale_by_sigma,
            sigma_embed_dim=args.sigma_embed_dim, norm_by_sigma=g("norm_by_sigma", False), ns=args.ns, nv=args.nv,
            distance_embed_dim=args.distance_embed_dim, cross_distance_embed_dim=args.cross_distance_embed_dim,
            batch_norm=not args.no_batch_norm, dropout=args.dropout, use_second_order_repr=args.use_second_order_repr,
            cross_max_distance=args.cross_max_distance, dynamic_max_cross=args.dynamic_max_cross,
            separate_noise_schedule=args.separate_noise_schedule, smooth_edges=g("smooth_edges", False),
            odd_parity=g("odd_parity", False), lm_embedding_type=lm, confidence_mode=confidence_mode,
            asyncronous_noise_schedule=g("asyncronous_noise_schedule", False),
            affinity_prediction=g("affinity_prediction", False), parallel=g("parallel", 1),
            num_confidence_outputs=n_out("rmsd_classification_cutoff"),
            atom_num_confidence_outputs=n_out("atom_rmsd_classification_cutoff"),
            parallel_aggregators=g("parallel_aggregators", ""),
            fixed_center_conv=(not args.not_fixed_center_conv) if "not_fixed_center_conv" in args else False,
            no_aminoacid_identities=g("no_aminoacid_identities", False),
            include_miscellaneous_atoms=g("include_miscellaneous_atoms", False), sh_lmax=g("sh_lmax", 2),
            differentiate_convolutions=(not args.no_differentiate_convolutions) if "no_differentiate_convolutions" in args else True,
            tp_weights_layers=g("tp_weights_layers", 2), num_prot_emb_layers=g("num_prot_emb_layers", 0),
            reduce_pseudoscalars=g("reduce_pseudoscalars", False), embed_also_ligand=g("embed_also_ligand", False),
            atom_confidence=(args.atom_confidence_loss_weight > 0.0) if "atom_confidence_loss_weight" in args else False,
            sidechain_pred=(g("sidechain_loss_weight", 0) > 0) or (g("backbone_loss_weight", 0) > 0),
            depthwise_convolution=g("depthwise_convolution", False), embedding_scale=emb_scale)
        model.to(device)
        return model
    model = TensorProductScoreModel(
        t_to_sigma=t_to_sigma, device=device, no_torsion=args.no_torsion, timestep_emb_func=timestep_emb_func,
        num_conv_layers=args.num_conv_layers, lig_max_radius=args.max_radius, scale_by_sigma=args.scale_by_sigma,
        sigma_embed_dim=args.sigma_embed_dim, norm_by_sigma=g("norm_by_sigma", False), ns=args.ns, nv=args.nv,
        distance_embed_dim=args.distance_embed_dim, cross_distance_embed_dim=args.cross_distance_embed_dim,
        batch_norm=not args.no_batch_norm, dropout=args.dropout, use_second_order_repr=args.use_second_order_repr,
        cross_max_distance=args.cross_max_distance, dynamic_max_cross=args.dynamic_max_cross,
        separate_noise_schedule=args.separate_noise_schedule, smooth_edges=g("smooth_edges", False),
        odd_parity=g("odd_parity", False), lm_embedding_type=lm, confidence_mode=confidence_mode,
        asyncronous_noise_schedule=g("asyncronous_noise_schedule", False),
        fixed_center_conv=(not args.not_fixed_center_conv) if "not_fixed_center_conv" in args else False,
        no_aminoacid_identities=g("no_aminoacid_identities", False),
        include_miscellaneous_atoms=g("include_miscellaneous_atoms", False), sh_lmax=g("sh_lmax", 2),
        differentiate_convolutions=(not args.no_differentiate_convolutions) if "no_differentiate_convolutions" in args else True,
        tp_weights_layers=g("tp_weights_layers", 2), num_prot_emb_layers=g("num_prot_emb_layers", 0),
        reduce_pseudoscalars=g("reduce_pseudoscalars", False), embed_also_ligand=g("embed_also_ligand", False),
        sidechain_pred=(g("sidechain_loss_weight", 0) > 0) or (g("backbone_loss_weight", 0) > 0),
        depthwise_convolution=g("depthwise_convolution", False), embedding_scale=emb_scale)
    # the reference wraps in torch_geometric DataParallel unless no_parallel; the MI355X build is one process
    # per GPU (distributed.py), so the module is returned bare either way.
    model.to(device)
    return model


def make_confidence_model(device="cpu", seed=5, args=None, eval_mode=True):
    """Random-init all-atom confidence model of the shipped architecture (see make_score_model)."""
    return make_score_model(device, seed, args or load_model_args(_CONFIDENCE_YML), eval_mode, confidence_mode=True)


def make_score_model(device="cpu", seed=0, args=None, eval_mode=True, confidence_mode=False):
    """Random-init score model of the shipped architecture with non-trivial BatchNorm statistics
    (checkpoints are not available offline)."""
    args = args or load_model_args()
    gen = torch.random.get_rng_state()
    torch.manual_seed(seed)
    try:
        model = get_model(args, torch.device("cpu"), partial(t_to_sigma_compl, args=args), no_parallel=True,
                          confidence_mode=confidence_mode)
        g = torch.Generator().manual_seed(seed + 1)
        for name, buf in model.named_buffers():
            if name.endswith("running_var"):
                buf.copy_(torch.rand(buf.shape, generator=g) + 0.5)
            elif name.endswith("running_mean"):
                buf.copy_(torch.randn(buf.shape, generator=g) * 0.1)
        for name, p in model.named_parameters():
            if "batch_norm" in name or "confidence_predictor.1." in name or "confidence_predictor.5." in name:
                with torch.no_grad():
                    p.copy_(p + 0.1 * torch.randn(p.shape, generator=g))
    finally:
        torch.random.set_rng_state(gen)
    if eval_mode:
        model.eval()
    return model.to(device), args



def unfreeze_layer(module):
    """requires_grad = True for every parameter below `module` (reference utils/utils.py, used by the layer-wise warm-up)."""
    for p in module.parameters():
        p.requires_grad = True


# parameter groups the layer-wise warm-up releases (reference utils/utils.py:143-153): heads first, then one interaction layer per stage
# from the last to the first, then the embeddings
_WARMUP_HEADS = ("center_edge_embedding", "final_conv", "tr_final_layer", "rot_final_layer", "final_edge_embedding", "final_tp_tor",
                 "tor_bond_conv", "tor_final_layer")
_WARMUP_EMBEDDINGS = ("lig_node_embedding", "lig_edge_embedding", "rec_node_embedding", "rec_edge_embedding", "rec_sigma_embedding",
                      "cross_edge_embedding", "rec_emb_layers", "lig_emb_layers")


def get_optimizer_and_scheduler(args, model, scheduler_mode="min", step=0, optimizer=None):
    """Adam + learning-rate scheduler of the fine-tuning loop, with the reference's signature and stages (utils/utils.py:134-172;
    called by finetune_train.py:246-263): `args.scheduler` in {'plateau', 'linear_warmup', 'layer_linear_warmup', other -> None}.
    'layer_linear_warmup' freezes everything but the batch norms at stage 0, then releases the heads, one interaction layer per stage
    and finally the embeddings, building a new optimizer over the trainable parameters at every stage.
    On a GPU the optimizer is PyTorch's fused Adam (one multi-tensor kernel per step instead of ~20 foreach kernels: same update)."""
    layerwise = args.scheduler == "layer_linear_warmup"
    if layerwise:
        if step == 0:
            for name, child in model.named_children():
                if "batch_norm" in name:
                    continue
                for pname, p in child.named_parameters():
                    if "batch_norm" not in pname:
                        p.requires_grad = False
            for name in _WARMUP_HEADS:
                if hasattr(model, name):          # final_tp_tor holds no parameters here (closed-form kernel)
                    unfreeze_layer(getattr(model, name))
        elif 0 < step <= args.num_conv_layers:
            unfreeze_layer(model.conv_layers[-step])
        elif step == args.num_conv_layers + 1:
            for name in _WARMUP_EMBEDDINGS:
                unfreeze_layer(getattr(model, name))
    if step == 0 or layerwise:
        params = [p for p in model.parameters() if p.requires_grad]
        fused = bool(params) and all(p.is_cuda and torch.is_floating_point(p) for p in params)
        optimizer = torch.optim.Adam(params, lr=args.lr, weight_decay=args.w_decay, **({"fused": True} if fused else {}))
    plateau = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, mode=scheduler_mode, factor=0.7, patience=args.scheduler_patience,
                                                         min_lr=args.lr / 100)
    if args.scheduler == "plateau":
        scheduler = plateau
    elif args.scheduler in ("linear_warmup", "layer_linear_warmup"):
        warming = step < 1 if args.scheduler == "linear_warmup" else step <= args.num_conv_layers + 1
        scheduler = torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=args.lr_start_factor, end_factor=1.0,
                                                      total_iters=args.warmup_dur) if warming else plateau
    else:
        print("No scheduler")
        scheduler = None
    return optimizer, scheduler


class ExponentialMovingAverage:
    """Exponential moving average of the trainable parameters with the interface finetune_train.py uses (`update`, `copy_to`,
    `store`, `restore`, `state_dict`, `load_state_dict`; reference utils/utils.py:306-392): with `use_num_updates` the decay warms
    up as min(decay, (1 + n) / (10 + n)).  All parameters are updated by one fused multi-tensor lerp on the device."""

    def __init__(self, parameters, decay, use_num_updates=True):
        if not 0.0 <= decay <= 1.0:
            raise ValueError("Decay must be between 0 and 1")
        self.decay = decay
        self.num_updates = 0 if use_num_updates else None
        self.shadow_params = [p.detach().clone() for p in parameters if p.requires_grad]
        self.collected_params = []

    def update(self, parameters):
        decay = self.decay
        if self.num_updates is not None:
            self.num_updates += 1
            decay = min(decay, (1 + self.num_updates) / (10 + self.num_updates))
        live = [p.detach() for p in parameters if p.requires_grad]
        with torch.no_grad():
            torch._foreach_lerp_(self.shadow_params, live, 1.0 - decay)   # s += (1 - decay) * (p - s)

    def copy_to(self, parameters):
        with torch.no_grad():
            for s, p in zip(self.shadow_params, [p for p in parameters if p.requires_grad]):
                p.copy_(s)

    def store(self, parameters):
        self.collected_params = [p.detach().clone() for p in parameters]

    def restore(self, parameters):
        with torch.no_grad():
            for c, p in zip(self.collected_params, parameters):
                p.copy_(c)

    def state_dict(self):
        return dict(decay=self.decay, num_updates=self.num_updates, shadow_params=self.shadow_params)

    def load_state_dict(self, state_dict, device):
        self.decay = state_dict["decay"]
        self.num_updates = state_dict["num_updates"]
        self.shadow_params = [t.to(device) for t in state_dict["shadow_params"]]


def subgraph_mask(keep: torch.Tensor, edge_index: torch.Tensor):
    """torch_geometric.utils.subgraph(keep, edge_index, relabel_nodes=True)[0] for a boolean node mask: the edges whose both ends are
    kept, in their original order, re-numbered by rank among the kept nodes."""
    m = keep[edge_index[0]] & keep[edge_index[1]]
    return (torch.cumsum(keep.long(), 0) - 1)[edge_index[:, m]]


def crop_beyond(complex_graph, cutoff, all_atoms):
    """Reference utils/utils.py:395-420, same in-place semantics: residues without a ligand atom closer than `cutoff` are dropped from
    the receptor stores (x, pos, side_chain_vecs), the C-alpha graph keeps the edges between kept residues (re-numbered), and with
    `all_atoms` the atoms of dropped residues go as well (atom graph restricted, atom -> residue map re-numbered)."""
    lig_pos, rec_pos = complex_graph["ligand"].pos, complex_graph["receptor"].pos
    keep = torch.any(torch.sum((lig_pos.unsqueeze(0) - rec_pos.unsqueeze(1)) ** 2, -1) < cutoff ** 2, dim=1)
    rec = complex_graph["receptor"]
    if all_atoms:
        a2r = complex_graph["atom", "atom_rec_contact", "receptor"].edge_index[1]
        atoms_keep = keep[a2r]
        new_map = (torch.cumsum(keep.long(), dim=0) - 1)[a2r][atoms_keep]
    rec.pos, rec.x = rec.pos[keep], rec.x[keep]
    if "side_chain_vecs" in rec:
        rec.side_chain_vecs = rec.side_chain_vecs[keep]
    rr = complex_graph["receptor", "rec_contact", "receptor"]
    rr.edge_index = subgraph_mask(keep, rr.edge_index)
    if all_atoms:
        at = complex_graph["atom"]
        at.x, at.pos = at.x[atoms_keep], at.pos[atoms_keep]
        aa = complex_graph["atom", "atom_contact", "atom"]
        aa.edge_index = subgraph_mask(atoms_keep, aa.edge_index)
        complex_graph["atom", "atom_rec_contact", "receptor"].edge_index = torch.stack([torch.arange(len(new_map), device=new_map.device), new_map])
    return keep
