"""Import machinery that lets the REFERENCE's own Python modules run in this container.

TEST INFRASTRUCTURE ONLY, and only usable where /root/reference exists (never on the GPU box).
Used by oracle/make_golden.py to generate tests/golden/*.npz.

What is real and what is a shim when reference code executes under this loader:
  real  : models/tensor_layers.py (FasterTensorProduct, TensorProductConvLayer, get_irrep_seq),
          models/layers.py (FCBlock), models/score_model.py (TensorProductScoreModel wiring, AtomEncoder,
          GaussianSmearing), utils/sampling.py (sampling, randomize_position), utils/diffusion_utils.py,
          utils/torsion.py, utils/geometry.py, utils/so3.py, utils/torus.py (table look-ups on cached tables)
  shims : e3nn (oracle/e3nn_ref.py), torch_scatter / torch_cluster (oracle/graph_ref.py),
          torch_geometric DataLoader/Batch (confidence_bootstrapping_amd/hetero.py containers),
          rdkit / Bio / prody / esm / wandb (MagicMock; never executed on the hot path)
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types
from unittest.mock import MagicMock

REF = "/root/reference"
_MOCK_ROOTS = ("rdkit", "Bio", "prody", "esm", "wandb", "torch_geometric", "openbabel", "MDAnalysis", "sklearn_extra")


class _MockLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = MagicMock(name=spec.name)
        m.__path__ = []
        m.__name__ = spec.name
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


class _MockFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in _MOCK_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, _MockLoader(), is_package=True)
        return None


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install(scratch_dir="/tmp/cb_tables", torus_seed=0, load_tables=True):
    """Prepare sys.modules so `import models.score_model`, `import utils.sampling` ... resolve to the
    reference.  Must run before anything imports HuggingFace `datasets`."""
    if not os.path.isdir(REF):
        raise RuntimeError("/root/reference is not available here")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import e3nn_ref as e3, graph_ref as gr
    from confidence_bootstrapping_amd import hetero

    if not any(isinstance(f, _MockFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _MockFinder())
    # namespace packages of the reference, pinned ahead of same-named site-packages
    for pkg in ("datasets", "utils", "models", "confidence", "bootstrapping", "spyrmsd"):
        for k in [k for k in sys.modules if k == pkg or k.startswith(pkg + ".")]:
            del sys.modules[k]
        m = types.ModuleType(pkg)
        m.__path__ = [os.path.join(REF, pkg)]
        sys.modules[pkg] = m

    # ---- functional shims
    o3 = _module("e3nn.o3", Irreps=e3.Irreps, Irrep=e3.Irrep, spherical_harmonics=e3.spherical_harmonics,
                 FullyConnectedTensorProduct=e3.FullyConnectedTensorProduct, FullTensorProduct=e3.FullTensorProduct,
                 Linear=MagicMock(name="o3.Linear"), TensorProduct=MagicMock(name="o3.TensorProduct"))
    nn_ = _module("e3nn.nn", BatchNorm=e3.BatchNorm)
    e3nn = _module("e3nn", o3=o3, nn=nn_)
    e3nn.__path__ = []
    _module("torch_scatter", scatter=gr.scatter, scatter_mean=gr.scatter_mean)
    _module("torch_cluster", radius=gr.radius, radius_graph=gr.radius_graph, knn_graph=MagicMock())
    # base classes the reference subclasses at import time (a MagicMock base would turn the subclass into a mock)
    _module("torch_geometric.transforms", BaseTransform=object)
    _module("torch_geometric.data", Dataset=object, HeteroData=hetero.HeteroData, Batch=hetero.Batch, Data=object)

    # ---- score-normaliser tables: import the reference modules on their cached .npy files
    if load_tables:
        import numpy as np
        cwd = os.getcwd()
        os.chdir(scratch_dir)
        try:
            import utils.so3  # noqa: F401  (loads .so3_*4.npy caches)
            np.random.seed(torus_seed)
            import utils.torus  # noqa: F401 (loads .p.npy/.score.npy, re-draws the Monte-Carlo table under the seed)
        finally:
            os.chdir(cwd)
    else:
        # callers that never evaluate the score normalisers (confidence model): skip the minutes-long table build
        for name in ("so3", "torus"):
            mock = MagicMock(name=f"utils.{name}")
            sys.modules[f"utils.{name}"] = mock
            setattr(sys.modules["utils"], name, mock)
    return hetero


def reference_score_model(state_dict=None, dropout=None, asyncronous_noise_schedule=False):
    """Construct the REFERENCE TensorProductScoreModel class with the shipped yml's kwargs
    (utils/utils.py:239-283 mapping), optionally loading a state dict produced by the build."""
    from functools import partial
    import torch
    from confidence_bootstrapping_amd.utils import load_model_args
    from utils.diffusion_utils import t_to_sigma as ref_t_to_sigma, get_timestep_embedding
    from models.score_model import TensorProductScoreModel as RefModel
    args = load_model_args()
    if dropout is not None:
        args.dropout = dropout
    emb = get_timestep_embedding(embedding_type=args.embedding_type, embedding_dim=args.sigma_embed_dim,
                                 embedding_scale=args.embedding_scale)
    model = RefModel(t_to_sigma=partial(ref_t_to_sigma, args=args), device=torch.device("cpu"), no_torsion=args.no_torsion,
                     timestep_emb_func=emb, num_conv_layers=args.num_conv_layers, lig_max_radius=args.max_radius,
                     scale_by_sigma=args.scale_by_sigma, sigma_embed_dim=args.sigma_embed_dim, norm_by_sigma=False,
                     ns=args.ns, nv=args.nv, distance_embed_dim=args.distance_embed_dim,
                     cross_distance_embed_dim=args.cross_distance_embed_dim, batch_norm=not args.no_batch_norm,
                     dropout=args.dropout, use_second_order_repr=args.use_second_order_repr,
                     cross_max_distance=args.cross_max_distance, dynamic_max_cross=args.dynamic_max_cross,
                     separate_noise_schedule=False, smooth_edges=False, odd_parity=False,
                     lm_embedding_type="precomputed", confidence_mode=False, asyncronous_noise_schedule=asyncronous_noise_schedule,
                     fixed_center_conv=not args.not_fixed_center_conv, no_aminoacid_identities=False,
                     include_miscellaneous_atoms=False, sh_lmax=args.sh_lmax, differentiate_convolutions=True,
                     tp_weights_layers=args.tp_weights_layers, num_prot_emb_layers=args.num_prot_emb_layers,
                     reduce_pseudoscalars=args.reduce_pseudoscalars, embed_also_ligand=args.embed_also_ligand,
                     atom_confidence=False, sidechain_pred=False, depthwise_convolution=False)
    if state_dict is not None:
        missing = model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model, args


def subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None):
    """torch_geometric.utils.subgraph for a boolean node mask (the only form utils/utils.py:409-417 uses)."""
    import torch
    assert subset.dtype == torch.bool
    m = subset[edge_index[0]] & subset[edge_index[1]]
    ei = edge_index[:, m]
    if relabel_nodes:
        ei = (torch.cumsum(subset.long(), 0) - 1)[ei]
    return ei, (edge_attr[m] if edge_attr is not None else None)


def reference_confidence_model(state_dict=None):
    """Construct the REFERENCE all-atom TensorProductScoreModel in confidence mode through the reference's own
    get_model() (utils/utils.py:175-288) with the confidence yml, optionally loading a state dict of the build."""
    import torch
    import utils.utils as ref_utils
    from confidence_bootstrapping_amd.utils import load_model_args, _CONFIDENCE_YML
    ref_utils.subgraph = subgraph
    args = load_model_args(_CONFIDENCE_YML)
    model = ref_utils.get_model(args, torch.device("cpu"), t_to_sigma=None, no_parallel=True, confidence_mode=True)
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    model.eval()
    return model, args
