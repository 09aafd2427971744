"""Readers for the reference's pre-processed dataset caches -- without torch_geometric, rdkit or prody.

The reference featurises complexes once (rdkit / prody / ESM, `datasets/process_mols.py:415-589,744-857`) and pickles the resulting
`torch_geometric.data.HeteroData` graphs: `receptors{i}.pkl`, `ligands.pkl`, `rdkit_ligands.pkl` (`datasets/moad.py:297-470`), ESM
embeddings as a `.pt` dictionary (`moad.py:309`).  The MI355X engines need only the arrays inside those graphs
(`engine.DockEngine.set_complex`, `ConfidenceEngine.set_complex`), so a user who has caches written by the reference's own
environment can run inference here from them:

  * `load_pyg_cache(path)`     -- restricted unpickler: maps the pickled PyG 2.0.4 classes (`HeteroData`, `BaseStorage`, `NodeStorage`,
                                  `EdgeStorage`; layout of torch_geometric/data/{hetero_data,storage}.py at that version) onto
                                  `hetero.HeteroData`, rebuilds tensors / numpy arrays, keeps rdkit molecules as opaque blobs and
                                  REFUSES every other global (a cache file is untrusted input; plain `pickle.load` would execute it);
  * `complex_from_arrays(...)` -- the same graph from plain arrays / a dict (the schema of `process_mols.py:448-526`), validated;
  * `attach_lm_embeddings(...)`-- `receptor.x = [residue type | ESM embedding]` as `new_extract_receptor_structure` builds it
                                  (`process_mols.py:489-490`) from the per-chain `.pt` dictionary;
  * `merge_ligand_receptor(...)` -- `moad.py:202-212`: ligand stores copied into the receptor graph, poses re-centred on
                                  `original_center`.
STATUS: validated against a pickle EMULATED by this repository (oracle/make_cache_fixture.py writes the PyG 2.0.4 class layout by hand);
no real torch_geometric pickle existed in the build image.  tools/pin_with_rdkit.py (run where rdkit + torch_geometric are installed)
writes a real one and diffs what this reader returns.
What is NOT here: producing the features from PDB / SDF files (rdkit chemistry, prody selections, ESM inference).
"""
from __future__ import annotations

import io
import pickle
from typing import Any, Dict, Optional

import numpy as np
import torch

from ..hetero import HeteroData

LIG_FEATURE_DIMS = [119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2]     # datasets/process_mols.py:95-112
REC_RESIDUE_TYPES = 38
REC_ATOM_FEATURE_DIMS = [38, 119, 23, 38]                                  # datasets/process_mols.py:114-119


# ------------------------------------------------------------------------------------------------- restricted unpickler
class _PygStorage:
    """Stand-in for torch_geometric.data.storage.{Base,Node,Edge,Global}Storage: state = the instance __dict__
    (`_mapping` {attr: value}, `_key`, `_parent`), storage.py `__getstate__` / `__setstate__` of PyG 2.0.4."""

    def __init__(self, *a, **k):
        self._mapping = {}

    def __setstate__(self, state):
        self.__dict__.update(state)


class _PygHeteroData:
    """Stand-in for torch_geometric.data.hetero_data.HeteroData: default pickling of `__dict__` = `_global_store`,
    `_node_store_dict` {node type: NodeStorage}, `_edge_store_dict` {(src, rel, dst): EdgeStorage}."""

    def __init__(self, *a, **k):
        pass

    def __setstate__(self, state):
        self.__dict__.update(state)


class OpaqueObject:
    """A pickled object the engines do not need (rdkit molecules): kept as its constructor arguments, never executed."""

    def __init__(self, *args):
        self.args = args

    def __setstate__(self, state):
        self.state = state


def _load_storage_from_bytes(b):
    # torch.storage._load_from_bytes is `torch.load(io.BytesIO(b))`: restrict it to tensors / storages
    return torch.load(io.BytesIO(b), weights_only=True)


_SAFE_BUILTINS = {"set", "frozenset", "slice", "complex", "bytearray", "range", "tuple", "list", "dict", "int", "float", "bool", "str", "bytes"}
_NUMPY_OK = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
             ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy.core.numeric", "_frombuffer"),
             ("numpy._core.numeric", "_frombuffer")}
_TORCH_OK = {("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
             ("torch", "Size"), ("torch", "device"), ("torch", "dtype")}
_TORCH_STORAGES = {"FloatStorage", "DoubleStorage", "LongStorage", "IntStorage", "ShortStorage", "CharStorage", "ByteStorage", "BoolStorage",
                   "HalfStorage", "BFloat16Storage", "UntypedStorage"}


class RestrictedUnpickler(pickle.Unpickler):
    """Unpickler for the reference's graph caches: an allow-list of constructors, everything else raises UnpicklingError."""

    def find_class(self, module, name):
        if module.startswith("torch_geometric.data"):
            if name in ("HeteroData", "Data", "Batch"):
                return _PygHeteroData
            if name.endswith("Storage"):
                return _PygStorage
        if module == "collections" and name in ("OrderedDict", "defaultdict"):
            import collections
            return getattr(collections, name)
        if module == "builtins" and name in _SAFE_BUILTINS:
            import builtins
            return getattr(builtins, name)
        if (module, name) in _NUMPY_OK:
            mod = __import__(module, fromlist=[name])
            return getattr(mod, name)
        if (module, name) in _TORCH_OK:
            mod = __import__(module, fromlist=[name])
            return getattr(mod, name)
        if module == "torch.storage" and name == "_load_from_bytes":
            return _load_storage_from_bytes
        if module == "torch" and name in _TORCH_STORAGES:
            return getattr(torch, name)
        if module.startswith("rdkit."):
            return OpaqueObject
        raise pickle.UnpicklingError(f"refusing to load {module}.{name}: not part of the graph-cache schema")


def _convert(obj):
    """PyG stand-ins -> hetero.HeteroData (recursively through lists / dicts / tuples)."""
    if isinstance(obj, _PygHeteroData):
        g = HeteroData()
        glob = getattr(obj, "_global_store", None)
        for k, v in (getattr(glob, "_mapping", {}) or {}).items():
            setattr(g, k, _convert(v))
        for key, st in (getattr(obj, "_node_store_dict", {}) or {}).items():
            for k, v in st._mapping.items():
                setattr(g[key], k, _convert(v))
        for key, st in (getattr(obj, "_edge_store_dict", {}) or {}).items():
            for k, v in st._mapping.items():
                setattr(g[tuple(key)], k, _convert(v))
        return g
    if isinstance(obj, dict):
        return {k: _convert(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_convert(v) for v in obj]
    if isinstance(obj, tuple):
        return tuple(_convert(v) for v in obj)
    return obj


def load_pyg_cache(path_or_file):
    """`pickle.load` of a reference cache file (`receptors{i}.pkl`, `ligands.pkl`, a single graph, a list or a dict of graphs) with the
    restricted unpickler; PyG graphs come back as `hetero.HeteroData`."""
    if hasattr(path_or_file, "read"):
        return _convert(RestrictedUnpickler(path_or_file).load())
    with open(path_or_file, "rb") as f:
        return _convert(RestrictedUnpickler(f).load())


def load_lm_embeddings(path):
    """The ESM `.pt` dictionary (`moad.py:309`: {'<pdb>_chain_<i>': tensor [L, 1280]}) with weights_only loading."""
    return torch.load(path, map_location="cpu", weights_only=True)


# ------------------------------------------------------------------------------------------------- plain arrays
def _t(x, dtype):
    return x.to(dtype) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x), dtype=dtype)


def complex_from_arrays(ligand: Dict[str, Any], receptor: Dict[str, Any], atoms: Optional[Dict[str, Any]] = None, name=None,
                        original_center=None) -> HeteroData:
    """One complex graph from plain arrays in the cache schema (`process_mols.py:448-526,567-589`):
      ligand:   x [Nl,16] int, pos [Nl,3], edge_index [2, 2*bonds], edge_attr [2*bonds,4], edge_mask [2*bonds] bool, mask_rotate [R,Nl] bool
      receptor: x [Nr, 1 (+1280)] float (col 0 residue type), pos [Nr,3], edge_index [2,Err]  (optional side_chain_vecs, chain_ids)
      atoms:    x [Na,4] float, pos [Na,3], edge_index [2,Eaa], atom_res [Na] (residue of every atom)   (all-atom confidence model)
    Shapes, dtypes and index / category ranges are validated here so that a malformed cache fails with a message, not in a kernel."""
    g = HeteroData()
    lx = _t(ligand["x"], torch.long)
    Nl = lx.shape[0]
    if lx.ndim != 2 or lx.shape[1] != len(LIG_FEATURE_DIMS):
        raise ValueError(f"ligand.x must be [Nl, {len(LIG_FEATURE_DIMS)}]")
    if bool((lx < 0).any()) or any(int(lx[:, i].max()) >= d for i, d in enumerate(LIG_FEATURE_DIMS)):
        raise ValueError("ligand.x has categorical features outside lig_feature_dims (process_mols.py:95-112)")
    ei = _t(ligand["edge_index"], torch.long)
    ea = _t(ligand["edge_attr"], torch.float32)
    em = _t(ligand["edge_mask"], torch.bool)
    mr = np.asarray(ligand["mask_rotate"].cpu() if torch.is_tensor(ligand["mask_rotate"]) else ligand["mask_rotate"], dtype=bool)
    if ei.shape[0] != 2 or ea.shape != (ei.shape[1], 4) or em.shape != (ei.shape[1],):
        raise ValueError("ligand bond arrays are inconsistent")
    if ei.numel() and (int(ei.min()) < 0 or int(ei.max()) >= Nl):
        raise ValueError("ligand.edge_index out of range")
    R = int(em.sum())
    if mr.size and mr.shape != (R, Nl):
        raise ValueError(f"mask_rotate must be [{R}, {Nl}] (one row per masked bond direction, utils/torsion.py:15-45)")
    g["ligand"].x, g["ligand"].pos = lx, _t(ligand["pos"], torch.float32).reshape(Nl, 3)
    g["ligand"].edge_mask, g["ligand"].mask_rotate = em, mr.reshape(R, Nl)
    if "orig_pos" in ligand:
        g["ligand"].orig_pos = np.asarray(ligand["orig_pos"], dtype=np.float32)
    g["ligand", "ligand"].edge_index, g["ligand", "ligand"].edge_attr = ei, ea
    rx = _t(receptor["x"], torch.float32)
    Nr = rx.shape[0]
    if rx.ndim != 2 or rx.shape[1] not in (1, 1 + 1280):
        raise ValueError("receptor.x must be [Nr, 1] or [Nr, 1 + 1280] (residue type | ESM embedding)")
    if bool((rx[:, 0] < 0).any()) or float(rx[:, 0].max()) >= REC_RESIDUE_TYPES:
        raise ValueError("receptor residue types outside possible_amino_acids (process_mols.py:83-85)")
    rei = _t(receptor["edge_index"], torch.long)
    if rei.shape[0] != 2 or (rei.numel() and (int(rei.min()) < 0 or int(rei.max()) >= Nr)):
        raise ValueError("receptor.edge_index out of range")
    g["receptor"].x, g["receptor"].pos = rx, _t(receptor["pos"], torch.float32).reshape(Nr, 3)
    for k in ("side_chain_vecs", "chain_ids", "sequence"):
        if k in receptor:
            setattr(g["receptor"], k, receptor[k])
    g["receptor", "rec_contact", "receptor"].edge_index = rei
    if atoms is not None:
        ax = _t(atoms["x"], torch.float32)
        Na = ax.shape[0]
        if ax.ndim != 2 or ax.shape[1] != len(REC_ATOM_FEATURE_DIMS):
            raise ValueError("atom.x must be [Na, 4]")
        ares = _t(atoms["atom_res"], torch.long)
        aei = _t(atoms["edge_index"], torch.long)
        if ares.shape != (Na,) or (Na and (int(ares.min()) < 0 or int(ares.max()) >= Nr)):
            raise ValueError("atom_res must map every atom to a residue")
        if aei.shape[0] != 2 or (aei.numel() and (int(aei.min()) < 0 or int(aei.max()) >= Na)):
            raise ValueError("atom edge_index out of range")
        g["atom"].x, g["atom"].pos = ax, _t(atoms["pos"], torch.float32).reshape(Na, 3)
        g["atom", "atom_contact", "atom"].edge_index = aei
        g["atom", "atom_rec_contact", "receptor"].edge_index = torch.stack([torch.arange(Na), ares])
    g.original_center = torch.zeros(1, 3) if original_center is None else _t(original_center, torch.float32).reshape(1, 3)
    if name is not None:
        g.name = name
    return g


def attach_lm_embeddings(g: HeteroData, embeddings) -> HeteroData:
    """receptor.x = cat([residue type, ESM embedding]) (`process_mols.py:487-490`).  `embeddings`: a tensor [Nr, 1280] or the list of
    per-chain tensors in chain order (the values of the `.pt` dictionary for this complex)."""
    emb = torch.cat([_t(e, torch.float32) for e in embeddings], 0) if isinstance(embeddings, (list, tuple)) else _t(embeddings, torch.float32)
    x = g["receptor"].x
    if emb.shape != (x.shape[0], 1280):
        raise ValueError(f"language-model embeddings have shape {tuple(emb.shape)}, expected ({x.shape[0]}, 1280)")
    g["receptor"].x = torch.cat([x[:, :1].float(), emb], 1)
    return g


def merge_ligand_receptor(receptor_graph: HeteroData, ligand_graph: HeteroData) -> HeteroData:
    """`datasets/moad.py:202-212` (get_by_name): every ligand node / edge store is copied into (a shallow copy of) the receptor graph,
    the name follows the ligand and the ligand pose(s) are re-centred on the receptor's `original_center`."""
    g = receptor_graph.shallow_copy()
    for key in list(ligand_graph.node_types) + list(ligand_graph.edge_types):
        for k in ligand_graph[key].keys():
            setattr(g[key], k, getattr(ligand_graph[key], k))
    if hasattr(ligand_graph, "name"):
        g.name = ligand_graph.name
    center = getattr(g, "original_center", None)
    if center is not None:
        pos = g["ligand"].pos
        g["ligand"].pos = [p - center for p in pos] if isinstance(pos, list) else pos - center
    return g
