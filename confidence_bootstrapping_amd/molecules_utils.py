"""Symmetry-corrected ligand RMSD (reference utils/molecules_utils.py:3-18 -> vendored spyrmsd `rmsd.symmrmsd`,
spyrmsd/rmsd.py:116-203,209-303) for batches of poses on the MI355X.

Host: graph isomorphisms of the molecular graph (atomic numbers as node labels) with networkx, exactly the enumeration
spyrmsd's networkx back-end performs (spyrmsd/graphs/nx.py:52-100).  Device: for every pose the minimum over isomorphisms of
the summed squared displacement, then sqrt(min / n)  (`center=False, minimize=False`, the reference's call) -- `cbd_symm_rmsd`.
There is no CPU fallback for the reduction."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple, Union

import numpy as np
import torch


class IsomorphismLimit(RuntimeError):
    """The enumeration of graph isomorphisms exceeded its time or count bound (highly symmetric ligand)."""


# The reference wraps the symmetry-corrected RMSD in `time_limit(10)` and falls back to the plain RMSD when it fires
# (inference.py:511-520, finetune_train.py:205-214).  The same bound here, plus a count cap so that the [K, N] index tables
# handed to the GPU stay small (ligands with factorially many automorphisms would otherwise exhaust host memory first).
MAX_ISOMORPHISMS = 100_000
ISOMORPHISM_TIME_LIMIT_S = 10.0


def graph_isomorphisms(atomicnums, adjacency, atomicnums2=None, adjacency2=None, max_isomorphisms=None,
                       time_limit_s=None) -> Tuple[np.ndarray, np.ndarray]:
    """All label-preserving isomorphisms between two molecular graphs as index arrays (idx1 [K,N], idx2 [K,N]) such that atom
    idx1[k, i] of molecule 1 corresponds to atom idx2[k, i] of molecule 2.  Raises ValueError if the graphs differ and
    IsomorphismLimit when more than `max_isomorphisms` exist or the enumeration takes longer than `time_limit_s` seconds
    (the callers then use the uncorrected RMSD, like the reference after its time_limit)."""
    import time
    import networkx as nx
    max_isomorphisms = MAX_ISOMORPHISMS if max_isomorphisms is None else max_isomorphisms
    time_limit_s = ISOMORPHISM_TIME_LIMIT_S if time_limit_s is None else time_limit_s
    a1, m1 = np.asarray(atomicnums), np.asarray(adjacency)
    a2 = a1 if atomicnums2 is None else np.asarray(atomicnums2)
    m2 = m1 if adjacency2 is None else np.asarray(adjacency2)
    G1, G2 = nx.Graph(m1), nx.Graph(m2)
    nx.set_node_attributes(G1, {i: v for i, v in enumerate(a1.tolist())}, "aprops")
    nx.set_node_attributes(G2, {i: v for i, v in enumerate(a2.tolist())}, "aprops")
    gm = nx.algorithms.isomorphism.GraphMatcher(G1, G2, lambda x, y: x["aprops"] == y["aprops"])
    if not gm.is_isomorphic():
        raise ValueError("Graphs are not isomorphic.")
    iso, t0 = [], time.monotonic()
    for m in gm.isomorphisms_iter():
        iso.append((list(m.keys()), list(m.values())))
        if len(iso) > max_isomorphisms:
            raise IsomorphismLimit(f"more than {max_isomorphisms} graph isomorphisms")
        if (len(iso) & 63) == 0 and time.monotonic() - t0 > time_limit_s:
            raise IsomorphismLimit(f"graph isomorphism enumeration exceeded {time_limit_s} s ({len(iso)} found)")
    idx1 = np.asarray([i for i, _ in iso], dtype=np.int32)
    idx2 = np.asarray([j for _, j in iso], dtype=np.int32)
    return idx1, idx2


def symmetry_rmsd(coords_ref, coords, atomicnums, adjacency, atomicnums2=None, adjacency2=None, device=None,
                  return_permutation=False, isomorphisms=None):
    """spyrmsd.rmsd.symmrmsd(coords_ref, coords, atomicnums, atomicnums2, adjacency, adjacency2) for one pose [N,3] or a
    batch / list of poses [B,N,3]: float or list of floats like the reference (+ the minimising (idx1, idx2) pairs)."""
    from .engine import load_library, _check, _dptr
    lib = load_library()
    single = not isinstance(coords, (list, tuple)) and np.asarray(coords if not torch.is_tensor(coords) else coords.cpu()).ndim == 2
    # default device: the tensors' own, else this process's CURRENT device (one process per GPU: never everybody's cuda:0)
    dev = torch.device(device) if device is not None else (coords.device if torch.is_tensor(coords) and coords.is_cuda
                                                            else torch.device("cuda", torch.cuda.current_device()))
    if dev.type != "cuda":
        raise RuntimeError("symmetry_rmsd runs on an MI355X (device type 'cuda' under ROCm)")
    to_t = lambda x: x.to(dev, torch.float32) if torch.is_tensor(x) else torch.as_tensor(np.asarray(x), dtype=torch.float32, device=dev)
    pos = to_t(coords[None] if single and torch.is_tensor(coords) else (np.asarray(coords)[None] if single else
               (torch.stack([to_t(c) for c in coords]) if isinstance(coords, (list, tuple)) else coords))).contiguous()
    ref = to_t(coords_ref).contiguous()
    if isomorphisms is None:
        isomorphisms = graph_isomorphisms(atomicnums, adjacency, atomicnums2, adjacency2)
    idx1, idx2 = isomorphisms
    B, N, K = pos.shape[0], pos.shape[1], idx1.shape[0]
    if ref.shape != (N, 3) or idx1.shape != (K, N):
        raise ValueError("coordinate / isomorphism shapes do not match")
    d1 = torch.as_tensor(idx1, dtype=torch.int32, device=dev).contiguous()
    d2 = torch.as_tensor(idx2, dtype=torch.int32, device=dev).contiguous()
    out = torch.empty(B, device=dev)
    arg = torch.empty(B, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _check(lib.cbd_symm_rmsd(B, N, K, _dptr(pos), _dptr(ref), _dptr(d1), _dptr(d2), _dptr(out), _dptr(arg), stream))
    vals = out.cpu().tolist()
    res = vals[0] if single else vals
    if return_permutation:
        a = arg.cpu().tolist()
        perms = [(idx1[k].tolist(), idx2[k].tolist()) for k in a]
        return res, (perms[0] if single else perms)
    return res


def _graph_of(mol):
    """(atomicnums, adjacency) of a spyrmsd-Molecule-like object or an rdkit Mol (heavy atoms as given)."""
    if hasattr(mol, "atomicnums") and hasattr(mol, "adjacency_matrix"):
        return np.asarray(mol.atomicnums), np.asarray(mol.adjacency_matrix)
    if hasattr(mol, "GetAtoms") and hasattr(mol, "GetBonds"):
        nums = np.asarray([a.GetAtomicNum() for a in mol.GetAtoms()])
        am = np.zeros((len(nums), len(nums)), dtype=int)
        for b in mol.GetBonds():
            i, j = b.GetBeginAtomIdx(), b.GetEndAtomIdx()
            am[i, j] = am[j, i] = 1
        return nums, am
    raise TypeError("mol must provide atomicnums/adjacency_matrix (spyrmsd Molecule) or the rdkit Mol API")


class _HeavyAtomGraph:
    """what rdkit's RemoveAllHs leaves of a molecule, as far as the symmetry-corrected RMSD needs it: atomic numbers and adjacency"""

    def __init__(self, atomicnums, adjacency_matrix):
        self.atomicnums, self.adjacency_matrix = atomicnums, adjacency_matrix


def remove_all_hs(mol):
    """Heavy-atom graph of `mol` (the reference passes RemoveAllHs(orig_complex_graph.mol[0]) to get_symmetry_rmsd next to coordinates
    filtered with filterHs: utils/training.py:352, finetune_train.py:210).  A molecule without hydrogens comes back unchanged; None stays
    None (the callers then take their non-symmetry-corrected branch, as they do for any failure)."""
    if mol is None:
        return None
    nums, am = _graph_of(mol)
    keep = nums != 1
    if keep.all():
        return mol
    return _HeavyAtomGraph(nums[keep], am[np.ix_(keep, keep)])


def get_symmetry_rmsd(mol, coords1, coords2, mol2=None, return_permutation=False, device=None):
    """Same call as the reference (utils/molecules_utils.py:3): coords1 = reference pose, coords2 = pose or list of poses.
    `device` (extension): the GPU to reduce on; default = the poses' device or the process's current device."""
    n1, a1 = _graph_of(mol)
    n2, a2 = _graph_of(mol2) if mol2 is not None else (n1, a1)
    return symmetry_rmsd(coords1, coords2, n1, a1, n2, a2, device=device, return_permutation=return_permutation)
