"""Golden vectors for the ligand side of the featurisation: tests/golden/g17_torsion_masks.npz.

TEST INFRASTRUCTURE ONLY; needs /root/reference (this container).  RUNS the reference's own `get_transformation_mask`
(utils/torsion.py:15-45, networkx is installed here) on
  * the heavy-atom graph of data/1a0q/1a0q_ligand.sdf (atoms / bonds in file order, hydrogens removed -- the edge list the reference's
    get_lig_graph builds, datasets/process_mols.py:567-589), and
  * 24 seeded random molecular graphs: trees with branches, fused and spiro rings, double-bond bridges (the function has no bond-order
    test), bridges with one atom on a side, equal-sized halves (tie rule), two-fragment ligands (the smallest component of the whole
    graph quirk) and a duplicated bond (the DiGraph edge-count bound).
What is NOT the reference's code: torch_geometric's `to_networkx` / `HeteroData.to_homogeneous` (absent here), restated below from their
published semantics for the one graph the reference passes in -- a ligand-only heterograph: nodes 0..N-1 in order, one directed edge per
edge_index column.  rdkit is absent, so the 1a0q edge list comes from the build's own SDF reader; the file's atom and bond records are
stored next to it so that the test can check the reader against the raw file as well."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
SDF = "/root/reference/data/1a0q/1a0q_ligand.sdf"


class _Store:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _Homogeneous:
    def __init__(self, n, edge_index):
        self.num_nodes, self.edge_index = n, edge_index


class LigandOnlyGraph:
    """what the reference's complex_graph is when get_lig_graph_with_matching calls get_transformation_mask (process_mols.py:650)"""

    def __init__(self, n, edge_index):
        self.n, self.ei = n, torch.as_tensor(edge_index, dtype=torch.long).reshape(2, -1)

    def __getitem__(self, key):
        assert key == ("ligand", "ligand")
        return _Store(edge_index=self.ei)

    def to_homogeneous(self):
        return _Homogeneous(self.n, self.ei)


def to_networkx(data, to_undirected=False):
    import networkx as nx
    G = nx.Graph() if to_undirected else nx.DiGraph()
    G.add_nodes_from(range(data.num_nodes))
    for u, v in data.edge_index.t().tolist():
        G.add_edge(u, v)
    return G


def random_graph(rng, kind):
    """bond list [(a, b)] of a small molecule-like graph"""
    n = int(rng.integers(6, 30))
    bonds = [(int(rng.integers(0, k)), k) for k in range(1, n)]                 # random tree: branches
    if kind in ("rings", "fused", "mixed"):
        for _ in range(int(rng.integers(1, 4))):                                # ring closures
            a, b = sorted(rng.choice(n, 2, replace=False).tolist())
            if (a, b) not in bonds and (b, a) not in bonds:
                bonds.append((a, b))
    if kind == "fragments":                                                     # a second fragment of 1..4 atoms
        m = int(rng.integers(1, 5))
        bonds += [(n + k - 1, n + k) for k in range(1, m)]
        n += m
        if m == 1:
            pass                                                                # isolated atom: a component of size one
    if kind == "halves":                                                        # two equal halves joined by one bond (tie rule)
        h = n // 2
        bonds = [(k - 1, k) for k in range(1, h)] + [(h + k - 1, h + k) for k in range(1, h)] + [(int(rng.integers(0, h)), h + int(rng.integers(0, h)))]
        n = 2 * h
    if kind == "duplicate" and bonds:
        bonds.append(bonds[int(rng.integers(0, len(bonds)))])
    perm = rng.permutation(n)                                                   # atom numbering unrelated to the construction order
    bonds = [(int(perm[a]), int(perm[b])) if rng.random() < 0.5 else (int(perm[b]), int(perm[a])) for a, b in bonds]
    order = rng.permutation(len(bonds))
    return n, [bonds[k] for k in order]


def edge_index_of(bonds):
    row, col = [], []
    for a, b in bonds:
        row += [a, b]
        col += [b, a]
    return np.asarray([row, col], dtype=np.int64).reshape(2, -1)


def main():
    from oracle import ref_import
    ref_import.install(load_tables=False)
    import utils.torsion as rt
    rt.to_networkx = to_networkx
    from confidence_bootstrapping_amd.datasets.molfile import read_sdf, remove_hs
    out = {}
    mol = remove_hs(read_sdf(SDF))
    n = mol.GetNumAtoms()
    bonds = [(b.GetBeginAtomIdx(), b.GetEndAtomIdx()) for b in mol.GetBonds()]
    ei = edge_index_of(bonds)
    me, mr = rt.get_transformation_mask(LigandOnlyGraph(n, ei))
    out["sdf_edge_index"], out["sdf_mask_edges"], out["sdf_mask_rotate"] = ei, me, mr
    out["sdf_bond_type"] = np.asarray([b.type for b in mol.GetBonds()], dtype=np.int64)       # as in the file (before any perception)
    out["sdf_z"] = np.asarray([a.GetAtomicNum() for a in mol.GetAtoms()], dtype=np.int64)
    out["sdf_pos"] = mol.GetConformer().GetPositions()
    print("1a0q: atoms", n, "bonds", len(bonds), "rotatable", int(me.sum()))
    rng = np.random.default_rng(17)
    kinds = ["tree"] * 4 + ["rings"] * 5 + ["fused"] * 3 + ["mixed"] * 3 + ["fragments"] * 4 + ["halves"] * 3 + ["duplicate"] * 2
    ei_all, off, me_all, mr_all, ns = [], [0], [], [], []
    for k, kind in enumerate(kinds):
        n, bonds = random_graph(rng, kind)
        ei = edge_index_of(bonds)
        me, mr = rt.get_transformation_mask(LigandOnlyGraph(n, ei))
        ei_all.append(ei)
        off.append(off[-1] + ei.shape[1])
        me_all.append(me)
        mr_all.append(np.packbits(mr, axis=1) if mr.size else np.zeros((mr.shape[0], (n + 7) // 8), dtype=np.uint8))
        ns.append(n)
        print(kind, "n", n, "bonds", len(bonds), "rotatable", int(me.sum()))
    out["rand_kinds"] = np.asarray(kinds)
    out["rand_n"] = np.asarray(ns, dtype=np.int64)
    out["rand_edge_index"] = np.concatenate(ei_all, axis=1)
    out["rand_offsets"] = np.asarray(off, dtype=np.int64)
    out["rand_mask_edges"] = np.concatenate(me_all)
    for k, m in enumerate(mr_all):
        out[f"rand_mask_rotate_packed_{k}"] = m
    path = os.path.join(ROOT, "tests", "golden", "g17_torsion_masks.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
