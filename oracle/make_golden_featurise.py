"""Golden vectors for the receptor featurisation: tests/golden/g15_featurise_1a0q.npz.

TEST INFRASTRUCTURE ONLY; needs /root/reference (this container).  RUNS the reference's own array code on the residues of
data/1a0q/1a0q_protein_processed.pdb:
  datasets/process_mols.py::new_extract_receptor_structure (448-526: side_chain_vecs, residue features, edge lists for BOTH branches --
      knn_only_graph as in the shipped ymls and the cutoff / max_neighbors loops -- the all-atom stores), ::get_moad_atom_feats (532-564),
  datasets/parse_chi.py::get_chi_angles / get_onehot_sequence (78-113).
What is NOT the reference's code: the PDB reader (prody is absent: `seq` and the [N,14,3] coordinate array come from the build's
parse_pdb, in the 14-slot layout of the reference's own datasets/constants.py::atom_order, which this script cross-checks) and
torch_cluster.knn_graph, restated below from its published semantics (k nearest by Euclidean distance, self excluded, rows
[neighbour; centre], centres ascending) in float64 -- parity unpinned at that boundary."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
PDB = "/root/reference/data/1a0q/1a0q_protein_processed.pdb"


def knn_graph(x, k, **kw):
    d = torch.cdist(x.double(), x.double())
    d.fill_diagonal_(float("inf"))
    nbr = torch.argsort(d, dim=1, stable=True)[:, :k]
    n = x.shape[0]
    return torch.stack([nbr.reshape(-1), torch.arange(n).repeat_interleave(k)])


def main():
    from oracle import ref_import
    hetero = ref_import.install(load_tables=False)
    sys.modules["torch_cluster"].knn_graph = knn_graph
    # rdkit is mocked: give the module-level periodic table of process_mols a real GetAtomicNumber for the four protein elements
    import datasets.process_mols as rpm
    import datasets.constants as rc
    import datasets.parse_chi as rchi
    rpm.knn_graph = knn_graph
    rpm.periodic_table.GetAtomicNumber = lambda s: {"C": 6, "N": 7, "O": 8, "S": 16}[s]
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    # the build's constants against the reference's own tables
    assert {k: list(v) for k, v in rc.atom_order.items()} == pm.atom_order
    assert rc.aa_short2long == pm.aa_short2long and rc.aa_long2short == pm.aa_long2short
    assert rpm.allowable_features["possible_amino_acids"] == pm.AMINO_ACIDS
    assert rpm.allowable_features["possible_atom_type_2"] == pm.ATOM_TYPE_2 and rpm.allowable_features["possible_atom_type_3"] == pm.ATOM_TYPE_3
    for aa in rc.atom_order:
        ref_rows = rchi.dihedral_indices[aa]
        assert np.array_equal(np.isnan(ref_rows), np.isnan(pm._CHI_SLOTS[aa])) and np.array_equal(np.nan_to_num(ref_rows), np.nan_to_num(pm._CHI_SLOTS[aa])), aa
    pdb = pm.parse_pdb(PDB)
    seq, coords = pdb.seq, pdb.coords
    # PDB coordinates have three decimals: stored exactly as integer milli-Angstrom (absent slot: INT32_MIN); x / 1000.0 in float64
    # is the double the parser read
    milli = np.where(np.isnan(coords), np.iinfo(np.int32).min, np.round(np.nan_to_num(coords) * 1000)).astype(np.int32)
    assert np.array_equal(np.where(milli == np.iinfo(np.int32).min, np.nan, milli / 1000.0), coords, equal_nan=True)
    out = {"seq": np.array(list(seq)), "coords_milli": milli}
    chi = rchi.get_chi_angles(coords.copy(), seq)
    out["chi"] = chi
    rng = np.random.default_rng(0)
    lm = [rng.normal(0, 0.5, size=(len(seq), 8)).astype(np.float32)]
    for tag, knn_only in (("knn", True), ("cut", False)):
        g = hetero.HeteroData()
        rpm.new_extract_receptor_structure(seq, coords.copy(), g, neighbor_cutoff=15.0, max_neighbors=24, lm_embeddings=lm,
                                           knn_only_graph=knn_only, all_atoms=True, atom_cutoff=5, atom_max_neighbors=8)
        out[f"{tag}_rec_edge_index"] = g["receptor", "receptor"].edge_index.numpy().astype(np.int32)
        out[f"{tag}_atom_edge_index"] = g["atom", "atom"].edge_index.numpy().astype(np.int32)
        if tag == "knn":          # the node stores do not depend on the branch
            out["rec_x"] = g["receptor"].x.numpy()
            out["rec_pos"] = g["receptor"].pos.numpy()
            out["side_chain_vecs"] = g["receptor"].side_chain_vecs.numpy()
            out["atom_x"] = g["atom"].x.numpy().astype(np.int16)
            out["atom_pos"] = g["atom"].pos.numpy()
            out["atom_res"] = g["atom", "receptor"].edge_index.numpy().astype(np.int32)
    # a small cutoff so that the "no neighbour -> nearest node" branch (process_mols.py:470-474) is exercised too
    g = hetero.HeteroData()
    rpm.new_extract_receptor_structure(seq, coords.copy(), g, neighbor_cutoff=4.2, max_neighbors=3, lm_embeddings=None,
                                       knn_only_graph=False, all_atoms=False)
    out["tight_rec_edge_index"] = g["receptor", "receptor"].edge_index.numpy().astype(np.int32)
    path = os.path.join(ROOT, "tests", "golden", "g15_featurise_1a0q.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
