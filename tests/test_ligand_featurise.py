"""Ligand side of the featurisation: `datasets/molfile.py`, `datasets/process_mols.py` (bottom), `torsion.py`.

Pinned by the reference: `get_transformation_mask` against tests/golden/g17_torsion_masks.npz, which oracle/make_golden_ligand.py
produced by RUNNING the reference's utils/torsion.py:15-45 on the heavy-atom graph of data/1a0q/1a0q_ligand.sdf and on 24 random
graphs (branches, rings, double-bond bridges, equal halves, several fragments, a duplicated bond) -- bit for bit.
Checked against the raw files (tests/golden/1a0q/, the reference's example complex): atom order, coordinates, bond order and bond types
of the SDF reader, the heavy-atom graph of the MOL2 reader.
NOT pinned (rdkit is absent; molfile.LIG_FEATURE_SOURCES says which column is which): the perceived features -- aromaticity,
hybridisation, chirality.  They are checked on molecules whose textbook answer is not in doubt.
GPU: the full 1a0q complex from its three files through both engines."""
import gzip
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
D = os.path.join(HERE, "golden", "1a0q")
SDF, MOL2 = os.path.join(D, "1a0q_ligand.sdf"), os.path.join(D, "1a0q_ligand.mol2")


@pytest.fixture(scope="module")
def g17():
    return dict(np.load(os.path.join(HERE, "golden", "g17_torsion_masks.npz")))


def _graph(n, edge_index):
    from confidence_bootstrapping_amd.hetero import HeteroData
    g = HeteroData()
    g["ligand"].x = torch.zeros(n, 16, dtype=torch.long)
    g["ligand", "lig_bond", "ligand"].edge_index = torch.as_tensor(edge_index, dtype=torch.long)
    return g


def test_transformation_mask_matches_the_reference_on_1a0q(g17):
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    g = pm.get_ligand(SDF, "1a0q")
    assert np.array_equal(g["ligand", "ligand"].edge_index.numpy(), g17["sdf_edge_index"])
    assert np.array_equal(g["ligand"].edge_mask.numpy(), g17["sdf_mask_edges"])
    assert g["ligand"].mask_rotate.dtype == bool and np.array_equal(g["ligand"].mask_rotate, g17["sdf_mask_rotate"])
    assert int(g["ligand"].edge_mask.sum()) == 11 and g["ligand"].pos.shape == (23, 3)
    assert np.array_equal(g["ligand"].x[:, 0].numpy(), g17["sdf_z"] - 1)
    assert np.allclose(g["ligand"].orig_pos, g17["sdf_pos"]) and np.allclose(g["ligand"].pos.numpy(), g17["sdf_pos"], atol=1e-5)


def test_transformation_mask_matches_the_reference_on_random_graphs(g17):
    from confidence_bootstrapping_amd.torsion import get_transformation_mask
    off, n_rot_total = g17["rand_offsets"], 0
    for k, n in enumerate(g17["rand_n"].tolist()):
        ei = g17["rand_edge_index"][:, off[k]:off[k + 1]]
        me, mr = get_transformation_mask(_graph(n, ei))
        assert np.array_equal(me, g17["rand_mask_edges"][off[k]:off[k + 1]]), (k, g17["rand_kinds"][k])
        want = np.unpackbits(g17[f"rand_mask_rotate_packed_{k}"], axis=1)[:, :n].astype(bool)
        assert mr.shape == want.shape and np.array_equal(mr, want), (k, g17["rand_kinds"][k])
        n_rot_total += int(me.sum())
    assert n_rot_total > 100                                   # the cases are not degenerate
    # what the cases exercise: a double-bond bridge is rotatable (no bond-order test), a bridge to a single atom is not
    me, mr = get_transformation_mask(_graph(6, [[0, 1, 1, 2, 2, 3, 3, 4, 4, 5], [1, 0, 2, 1, 3, 2, 4, 3, 5, 4]]))
    assert me.tolist() == [0, 0, 0, 1, 0, 1, 1, 0, 0, 0] and mr.sum(1).tolist() == [2, 3, 2]
    assert mr[1].tolist() == [1, 1, 1, 0, 0, 0]                # tie of the two halves: the component of the lowest-numbered atom


def _raw_sdf(path):
    lines = open(path).read().splitlines()
    na, nb = int(lines[3][:3]), int(lines[3][3:6])
    atoms = [(l[31:34].strip(), float(l[0:10]), float(l[10:20]), float(l[20:30])) for l in lines[4:4 + na]]
    bonds = [(int(l[0:3]) - 1, int(l[3:6]) - 1, int(l[6:9])) for l in lines[4 + na:4 + na + nb]]
    return atoms, bonds


def test_sdf_reader_keeps_the_file_order():
    from confidence_bootstrapping_amd.datasets.molfile import read_sdf, remove_hs
    atoms, bonds = _raw_sdf(SDF)
    mol = read_sdf(SDF)
    assert [a.GetSymbol() for a in mol.GetAtoms()] == [a[0] for a in atoms]
    assert np.array_equal(mol.GetConformer().GetPositions(), np.asarray([a[1:] for a in atoms]))
    assert [(b.GetBeginAtomIdx(), b.GetEndAtomIdx(), b.type) for b in mol.GetBonds()] == bonds
    heavy = [i for i, a in enumerate(atoms) if a[0] != "H"]
    remap = {a: k for k, a in enumerate(heavy)}
    noh = remove_hs(mol)
    assert noh.GetNumAtoms() == len(heavy) == noh.GetNumHeavyAtoms()
    assert [(b.GetBeginAtomIdx(), b.GetEndAtomIdx()) for b in noh.GetBonds()] == [(remap[a], remap[b]) for a, b, _ in bonds if a in remap and b in remap]
    assert np.array_equal(noh.GetConformer().GetPositions(), np.asarray([atoms[i][1:] for i in heavy]))


def test_mol2_reader_gives_the_same_heavy_atom_graph():
    """the MOL2 of 1a0q lists the same molecule with its own atom order and SYBYL's delocalised acid groups"""
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    import networkx as nx
    a = pm.read_molecule(SDF, sanitize=True, remove_hs=True)
    b = pm.read_molecule(MOL2, sanitize=True, remove_hs=True)
    assert a.GetNumAtoms() == b.GetNumAtoms() == 23 and len(a.GetBonds()) == len(b.GetBonds())

    def nxg(m):
        g = nx.Graph()
        for at in m.GetAtoms():
            g.add_node(at.GetIdx(), z=at.GetAtomicNum())
        g.add_edges_from((x.GetBeginAtomIdx(), x.GetEndAtomIdx()) for x in m.GetBonds())
        return g
    assert nx.is_isomorphic(nxg(a), nxg(b), node_match=lambda p, q: p["z"] == q["z"])
    # carboxylate and phosphonate of the MOL2: one double bond each, the other oxygen charged
    assert sorted(x.GetFormalCharge() for x in b.GetAtoms() if x.GetAtomicNum() == 8) == [-1, -1, 0, 0, 0, 0]
    assert sum(1 for x in b.GetBonds() if x.GetBondType() == "AROMATIC") == 6
    mol, problem = pm.read_sdf_or_mol2(SDF, MOL2)
    assert not problem and mol.GetNumAtoms() == 23
    mol, problem = pm.read_sdf_or_mol2(os.path.join(D, "missing.sdf"), MOL2)
    assert not problem and mol.GetNumAtoms() == 23
    assert pm.read_sdf_or_mol2(os.path.join(D, "missing.sdf"), os.path.join(D, "missing.mol2")) == (None, True)
    assert isinstance(pm.read_molecule("x.xyz"), ValueError)                     # the reference RETURNS the error object
    with pytest.raises(NotImplementedError):
        pm.read_molecule("ligand.pdbqt")


def test_atom_features_of_1a0q():
    """vocabulary indices of the columns that follow from the file alone, and the textbook perception of this ligand: a phenyl ester
    of a phosphonic acid with an amide and a carboxylic acid"""
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    g = pm.get_ligand(SDF, "1a0q")
    x = g["ligand"].x
    assert x.dtype == torch.long and x.shape == (23, 16)
    assert all(int(x[:, c].max()) < pm.lig_feature_dims[0][c] for c in range(16))
    atoms, bonds = _raw_sdf(SDF)
    heavy = [i for i, a in enumerate(atoms) if a[0] != "H"]
    nh = {i: 0 for i in heavy}
    deg = {i: 0 for i in heavy}
    for a, b, _ in bonds:
        for p, q in ((a, b), (b, a)):
            if atoms[p][0] != "H":
                deg[p] += 1
                nh[p] += atoms[q][0] == "H"
    assert x[:, 2].tolist() == [deg[i] for i in heavy]                         # total degree = neighbours incl. hydrogens
    assert x[:, 4].tolist() == x[:, 5].tolist() == [nh[i] for i in heavy]      # implicit valence = hydrogens, all removed
    assert (x[:, 3] == 5).all() and (x[:, 6] == 0).all()                       # neutral, no radicals
    ring6 = x[:, 13].bool()
    assert ring6.sum() == 6 and (x[ring6, 9] == 1).all() and (x[~ring6, 9] == 0).all() and x[:, [10, 11, 12, 14, 15]].sum() == 0
    assert (x[:, 8].bool() == ring6).all()                                     # the phenyl ring, and nothing else, is aromatic
    assert (x[ring6, 7] == 1).all()                                            # SP2
    ea = g["ligand", "ligand"].edge_attr
    assert ea.shape == (46, 4) and int(ea[:, 3].sum()) == 12 and int(ea[:, 1].sum()) == 6 and torch.equal(ea[0::2], ea[1::2])
    sym = [atoms[i][0] for i in heavy]
    assert x[sym.index("P"), 7] == 2 and x[sym.index("N"), 7] == 1             # P: SP3; the amide N: SP2 (conjugated)
    assert x[0, 1] in (1, 2)                                                   # C1 (P, N, C, H) is a stereocentre
    assert len(pm.LIG_FEATURE_SOURCES) == 11


MOLBLOCKS = {
    # indole-like bicycle + charged amine + H2 + a bridging hydrogen are built from atom / bond lists below
}


def _mol(symbols, bonds, charges=None, pos=None):
    from confidence_bootstrapping_amd.datasets.molfile import Atom, Bond, Mol, _Z_OF
    atoms = [Atom(i, _Z_OF[s.upper()], s, (charges or {}).get(i, 0)) for i, s in enumerate(symbols)]
    rng = np.random.default_rng(len(symbols))
    return Mol(atoms, [Bond(a, b, t) for a, b, t in bonds], rng.normal(size=(len(symbols), 3)) if pos is None else pos)


def test_perception_on_textbook_molecules():
    from confidence_bootstrapping_amd.datasets.molfile import perceive, remove_hs
    # pyridine (kekule form, hydrogens implicit): aromatic, N has no hydrogen
    m = perceive(_mol(list("NCCCCC"), [(0, 1, 2), (1, 2, 1), (2, 3, 2), (3, 4, 1), (4, 5, 2), (5, 0, 1)]))
    assert all(a.GetIsAromatic() for a in m.GetAtoms()) and [a.GetTotalNumHs() for a in m.GetAtoms()] == [0, 1, 1, 1, 1, 1]
    assert all(b.GetBondType() == "AROMATIC" for b in m.GetBonds())
    # pyrrole: the N-H lone pair completes the sextet
    m = perceive(_mol(list("NCCCC"), [(0, 1, 1), (1, 2, 2), (2, 3, 1), (3, 4, 2), (4, 0, 1)]))
    assert all(a.GetIsAromatic() for a in m.GetAtoms()) and m.GetAtoms()[0].GetTotalNumHs() == 1
    # cyclohexene and 2-pyranone's saturated cousin are not; cyclopentadiene is not (sp3 carbon)
    m = perceive(_mol(list("CCCCCC"), [(0, 1, 2), (1, 2, 1), (2, 3, 1), (3, 4, 1), (4, 5, 1), (5, 0, 1)]))
    assert not any(a.GetIsAromatic() for a in m.GetAtoms()) and [a.GetHybridization() for a in m.GetAtoms()][:3] == ["SP2", "SP2", "SP3"]
    m = perceive(_mol(list("CCCCC"), [(0, 1, 2), (1, 2, 1), (2, 3, 2), (3, 4, 1), (4, 0, 1)]))
    assert not any(a.GetIsAromatic() for a in m.GetAtoms())
    # 2-pyridone: the exocyclic C=O carbon gives no electron, the N-H two -> aromatic ring, the oxygen stays outside
    m = perceive(_mol(list("NCCCCCO"), [(0, 1, 1), (1, 2, 2), (2, 3, 1), (3, 4, 2), (4, 5, 1), (5, 0, 1), (5, 6, 2)]))
    assert [a.GetIsAromatic() for a in m.GetAtoms()] == [True] * 6 + [False]
    # naphthalene (fused): both rings; indole: both rings, the N is in the five-ring only
    naph = [(0, 1, 2), (1, 2, 1), (2, 3, 2), (3, 4, 1), (4, 5, 2), (5, 0, 1), (4, 6, 1), (6, 7, 2), (7, 8, 1), (8, 9, 2), (9, 5, 1)]
    m = perceive(_mol(list("C" * 10), naph))
    assert all(a.GetIsAromatic() for a in m.GetAtoms())
    ri = m.GetRingInfo()
    assert ri.NumRings() == 2 and ri.NumAtomRings(4) == 2 and ri.NumAtomRings(0) == 1 and ri.IsAtomInRingOfSize(4, 6) and not ri.IsAtomInRingOfSize(4, 5)
    # azulene: neither ring alone is 4n+2, the ten-atom envelope is; the shared bond stays single
    azu = [(0, 1, 2), (1, 2, 1), (2, 3, 2), (3, 4, 1), (4, 0, 1), (3, 5, 1), (5, 6, 2), (6, 7, 1), (7, 8, 2), (8, 9, 1), (9, 4, 2)]
    m = perceive(_mol(list("C" * 10), azu))
    assert all(a.GetIsAromatic() for a in m.GetAtoms())
    assert [b.GetBondType() for b in m.GetBonds() if {b.GetBeginAtomIdx(), b.GetEndAtomIdx()} == {3, 4}] == ["SINGLE"]
    # hybridisation: acetonitrile C#N, CO2-, ammonium
    m = perceive(_mol(list("CCN"), [(0, 1, 1), (1, 2, 3)]))
    assert [a.GetHybridization() for a in m.GetAtoms()] == ["SP3", "SP", "SP"] and m.GetAtoms()[0].GetTotalNumHs() == 3
    m = perceive(_mol(list("CCOO"), [(0, 1, 1), (1, 2, 2), (1, 3, 1)], charges={3: -1}))
    assert [a.GetHybridization() for a in m.GetAtoms()] == ["SP3", "SP2", "SP2", "SP2"] and m.GetAtoms()[3].GetTotalNumHs() == 0
    m = perceive(_mol(list("CN"), [(0, 1, 1)], charges={1: 1}))
    assert m.GetAtoms()[1].GetTotalNumHs() == 3 and m.GetAtoms()[1].GetHybridization() == "SP3"
    # hydrogen removal: H2 stays, a bridging hydrogen stays, an ordinary one goes and is counted
    m = remove_hs(perceive(_mol(list("HH"), [(0, 1, 1)])))
    assert m.GetNumAtoms() == 2
    m = remove_hs(perceive(_mol(["B", "H", "B", "H"], [(0, 1, 1), (1, 2, 1), (2, 3, 1)])))
    assert [a.GetSymbol() for a in m.GetAtoms()] == ["B", "H", "B"]
    # chirality: mirror images get opposite tags, a CH2 gets none, and removing a hydrogen that is not the last bond flips the parity
    tet = np.array([[0, 0, 0], [1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], dtype=float)
    sy, bd = ["C", "F", "Cl", "Br", "H"], [(0, 1, 1), (0, 2, 1), (0, 3, 1), (0, 4, 1)]
    a = perceive(_mol(sy, bd, pos=tet)).GetAtoms()[0].GetChiralTag()
    b = perceive(_mol(sy, bd, pos=tet * np.array([1, 1, -1.0]))).GetAtoms()[0].GetChiralTag()
    assert {a, b} == {"CHI_TETRAHEDRAL_CW", "CHI_TETRAHEDRAL_CCW"}
    assert perceive(_mol(["C", "F", "Cl", "H", "H"], bd, pos=tet)).GetAtoms()[0].GetChiralTag() == "CHI_UNSPECIFIED"
    last = remove_hs(perceive(_mol(sy, bd, pos=tet))).GetAtoms()[0].GetChiralTag()                       # H is the last bond: no flip
    sy2, pos2 = ["C", "H", "F", "Cl", "Br"], tet[[0, 4, 1, 2, 3]]
    first = remove_hs(perceive(_mol(sy2, bd, pos=pos2)))                                                  # H is the first of four: 3 swaps
    assert last == a and first.GetAtoms()[0].GetChiralTag() == a                                          # same molecule, same handedness


def test_evaluation_takes_the_file_molecule():
    """get_symmetry_rmsd on the molecule read from the SDF (what inference.py hands it): identity and a swap of the two acid oxygens"""
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    from confidence_bootstrapping_amd.molecules_utils import _graph_of
    g = pm.get_ligand(SDF, "1a0q")
    nums, am = _graph_of(g.mol)
    assert nums.shape == (23,) and am.sum() == 46


@pytest.mark.gpu
def test_complex_from_its_three_files_runs_through_both_engines(tmp_path, g17):
    """data/1a0q: protein PDB + ligand SDF (+ MOL2) -> get_complex -> score engine (one reverse step) and confidence engine."""
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model
    from confidence_bootstrapping_amd.engine import make_steps
    dev = torch.device("cuda:0")
    pdb = tmp_path / "1a0q_protein_processed.pdb"
    pdb.write_bytes(gzip.open(os.path.join(D, "1a0q_protein_processed.pdb.gz")).read())
    rng = np.random.default_rng(0)
    g = pm.get_complex(str(pdb), SDF, "1a0q", dev, mol2_file=MOL2, lm_embeddings=[rng.normal(0, 0.5, size=(416, 1280)).astype(np.float32)])
    assert g["receptor"].x.shape == (416, 1281) and g["ligand"].x.shape == (23, 16) and g["atom"].pos.shape[0] == 3181
    assert np.array_equal(g["ligand"].edge_mask.numpy(), g17["sdf_mask_edges"]) and np.array_equal(g["ligand"].mask_rotate, g17["sdf_mask_rotate"])
    assert float(g["receptor"].pos.mean(0).abs().max()) < 1e-4
    assert np.allclose(g["ligand"].pos.numpy() + g.original_center.numpy(), g17["sdf_pos"], atol=1e-4)
    smodel, sargs = make_score_model(device=dev, seed=0)
    cmodel, cargs = make_confidence_model(device=dev, seed=5)
    eng = smodel.engine()
    eng.set_complex(g)
    assert eng.R == 11
    pos = g["ligand"].pos[None].repeat(2, 1, 1).to(dev)
    tr, rot, tor = eng.score(pos, make_steps(np.array([0.5]), sargs, smodel.timestep_emb_func)[0])
    assert tor.numel() == 22 and torch.isfinite(tr).all() and torch.isfinite(rot).all() and torch.isfinite(tor).all()
    ceng = cmodel.engine(max_batch=2)
    ceng.set_complex(g)
    conf, _ = ceng.score(pos, cargs.crop_beyond)
    assert torch.isfinite(conf).all()
