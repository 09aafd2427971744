"""Receptor featurisation (`confidence_bootstrapping_amd/datasets/process_mols.py`) against
tests/golden/g15_featurise_1a0q.npz, which oracle/make_golden_featurise.py produced by RUNNING the reference's own
`new_extract_receptor_structure`, `get_moad_atom_feats` and `get_chi_angles` on the residues of data/1a0q.

CPU: the node stores (residue / atom features, positions, side-chain vectors, atom -> residue map), the PDB reader's rules.
GPU: the neighbour graphs built by the HIP kernels (cbd_knn_graph / cbd_radius_neighbors) -- exact edge lists for the kNN branch of
the shipped ymls; for the cutoff branch the reference thresholds distances from torch.cdist's matmul formulation in fp32 (error
~1e-4 A), so pairs closer than 2e-3 A to the cutoff (or to the rank-`max_neighbors` distance) may differ and nothing else."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g15_featurise_1a0q.npz")


@pytest.fixture(scope="module")
def g15():
    z = dict(np.load(GOLD))
    milli = z.pop("coords_milli")            # exact integer milli-Angstrom; INT32_MIN marks an absent slot
    z["coords"] = np.where(milli == np.iinfo(np.int32).min, np.nan, milli / 1000.0)
    return z


def _graph(g15, **kw):
    from confidence_bootstrapping_amd.hetero import HeteroData
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    g = HeteroData()
    rng = np.random.default_rng(0)
    seq = "".join(g15["seq"].tolist())
    lm = [rng.normal(0, 0.5, size=(len(seq), 8)).astype(np.float32)]
    pm.new_extract_receptor_structure(seq, g15["coords"], g, neighbor_cutoff=15.0, max_neighbors=24, lm_embeddings=lm,
                                      all_atoms=True, atom_cutoff=5, atom_max_neighbors=8, **kw)
    return g


def test_node_stores_match_the_reference(g15):
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    seq = "".join(g15["seq"].tolist())
    chi = pm.get_chi_angles(g15["coords"], seq)
    assert np.array_equal(np.isnan(chi), np.isnan(g15["chi"])) and np.allclose(np.nan_to_num(chi), np.nan_to_num(g15["chi"]), atol=1e-9)
    g = _graph(g15, knn_only_graph=True, rec_edge_index=g15["knn_rec_edge_index"], atom_edge_index=g15["knn_atom_edge_index"])
    assert np.array_equal(g["receptor"].x.numpy(), g15["rec_x"])
    assert np.array_equal(g["receptor"].pos.numpy(), g15["rec_pos"])
    a, b = g["receptor"].side_chain_vecs.numpy(), g15["side_chain_vecs"]
    assert np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(np.nan_to_num(a), np.nan_to_num(b), atol=1e-6)
    assert np.array_equal(g["atom"].x.numpy(), g15["atom_x"].astype(np.float32))
    assert np.array_equal(g["atom"].pos.numpy(), g15["atom_pos"])
    assert np.array_equal(g["atom", "atom_rec_contact", "receptor"].edge_index.numpy(), g15["atom_res"])
    assert g["receptor", "rec_contact", "receptor"].edge_index.dtype == torch.long
    # the schema the engines consume: the confidence engine wants atom -> residue as arange in row 0
    assert np.array_equal(g["atom", "receptor"].edge_index[0].numpy(), np.arange(g15["atom_x"].shape[0]))


def test_no_silent_cpu_neighbour_search(g15):
    with pytest.raises(RuntimeError, match="no device"):
        _graph(g15, knn_only_graph=True)
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    with pytest.raises(RuntimeError, match="MI355X"):
        pm.knn_graph(torch.zeros(5, 3), 2)


def test_parse_pdb_rules(tmp_path):
    from confidence_bootstrapping_amd.datasets import process_mols as pm

    def atom(rec, serial, name, alt, resn, chain, resi, x, y, z, el):
        return f"{rec:<6}{serial:>5} {name:<4}{alt}{resn:>3} {chain}{resi:>4}    {x:8.3f}{y:8.3f}{z:8.3f}  1.00  0.00          {el:>2}\n"
    lines = [atom("ATOM", 1, " N", " ", "GLY", "A", 1, 0, 0, 0, "N"), atom("ATOM", 2, " CA", " ", "GLY", "A", 1, 1.4, 0, 0, "C"),
             atom("ATOM", 3, " C", " ", "GLY", "A", 1, 2, 1.2, 0, "C"), atom("ATOM", 4, " O", " ", "GLY", "A", 1, 3, 1.2, 0, "O"),
             atom("ATOM", 5, " N", " ", "SER", "A", 2, 4, 0, 0, "N"), atom("ATOM", 6, " CA", "A", "SER", "A", 2, 5, 0, 0, "C"),
             atom("ATOM", 7, " CA", "B", "SER", "A", 2, 9, 9, 9, "C"), atom("ATOM", 8, " OG", " ", "SER", "A", 2, 6, 1, 1, "O"),
             atom("ATOM", 9, " H", " ", "SER", "A", 2, 6, 2, 1, "H"),
             atom("HETATM", 10, " N", " ", "MSE", "A", 3, 7, 0, 0, "N"), atom("HETATM", 11, " CA", " ", "MSE", "A", 3, 8, 0, 0, "C"),
             atom("HETATM", 12, "SE", " ", "MSE", "A", 3, 8, 2, 0, "SE"),
             atom("HETATM", 13, "CA", " ", " CA", "A", 4, 20, 0, 0, "CA"),                 # a calcium ion is not a C-alpha
             atom("ATOM", 14, " N", " ", "ALA", "B", 1, 0, 5, 0, "N"),                      # residue without CA: dropped
             atom("ATOM", 15, " CA", " ", "XYZ", "B", 2, 1, 5, 0, "C")]                    # unknown residue name -> 'X', backbone slots only
    p = tmp_path / "t.pdb"
    p.write_text("".join(lines) + "ENDMDL\n" + atom("ATOM", 16, " CA", " ", "GLY", "C", 1, 0, 0, 9, "C"))
    pdb = pm.parse_pdb(str(p))
    assert pdb.seq == "GSMX" and pdb.coords.shape == (4, 14, 3)
    assert np.allclose(pdb.coords[1, 1], [5, 0, 0])                       # first alternate location
    assert np.allclose(pdb.coords[1, 5], [6, 1, 1]) and np.isnan(pdb.coords[1, 4]).all()      # OG present, CB missing
    assert np.isnan(pdb.coords[2, 6]).all()                               # MSE uses MET's slots: 'SD' is absent (the selenium is 'SE')
    assert pdb.chain_ids.tolist() == [0, 0, 0, 1]
    onehot = pm.get_onehot_sequence(pdb.seq)
    assert onehot[3, 7] == 1                                               # unknown -> GLY column (parse_chi.py:79)
    with pytest.raises(KeyError):
        pm.get_moad_atom_feats("X", pdb.coords[3])                        # the reference fails on such a residue in all-atom mode too
    # insertion codes are residues of their own, and a residue key that comes back later in the file starts a NEW residue (prody's
    # resindex; it is not merged into its first occurrence)
    ins = [atom("ATOM", 1, " CA", " ", "GLY", "A", "52", 0, 0, 0, "C"), atom("ATOM", 2, " CA", " ", "SER", "A", "52A", 3.8, 0, 0, "C"),
           atom("ATOM", 3, " CA", " ", "ALA", "A", "53", 7.6, 0, 0, "C"), atom("ATOM", 4, " CB", " ", "GLY", "A", "52", 1, 1, 1, "C"),
           atom("ATOM", 5, " CA", " ", "GLY", "A", "52", 2, 2, 2, "C")]
    ins = [l[:22] + f"{l[22:27].strip():>4} "[:5] + l[27:] if not l[22:27].strip()[-1].isalpha() else l[:22] + f"{l[22:27].strip():>5}" + l[27:] for l in ins]
    q = tmp_path / "ins.pdb"
    q.write_text("".join(ins))
    pdb = pm.parse_pdb(str(q))
    assert pdb.seq == "GSAG" and np.allclose(pdb.coords[3, 1], [2, 2, 2]) and np.allclose(pdb.coords[0, 1], [0, 0, 0])


def _pairs(ei):
    return set(map(tuple, np.asarray(ei).T.tolist()))


@pytest.mark.gpu
def test_knn_graphs_on_the_gpu_match_the_reference(g15):
    dev = torch.device("cuda:0")
    g = _graph(g15, knn_only_graph=True, device=dev)
    assert np.array_equal(g["receptor", "receptor"].edge_index.numpy(), g15["knn_rec_edge_index"])
    assert np.array_equal(g["atom", "atom"].edge_index.numpy(), g15["knn_atom_edge_index"])
    # degenerate sizes: k >= n, n = 1, n = 0
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    p = torch.tensor([[0.0, 0, 0], [1, 0, 0], [3, 0, 0]], device=dev)
    assert pm.knn_graph(p, 5).cpu().tolist() == [[1, 2, 0, 2, 1, 0], [0, 0, 1, 1, 2, 2]]
    assert pm.knn_graph(p[:1], 3).shape == (2, 0) and pm.knn_graph(p[:0], 3).shape == (2, 0)


@pytest.mark.gpu
def test_cutoff_graphs_on_the_gpu_match_the_reference(g15):
    dev = torch.device("cuda:0")
    g = _graph(g15, knn_only_graph=False, device=dev)

    def check(got, want, pos, cutoff, cap):
        a, b = _pairs(got), _pairs(want)
        d = torch.cdist(torch.as_tensor(pos).double(), torch.as_tensor(pos).double())
        kth = torch.sort(d, dim=1)[0][:, min(cap, d.shape[0] - 1)]          # distance of the cap-th nearest other node (column 0 is self)
        for (j, i) in a ^ b:
            dij = float(d[i, j])
            assert abs(dij - cutoff) < 2e-3 or abs(dij - float(kth[i])) < 2e-3, (i, j, dij)
        assert len(a ^ b) <= 0.002 * len(b)
        return len(a ^ b)
    check(g["receptor", "receptor"].edge_index.numpy(), g15["cut_rec_edge_index"], g15["rec_pos"], 15.0, 24)
    check(g["atom", "atom"].edge_index.numpy(), g15["cut_atom_edge_index"], g15["atom_pos"], 5.0, 8)
    # tight cutoff: most residues have no neighbour within 4.2 A besides their chain neighbours; one without any gets its nearest
    from confidence_bootstrapping_amd.datasets import process_mols as pm
    tight = pm.cutoff_graph(torch.from_numpy(g15["rec_pos"]).to(dev), 4.2, 3).cpu().numpy()
    check(tight, g15["tight_rec_edge_index"], g15["rec_pos"], 4.2, 3)
    assert set(tight[1].tolist()) == set(range(g15["rec_pos"].shape[0]))      # every centre has at least one edge


@pytest.mark.gpu
def test_featurised_receptor_runs_through_both_engines(g15):
    """PDB-derived receptor stores (this module) + the 1a0q ligand of the C1 fixture -> score engine and confidence engine."""
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model
    from confidence_bootstrapping_amd.engine import make_steps
    from tests.helpers import load_c1_complex
    dev = torch.device("cuda:0")
    c1 = load_c1_complex()
    g = _graph(g15, knn_only_graph=True, device=dev)
    rng = np.random.default_rng(0)
    center = g["receptor"].pos.mean(0, keepdim=True)
    c1["receptor"].x = torch.cat([g["receptor"].x[:, :1], torch.from_numpy(rng.normal(0, 0.5, size=(416, 1280)).astype(np.float32))], 1)
    c1["receptor"].pos = g["receptor"].pos - center
    c1["receptor", "receptor"].edge_index = g["receptor", "receptor"].edge_index
    c1["atom"].x, c1["atom"].pos = g["atom"].x, g["atom"].pos - center
    c1["atom", "atom_contact", "atom"].edge_index = g["atom", "atom"].edge_index
    c1["atom", "atom_rec_contact", "receptor"].edge_index = g["atom", "receptor"].edge_index
    smodel, sargs = make_score_model(device=dev, seed=0)
    cmodel, cargs = make_confidence_model(device=dev, seed=5)
    eng = smodel.engine()
    eng.set_complex(c1)
    pos = c1["ligand"].pos[None].repeat(2, 1, 1).to(dev)
    tr, rot, tor = eng.score(pos, make_steps(np.array([0.5]), sargs, smodel.timestep_emb_func)[0])
    assert torch.isfinite(tr).all() and torch.isfinite(rot).all() and torch.isfinite(tor).all()
    ceng = cmodel.engine(max_batch=2)
    ceng.set_complex(c1)
    conf, _ = ceng.score(pos, cargs.crop_beyond)
    assert torch.isfinite(conf).all()
