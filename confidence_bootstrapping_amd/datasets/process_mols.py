"""Receptor-side featurisation of the MOAD / DockGen path WITHOUT prody / rdkit / torch_cluster (the part this
image allows): from a protein PDB file to the graph stores the score model and the all-atom confidence model read.

Reference: datasets/process_mols.py:414-526 (`moad_extract_receptor_structure` -> `new_extract_receptor_structure`),
:532-564 (`get_moad_atom_feats`), datasets/parse_chi.py:60-107 (`get_coords`, `get_onehot_sequence`, `get_chi_angles`),
datasets/moad.py:394-419 (`get_receptor`: centring on the C-alpha centroid, `original_center`).  Same function names and argument
meaning; what differs:
  * the PDB file is read by the minimal parser below (`parse_pdb`) instead of prody -- ATOM / HETATM records of residues that own a
    CA atom, first alternate location, file order = prody's `resindex` order;
  * the two k-nearest-neighbour graphs (C-alpha kNN-24, heavy-atom kNN-8: `knn_graph` of torch_cluster, rows [neighbour; centre],
    centres ascending, neighbours by increasing distance) and the cutoff graphs of the non-kNN branch are built by HIP kernels
    (`cbd_knn_graph`, `cbd_radius_neighbors`; csrc/featurise.hip) when `device` is a GPU -- there is no CPU fallback inside the
    package: without a GPU pass the edge lists in (`rec_edge_index=` / `atom_edge_index=`) or call with `device=None` to get the
    node stores only;
  * language-model embeddings are passed in as arrays (ESM weights are not in the image).
The ligand side (bottom of this file): `read_molecule` / `read_sdf_or_mol2` (SDF V2000 and MOL2 readers, datasets/molfile.py),
`lig_atom_featurizer`, `get_lig_graph`, `get_lig_graph_with_matching` without the rdkit conformer matching (`matching=False`, what
inference on given poses uses), `get_transformation_mask` (torsion.py) and `get_complex` = receptor + ligand from the three files of a
PDBBind-style directory.  Which of the 16 atom features are exact and which are rdkit perception restated: molfile.LIG_FEATURE_SOURCES.

Vocabularies and the 14-slot heavy-atom layout are the reference's data contract (process_mols.py:60-123, constants.py:78-98): a
checkpoint's embedding tables are indexed by them.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import numpy as np
import torch

from ..hetero import HeteroData

# ---- vocabularies (index = embedding row; 'misc' is always the last entry) ---------------------------------------------------------
AMINO_ACIDS = ("ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL HIP HIE TPO HID LEV MEU PTR GLV CYT "
               "SEP HIZ CYM GLM ASQ TYS CYX GLZ misc").split()
ATOM_TYPE_2 = "C* CA CB CD CE CG CH CZ N* ND NE NH NZ O* OD OE OG OH OX S* SD SG misc".split()
ATOM_TYPE_3 = ("C CA CB CD CD1 CD2 CE CE1 CE2 CE3 CG CG1 CG2 CH2 CZ CZ2 CZ3 N ND1 ND2 NE NE1 NE2 NH1 NH2 NZ O OD1 OD2 OE1 OE2 OG OG1 "
               "OH OXT SD SG misc").split()
allowable_features = {"possible_amino_acids": AMINO_ACIDS, "possible_atomic_num_list": list(range(1, 119)) + ["misc"],
                      "possible_atom_type_2": ATOM_TYPE_2, "possible_atom_type_3": ATOM_TYPE_3}
rec_atom_feature_dims = ([len(AMINO_ACIDS), 119, len(ATOM_TYPE_2), len(ATOM_TYPE_3)], 0)
rec_residue_feature_dims = ([len(AMINO_ACIDS)], 0)

_STD = dict(zip("ALA ARG ASN ASP CYS GLN GLU GLY HIS ILE LEU LYS MET PHE PRO SER THR TRP TYR VAL".split(), "ARNDCQEGHILKMFPSTWYV"))
aa_short2long = {v: k for k, v in _STD.items()}
aa_long2short = dict(_STD, MSE="M")                  # slot layout of a residue (constants.py:21-22; anything else -> 'X': backbone only)
# one-letter code of the sequence = prody's `pdb.ca.getSequence()` in the reference (process_mols.py:418).  prody is an un-vendored
# dependency and absent here; its published residue map is the 20 standard names plus the modified residues below, anything else 'X'
_SEQ_MAP = dict(_STD, MSE="M", PTR="Y", TPO="T", SEP="S", CSO="C", HSD="H", HSP="H", HSE="H", ASX="B", GLX="Z", SEC="U", PYL="O", XLE="J")
_AA_IDX = {a: i for i, a in enumerate("ARNDCEQGHILKMFPSTWYV")}        # aa_name2aa_idx (constants.py:1-3: GLU 5, GLN 6)
_AA_IDX_INV = {i: a for a, i in _AA_IDX.items()}
# side-chain heavy atoms in slot order (slots 0-3 are N, CA, C, O)
_SIDE = {"G": "", "A": "CB", "S": "CB OG", "C": "CB SG", "T": "CB OG1 CG2", "P": "CB CG CD", "V": "CB CG1 CG2", "M": "CB CG SD CE",
         "N": "CB CG OD1 ND2", "I": "CB CG1 CG2 CD1", "L": "CB CG CD1 CD2", "D": "CB CG OD1 OD2", "E": "CB CG CD OE1 OE2",
         "K": "CB CG CD CE NZ", "Q": "CB CG CD OE1 NE2", "H": "CB CG ND1 CD2 CE1 NE2", "F": "CB CG CD1 CD2 CE1 CE2 CZ",
         "R": "CB CG CD NE CZ NH1 NH2", "Y": "CB CG CD1 CD2 CE1 CE2 CZ OH", "W": "CB CG CD1 CD2 CE2 CE3 NE1 CZ2 CZ3 CH2", "X": ""}
atom_order = {k: ["N", "CA", "C", "O"] + v.split() for k, v in _SIDE.items()}
# chi angles: the side-chain path N-CA-CB-(gamma)-(delta)-(epsilon)-(zeta), chi_n = four consecutive atoms of it
_CHI_PATH = {"C": "SG", "D": "CG OD1", "E": "CG CD OE1", "F": "CG CD1", "H": "CG ND1", "I": "CG1 CD1", "K": "CG CD CE NZ", "L": "CG CD1",
             "M": "CG SD CE", "N": "CG OD1", "P": "CG CD", "Q": "CG CD OE1", "R": "CG CD NE CZ", "S": "OG", "T": "OG1", "V": "CG1",
             "W": "CG CD1", "Y": "CG CD1"}
_Z = {"C": 6, "N": 7, "O": 8, "S": 16}


def safe_index(lst, e):
    try:
        return lst.index(e)
    except ValueError:
        return len(lst) - 1


def _chi_slots():
    out = {}
    for aa, order in atom_order.items():
        rows = np.full((4, 4), np.nan)
        if aa in _CHI_PATH:
            path = ["N", "CA", "CB"] + _CHI_PATH[aa].split()
            for n in range(len(path) - 3):
                rows[n] = [order.index(x) for x in path[n:n + 4]]
        out[aa] = rows
    return out


_CHI_SLOTS = _chi_slots()


# ---- PDB file -> residues ---------------------------------------------------------------------------------------------------------
class ParsedPDB:
    """seq (one-letter string over the CA-bearing residues), coords [Nres, 14, 3] float64 with NaN for absent slots, chain ids."""

    def __init__(self, seq, coords, chain_ids, resnames):
        self.seq, self.coords, self.chain_ids, self.resnames = seq, coords, chain_ids, resnames


def parse_pdb(path) -> ParsedPDB:
    """Minimal reader for what `prody.parsePDB(path)` + `pdb.ca` + parse_chi.get_coords give the reference: every residue (in file
    order) that owns an atom named CA, its one-letter code (unknown names -> 'X'), and the coordinates of its heavy atoms in the
    14-slot layout.  First model, first alternate location; hydrogens never match a slot name."""
    residues, prev = [], None
    with open(path) as f:
        for line in f:
            rec = line[:6]
            if rec.startswith("ENDMDL"):
                break
            if rec not in ("ATOM  ", "HETATM"):
                continue
            alt = line[16]
            if alt not in (" ", "A"):
                continue
            name, resn = line[12:16].strip(), line[17:20].strip()
            if rec == "HETATM" and resn not in _SEQ_MAP:
                continue                                                # prody's `.ca` = C-alphas of PROTEIN residues (not a Ca2+ ion)
            key = (line[72:76].strip(), line[21], line[22:27], resn)    # segment, chain, residue number + insertion code, name
            if key != prev:                                             # prody's resindex: a new residue whenever the key changes,
                prev = key                                              # also when the same key comes back later in the file
                residues.append({"resn": resn, "key": key, "atoms": {}})
            atoms = residues[-1]["atoms"]
            if name not in atoms:
                atoms[name] = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
    residues = [r for r in residues if "CA" in r["atoms"]]
    n = len(residues)
    coords = np.full((n, 14, 3), np.nan)
    seq = []
    for i, r in enumerate(residues):
        seq.append(_SEQ_MAP.get(r["resn"], "X"))
        for j, nm in enumerate(atom_order[aa_long2short.get(r["resn"], "X")]):       # parse_chi.get_coords
            if nm in r["atoms"]:
                coords[i, j] = r["atoms"][nm]
    ids = [r["key"][0] + r["key"][1] for r in residues]
    uniq = sorted(set(ids))
    chain_ids = np.asarray([uniq.index(c) for c in ids], dtype=np.int64)
    return ParsedPDB("".join(seq), coords, chain_ids, [r["resn"] for r in residues])


def get_onehot_sequence(seq):
    onehot = np.zeros((len(seq), 20))
    for i, aa in enumerate(seq):
        onehot[i, _AA_IDX.get(aa, 7)] = 1            # unknown -> GLY (parse_chi.py:79)
    return onehot


def get_chi_angles(coords, seq, return_onehot=False):
    """[N, 4] side-chain dihedrals in degrees in [0, 360), NaN where undefined (parse_chi.py:86-113); the residue type of an unknown
    letter is GLY, as there."""
    onehot = get_onehot_sequence(seq)
    slots = np.stack([_CHI_SLOTS[_AA_IDX_INV[int(k)]] for k in np.argmax(onehot, axis=1)]) if len(seq) else np.zeros((0, 4, 4))
    mask = np.isnan(slots)
    idx = np.where(mask, 0, slots).astype(int)
    p = coords[np.arange(len(seq))[:, None, None], idx, :]
    p[mask] = np.nan
    p = p.reshape(-1, 4, 3)
    with np.errstate(invalid="ignore", divide="ignore"):
        b0, b1, b2 = p[:, 0] - p[:, 1], p[:, 1] - p[:, 2], p[:, 2] - p[:, 3]
        n1, n2 = np.cross(b0, b1), np.cross(b1, b2)
        m1 = np.cross(n1, b1 / np.linalg.norm(b1, axis=1, keepdims=True))
        deg = np.degrees(np.arctan2(np.sum(m1 * n2, axis=1), np.sum(n1 * n2, axis=1)))
        deg[deg < 0] += 360
    chi = deg.reshape(len(seq), 4)
    return (chi, onehot) if return_onehot else chi


def get_moad_atom_feats(res, coords):
    """[n_present_atoms, 4] categorical features of one residue's heavy atoms: residue type, atomic-number index, two- and
    three-character atom type (process_mols.py:532-564)."""
    feats = []
    res_long = aa_short2long[res]          # KeyError for a non-standard letter, as in the reference (the caller skips that receptor)
    order = atom_order[res]
    for i, c in enumerate(coords):
        if np.any(np.isnan(c)):
            continue
        f = [safe_index(AMINO_ACIDS, res_long)]
        if i >= len(order):
            f += [118, len(ATOM_TYPE_2) - 1, len(ATOM_TYPE_3) - 1]
        else:
            nm = order[i]
            f += [safe_index(allowable_features["possible_atomic_num_list"], _Z.get(nm[:1], -1)),
                  safe_index(ATOM_TYPE_2, (nm + "*")[:2]), safe_index(ATOM_TYPE_3, nm)]
        feats.append(f)
    return np.asarray(feats, dtype=np.float64).reshape(-1, 4)


# ---- neighbour graphs on the GPU --------------------------------------------------------------------------------------------------
def _lib():
    from ..engine import load_library
    return load_library()


def knn_graph(pos: torch.Tensor, k: int) -> torch.Tensor:
    """torch_cluster.knn_graph(pos, k) for one example (loop=False, flow='source_to_target'): [2, N * min(k, N-1)] int64,
    row 0 = neighbour, row 1 = centre; centres ascending, the neighbours of a centre by increasing distance (ties: lower index).
    `pos` must live on the GPU (cbd_knn_graph, csrc/featurise.hip)."""
    if not pos.is_cuda:
        raise RuntimeError("knn_graph runs on the MI355X (cbd_knn_graph); there is no CPU path in the package")
    pos = pos.float().contiguous()
    n = pos.shape[0]
    kk = max(0, min(int(k), n - 1))
    out = torch.empty(n, max(kk, 1), dtype=torch.int32, device=pos.device)
    lib = _lib()
    with torch.cuda.device(pos.device):
        rc = lib.cbd_knn_graph(n, kk, C.c_void_p(pos.data_ptr()), C.c_void_p(out.data_ptr()),
                               C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cbdock error {rc}: {lib.cbd_last_error().decode()}")
    if kk == 0:
        return torch.zeros(2, 0, dtype=torch.long, device=pos.device)
    centre = torch.arange(n, device=pos.device).repeat_interleave(kk)
    return torch.stack([out[:, :kk].reshape(-1).long(), centre])


def cutoff_graph(pos: torch.Tensor, cutoff: float, max_neighbors: Optional[int]) -> torch.Tensor:
    """The non-kNN branch of new_extract_receptor_structure (process_mols.py:461-479): for every centre i the nodes closer than
    `cutoff` in index order; if they are more than `max_neighbors`, the `max_neighbors` nearest by increasing distance instead; a
    centre without any gets its single nearest node.  Rows [neighbour; centre].  GPU only (cbd_radius_neighbors)."""
    if not pos.is_cuda:
        raise RuntimeError("cutoff_graph runs on the MI355X (cbd_radius_neighbors); there is no CPU path in the package")
    pos = pos.float().contiguous()
    n = pos.shape[0]
    cap = max(1, min(int(max_neighbors) if max_neighbors else 1000, n - 1))
    idx = torch.empty(n, cap, dtype=torch.int32, device=pos.device)
    cnt = torch.empty(n, dtype=torch.int32, device=pos.device)
    lib = _lib()
    with torch.cuda.device(pos.device):
        rc = lib.cbd_radius_neighbors(n, float(cutoff), cap, C.c_void_p(pos.data_ptr()), C.c_void_p(idx.data_ptr()),
                                      C.c_void_p(cnt.data_ptr()), C.c_void_p(torch.cuda.current_stream(pos.device).cuda_stream))
    if rc != 0:
        raise RuntimeError(f"cbdock error {rc}: {lib.cbd_last_error().decode()}")
    keep = torch.arange(cap, device=pos.device)[None, :] < cnt[:, None]
    centre = torch.arange(n, device=pos.device)[:, None].expand(n, cap)
    return torch.stack([idx[keep].long(), centre[keep]])


# ---- the graph stores -------------------------------------------------------------------------------------------------------------
def new_extract_receptor_structure(seq, all_coords, complex_graph, neighbor_cutoff=20, max_neighbors=None, lm_embeddings=None,
                                   knn_only_graph=False, all_atoms=False, atom_cutoff=None, atom_max_neighbors=None, device=None,
                                   rec_edge_index=None, atom_edge_index=None):
    """Fills complex_graph['receptor'] (x = [residue type | LM embedding], pos, side_chain_vecs), its rec_contact edges and, with
    all_atoms, the 'atom' stores (x [Na,4], pos, atom_contact edges, atom_rec_contact map) -- reference process_mols.py:448-526.
    `device`: GPU on which the neighbour graphs are built; None: only if both edge lists are passed in."""
    all_coords = np.asarray(all_coords, dtype=np.float64)
    chi, _ = get_chi_angles(all_coords, seq, return_onehot=True)
    n_rel, c_rel = all_coords[:, 0] - all_coords[:, 1], all_coords[:, 2] - all_coords[:, 1]
    side_chain_vecs = torch.from_numpy(np.concatenate([chi / 360, n_rel, c_rel], axis=1))
    coords = torch.tensor(all_coords[:, 1, :], dtype=torch.float)
    if len(coords) > 3000:
        raise ValueError(f"The receptor is too large {len(coords)}")

    def graph(p, knn_k, cutoff, cap):
        if device is None:
            raise RuntimeError("no device given: pass the edge lists in, the package has no CPU neighbour search")
        pd = p.to(device)
        return (knn_graph(pd, knn_k) if knn_only_graph else cutoff_graph(pd, cutoff, cap)).cpu()

    if rec_edge_index is None:
        rec_edge_index = graph(coords, max_neighbors if max_neighbors else 32, neighbor_cutoff, max_neighbors)
    names = [aa_short2long.get(a, "misc") for a in seq]
    node_feat = torch.tensor([[safe_index(AMINO_ACIDS, r)] for r in names], dtype=torch.float32)
    if lm_embeddings is not None:
        lm = torch.as_tensor(np.concatenate(lm_embeddings, axis=0) if isinstance(lm_embeddings, (list, tuple)) else lm_embeddings)
        node_feat = torch.cat([node_feat, lm.float()], dim=1)
    complex_graph["receptor"].x = node_feat
    complex_graph["receptor"].pos = coords
    complex_graph["receptor"].side_chain_vecs = side_chain_vecs.float()
    complex_graph["receptor", "rec_contact", "receptor"].edge_index = torch.as_tensor(rec_edge_index).long()
    if all_atoms:
        flat = all_coords.reshape(-1, 3)
        atom_coords = torch.from_numpy(flat[~np.any(np.isnan(flat), axis=1)]).float()
        if atom_edge_index is None:
            atom_edge_index = graph(atom_coords, atom_max_neighbors if atom_max_neighbors else 1000, atom_cutoff, atom_max_neighbors)
        feats = [get_moad_atom_feats(res, all_coords[i]) for i, res in enumerate(seq)]
        atom_feat = torch.from_numpy(np.concatenate(feats, axis=0)).float()
        res_of = np.concatenate([np.zeros(len(f)) + i for i, f in enumerate(feats)])
        complex_graph["atom"].x = atom_feat
        complex_graph["atom"].pos = atom_coords
        assert len(atom_feat) == len(atom_coords)
        complex_graph["atom", "atom_contact", "atom"].edge_index = torch.as_tensor(atom_edge_index).long()
        complex_graph["atom", "atom_rec_contact", "receptor"].edge_index = torch.from_numpy(np.stack([np.arange(len(atom_feat)), res_of])).long()


def moad_extract_receptor_structure(path, complex_graph, neighbor_cutoff=20, max_neighbors=None, sequences_to_embeddings=None,
                                    knn_only_graph=False, lm_embeddings=None, all_atoms=False, atom_cutoff=None, atom_max_neighbors=None,
                                    device=None):
    """PDB file -> receptor stores (reference process_mols.py:414-445): per-chain sequences (for the LM-embedding look-up),
    chain ids, then new_extract_receptor_structure."""
    pdb = parse_pdb(path)
    onehot = get_onehot_sequence(pdb.seq)
    sequences, embs = [], [] if sequences_to_embeddings is not None else lm_embeddings
    for cid in np.unique(pdb.chain_ids):
        s = "".join(_AA_IDX_INV[int(k)] for k in np.argmax(onehot[pdb.chain_ids == cid], axis=1))
        sequences.append(s)
        if sequences_to_embeddings is not None:
            embs.append(sequences_to_embeddings[s])
    complex_graph["receptor"].sequence = sequences
    complex_graph["receptor"].chain_ids = torch.from_numpy(pdb.chain_ids).long()
    new_extract_receptor_structure(pdb.seq, pdb.coords, complex_graph, neighbor_cutoff=neighbor_cutoff, max_neighbors=max_neighbors,
                                   lm_embeddings=embs, knn_only_graph=knn_only_graph, all_atoms=all_atoms, atom_cutoff=atom_cutoff,
                                   atom_max_neighbors=atom_max_neighbors, device=device)
    return pdb


def get_receptor(path, name, device, receptor_radius=15.0, c_alpha_max_neighbors=24, knn_only_graph=True, all_atoms=True, atom_radius=5,
                 atom_max_neighbors=8, sequences_to_embeddings=None, lm_embeddings=None) -> HeteroData:
    """One receptor graph as `MOAD.get_receptor` builds it (datasets/moad.py:394-419; defaults = the shipped ymls): stores centred on
    the C-alpha centroid, `original_center` kept."""
    g = HeteroData()
    g.receptor_name = name
    moad_extract_receptor_structure(path, g, neighbor_cutoff=receptor_radius, max_neighbors=c_alpha_max_neighbors,
                                    sequences_to_embeddings=sequences_to_embeddings, knn_only_graph=knn_only_graph,
                                    lm_embeddings=lm_embeddings, all_atoms=all_atoms, atom_cutoff=atom_radius,
                                    atom_max_neighbors=atom_max_neighbors, device=device)
    center = torch.mean(g["receptor"].pos, dim=0, keepdim=True)
    g["receptor"].pos = g["receptor"].pos - center
    if all_atoms:
        g["atom"].pos = g["atom"].pos - center
    g.original_center = center
    return g


# ---- ligand side ------------------------------------------------------------------------------------------------------------------
from .molfile import Mol, read_sdf, read_mol2, perceive, remove_hs as _remove_hs, BOND_FEATURE_INDEX, LIG_FEATURE_SOURCES  # noqa: E402
from ..torsion import get_transformation_mask  # noqa: E402

lig_feature_dims = ([119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2], 0)          # process_mols.py:95-112
_CHIRALITY = ["CHI_UNSPECIFIED", "CHI_TETRAHEDRAL_CW", "CHI_TETRAHEDRAL_CCW", "CHI_OTHER"]
_HYBRIDIZATION = ["SP", "SP2", "SP3", "SP3D", "SP3D2", "misc"]
_RANGE = {"degree": list(range(11)) + ["misc"], "charge": list(range(-5, 6)) + ["misc"], "implicit": list(range(7)) + ["misc"],
          "numH": list(range(9)) + ["misc"], "radical": list(range(5)) + ["misc"], "numring": list(range(7)) + ["misc"]}


def read_molecule(molecule_file, sanitize=False, calc_charges=False, remove_hs=False):
    """Reference process_mols.py:923-957.  `.sdf` and `.mol2` are read by datasets/molfile.py (`.pdbqt` / `.pdb` ligands need rdkit's
    bond perception: NotImplementedError).  `sanitize` runs the perception pass (`molfile.perceive`: aromatic bonds, hydrogen counts,
    hybridisation, chirality); a file that cannot be parsed gives None like a failed sanitisation there."""
    if molecule_file.endswith(".mol2"):
        reader = read_mol2
    elif molecule_file.endswith(".sdf"):
        reader = read_sdf
    elif molecule_file.endswith(".pdbqt") or molecule_file.endswith(".pdb"):
        raise NotImplementedError("ligands from .pdb / .pdbqt need rdkit's bond perception; convert to .sdf or .mol2")
    else:
        return ValueError("Expect the format of the molecule_file to be one of .mol2, .sdf, .pdbqt and .pdb, got {}".format(molecule_file))
    try:
        mol = reader(molecule_file)
        if sanitize or calc_charges:
            perceive(mol)
        if remove_hs:
            mol = _remove_hs(mol)
    except Exception:
        return None
    return mol


def read_sdf_or_mol2(sdf_fileName, mol2_fileName):
    """Reference process_mols.py:960-977: the SDF, else the MOL2; hydrogens removed; `problem` when neither parses."""
    mol, problem = None, False
    try:
        mol = _remove_hs(perceive(read_sdf(sdf_fileName)))
    except Exception:
        problem = True
    if problem:
        try:
            mol = _remove_hs(perceive(read_mol2(mol2_fileName)))
            problem = False
        except Exception:
            problem = True
    return mol, problem


def lig_atom_featurizer(mol: Mol) -> torch.Tensor:
    """[N, 16] categorical features in the reference's column order and vocabularies (process_mols.py:141-175)."""
    if not mol.perceived:
        perceive(mol)
    ring = mol.GetRingInfo()
    rows = []
    for idx, atom in enumerate(mol.GetAtoms()):
        rows.append([
            safe_index(allowable_features["possible_atomic_num_list"], atom.GetAtomicNum()),
            _CHIRALITY.index(atom.GetChiralTag()),
            safe_index(_RANGE["degree"], atom.GetTotalDegree()),
            safe_index(_RANGE["charge"], atom.GetFormalCharge()),
            safe_index(_RANGE["implicit"], atom.GetImplicitValence()),
            safe_index(_RANGE["numH"], atom.GetTotalNumHs()),
            safe_index(_RANGE["radical"], atom.GetNumRadicalElectrons()),
            safe_index(_HYBRIDIZATION, atom.GetHybridization()),
            int(atom.GetIsAromatic()),
            safe_index(_RANGE["numring"], ring.NumAtomRings(idx)),
        ] + [int(ring.IsAtomInRingOfSize(idx, k)) for k in range(3, 9)])
    return torch.tensor(rows, dtype=torch.long).reshape(-1, 16)


def get_lig_graph(mol: Mol, complex_graph):
    """Reference process_mols.py:567-589: atom features, every bond twice in a row (begin -> end, end -> begin) in the molecule's
    bond order, bond-type one-hot (single / double / triple / aromatic; unspecified -> single), coordinates."""
    atom_feats = lig_atom_featurizer(mol)
    row, col, edge_type = [], [], []
    for bond in mol.GetBonds():
        start, end = bond.GetBeginAtomIdx(), bond.GetEndAtomIdx()
        row += [start, end]
        col += [end, start]
        edge_type += 2 * [BOND_FEATURE_INDEX.get(bond.GetBondType(), 0)]
    edge_index = torch.tensor([row, col], dtype=torch.long).reshape(2, -1)
    edge_attr = torch.nn.functional.one_hot(torch.tensor(edge_type, dtype=torch.long), num_classes=4).to(torch.float)
    complex_graph["ligand"].x = atom_feats
    complex_graph["ligand", "lig_bond", "ligand"].edge_index = edge_index
    complex_graph["ligand", "lig_bond", "ligand"].edge_attr = edge_attr
    if mol.GetNumConformers() > 0:
        complex_graph["ligand"].pos = torch.from_numpy(mol.GetConformer().GetPositions()).float()


def get_lig_graph_with_matching(mol_, complex_graph, popsize=None, maxiter=None, matching=False, keep_original=False, num_conformers=1,
                                remove_hs=False, tries=10, skip_matching=False):
    """Reference process_mols.py:609-657, the `matching=False` branch (poses as given in the file; what inference on holo ligands
    uses).  Conformer generation + torsion matching (`matching=True`) is rdkit's ETKDG and differential evolution: not built."""
    if matching:
        raise NotImplementedError("conformer matching needs rdkit (ETKDG embedding); call with matching=False")
    complex_graph.rmsd_matching = 0
    if remove_hs:
        mol_ = _remove_hs(mol_ if mol_.perceived else perceive(mol_))
    if keep_original:
        complex_graph["ligand"].orig_pos = mol_.GetConformer().GetPositions()
    get_lig_graph(mol_, complex_graph)
    edge_mask, mask_rotate = get_transformation_mask(complex_graph)
    complex_graph["ligand"].edge_mask = torch.tensor(edge_mask)
    complex_graph["ligand"].mask_rotate = mask_rotate
    return mol_


def get_ligand(ligand_file, name, remove_hs=True, mol2_file=None, keep_original=True) -> HeteroData:
    """One ligand graph from an .sdf (fallback .mol2) file: what `PDBBind.get_complex` does for the ligand (datasets/pdbbind.py:378-400,
    `read_mol` :464-469), absolute coordinates."""
    lig = read_molecule(ligand_file, remove_hs=False, sanitize=True)
    if lig is None and mol2_file is not None:
        lig = read_molecule(mol2_file, remove_hs=False, sanitize=True)
    if lig is None or isinstance(lig, Exception):
        raise ValueError(f"could not read the ligand {ligand_file}")
    g = HeteroData()
    g.name = name
    g.mol = get_lig_graph_with_matching(lig, g, matching=False, keep_original=keep_original, remove_hs=remove_hs)
    return g


def get_complex(protein_pdb, ligand_file, name, device, remove_hs=True, mol2_file=None, all_atoms=True, lm_embeddings=None,
                sequences_to_embeddings=None, **receptor_kwargs) -> HeteroData:
    """Receptor + ligand of one complex from its files (e.g. data/1a0q/1a0q_protein_processed.pdb + 1a0q_ligand.sdf), as the
    reference's datasets assemble them (pdbbind.py:378-432): ligand stores, receptor (and all-atom) stores, everything centred on the
    C-alpha centroid, `original_center` kept, `orig_pos` in absolute coordinates."""
    g = get_ligand(ligand_file, name, remove_hs=remove_hs, mol2_file=mol2_file)
    rec = get_receptor(protein_pdb, name, device, all_atoms=all_atoms, lm_embeddings=lm_embeddings,
                       sequences_to_embeddings=sequences_to_embeddings, **receptor_kwargs)
    for key, st in rec._stores.items():
        g._stores[key] = st
    g.original_center = rec.original_center
    g.receptor_name = rec.receptor_name
    g["ligand"].pos = g["ligand"].pos - rec.original_center
    return g
