"""Build tests/golden/c1_1a0q.npz from the reference's example complex data/1a0q (BASELINE.json configs[0]).

TEST INFRASTRUCTURE ONLY; needs /root/reference (this container).  rdkit / biopython / ESM are not available, so this
is a PLUMBING fixture: real C-alpha coordinates and residue types of the 416 residues, real heavy-atom
coordinates + bonds of the ligand from the SDF; per-atom categorical features reduced to what the files give
(atomic number, degree, aromatic bond flag, ring flags = 0), rotatable bonds = the reference's own get_transformation_mask
(utils/torsion.py:15-45) run on the SDF's heavy-atom bond list, ESM block = seeded N(0, 0.5) placeholder.
(The ligand as the PACKAGE featurises it -- datasets/process_mols.get_ligand -- is covered by tests/test_ligand_featurise.py.)
Stored: arrays only (the graph schema of SURVEY.md 8b-4)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
D = "/root/reference/data/1a0q"
AA = ['ALA', 'ARG', 'ASN', 'ASP', 'CYS', 'GLN', 'GLU', 'GLY', 'HIS', 'ILE', 'LEU', 'LYS', 'MET', 'PHE', 'PRO', 'SER', 'THR',
      'TRP', 'TYR', 'VAL', 'HIP', 'HIE', 'TPO', 'HID', 'LEV', 'MEU', 'PTR', 'GLV', 'CYT', 'SEP', 'HIZ', 'CYM', 'GLM', 'ASQ',
      'TYS', 'CYX', 'GLZ']           # possible_amino_acids, datasets/process_mols.py:83-85 ('misc' = 37)
Z = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9, "P": 15, "S": 16, "CL": 17, "BR": 35, "I": 53}


def main():
    from scipy.spatial import cKDTree
    ca, rtype = [], []
    for line in open(os.path.join(D, "1a0q_protein_processed.pdb")):
        if line.startswith("ATOM") and line[12:16].strip() == "CA":
            ca.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
            r = line[17:20]
            rtype.append(AA.index(r) if r in AA else 37)
    ca = np.asarray(ca, dtype=np.float64)
    lines = open(os.path.join(D, "1a0q_ligand.sdf")).read().splitlines()
    na, nb = int(lines[3][:3]), int(lines[3][3:6])
    xyz, el = [], []
    for l in lines[4:4 + na]:
        xyz.append([float(l[0:10]), float(l[10:20]), float(l[20:30])])
        el.append(l[31:34].strip().upper())
    bonds = [(int(l[0:3]) - 1, int(l[3:6]) - 1, int(l[6:9])) for l in lines[4 + na:4 + na + nb]]
    heavy = [i for i, e in enumerate(el) if e != "H"]        # remove_hs: true
    remap = {a: k for k, a in enumerate(heavy)}
    hb = [(remap[a], remap[b], t) for a, b, t in bonds if a in remap and b in remap]
    Nl = len(heavy)
    lpos = np.asarray(xyz)[heavy]
    center = ca.mean(0)                                       # positions are stored relative to the protein centre
    deg = np.zeros(Nl, dtype=int)
    for a, b, _ in hb:
        deg[a] += 1
        deg[b] += 1
    x = np.zeros((Nl, 16), dtype=np.int64)
    x[:, 0] = [Z.get(el[i], 119) - 1 for i in heavy]          # possible_atomic_num_list index
    x[:, 2] = np.minimum(deg, 11)
    x[:, 3] = 5                                               # formal charge 0
    x[:, 7] = 2                                               # SP3 placeholder
    pairs = [(a, b) for a, b, _ in hb]
    edge_index = np.zeros((2, 2 * len(hb)), dtype=np.int64)
    edge_attr = np.zeros((2 * len(hb), 4), dtype=np.float32)
    for k, (a, b, t) in enumerate(hb):
        edge_index[:, 2 * k], edge_index[:, 2 * k + 1] = (a, b), (b, a)
        edge_attr[2 * k:2 * k + 2, {1: 0, 2: 1, 3: 2, 4: 3}.get(t, 0)] = 1
    # rotatable bonds: the REFERENCE's own get_transformation_mask (utils/torsion.py:15-45; no bond-order test there) run on this edge
    # list through the loader and the to_networkx restatement of oracle/make_golden_ligand.py
    from oracle import ref_import
    from oracle.make_golden_ligand import LigandOnlyGraph, to_networkx
    ref_import.install(load_tables=False)
    import utils.torsion as rt
    rt.to_networkx = to_networkx
    edge_mask, mask_rotate = rt.get_transformation_mask(LigandOnlyGraph(Nl, edge_index))
    rows = list(mask_rotate)
    Nr = len(ca)
    _, nbr = cKDTree(ca).query(ca, k=25)
    rec_edge_index = np.stack([nbr[:, 1:].reshape(-1), np.repeat(np.arange(Nr), 24)]).astype(np.int64)
    out = os.path.join(ROOT, "tests", "golden", "c1_1a0q.npz")
    np.savez_compressed(out, lig_x=x, lig_pos=(lpos - center).astype(np.float32), edge_index=edge_index, edge_attr=edge_attr,
                        edge_mask=edge_mask, mask_rotate=mask_rotate, rec_type=np.asarray(rtype, dtype=np.int64),
                        rec_pos=(ca - center).astype(np.float32), rec_edge_index=rec_edge_index,
                        original_center=center.astype(np.float32))
    print("wrote", out, os.path.getsize(out), "bytes; Nl", Nl, "Nr", Nr, "bonds", len(hb), "R", len(rows))


if __name__ == "__main__":
    main()
