"""Pin the two f3 pieces this repository could NOT pin in its build image (no rdkit, no torch_geometric): run by a user whose
environment is the reference's own (environment.yml: rdkit, torch_geometric 2.0.4) next to a checkout of the reference.

    python tools/pin_with_rdkit.py --reference /path/to/confidence-bootstrapping [--ligand file.sdf ...] [--write-fixtures]

What it does (nothing of the reference is copied: its modules are IMPORTED from the checkout at run time, this repository only reads
the arrays they return):
  1. ligand atom features -- for every ligand file (default: tests/golden/1a0q/1a0q_ligand.sdf and .mol2) the reference's
     `read_molecule` + `lig_atom_featurizer` (datasets/process_mols.py:141-175, 923-977; rdkit's perception) against this package's
     `datasets.process_mols.read_molecule` + `lig_atom_featurizer` (own valence / ring / aromaticity / hybridisation perception,
     datasets/molfile.py).  Prints a per-column table (LIG_FEATURE_SOURCES names the columns that are restated and unpinned: 1 chirality,
     7 hybridisation, 8 aromaticity) and the atoms that differ; also bond lists + bond types of `get_lig_graph`, and `edge_mask` /
     `mask_rotate` of `get_transformation_mask`.
  2. dataset cache -- builds the 1a0q ligand graph as a REAL torch_geometric HeteroData with the reference's `get_lig_graph`, pickles
     it the way `datasets/moad.py:297-470` does, reads the file back through this package's restricted unpickler
     (`datasets/cache_reader.load_pyg_cache`, so far validated only against a pickle EMULATED by oracle/make_cache_fixture.py) and
     compares every store and attribute.  `--cache file.pkl` does the same for an existing cache file of the user's.
  3. `--write-fixtures`: saves what the reference returned as tests/golden/g21_rdkit_ligand.npz + g21_pyg_cache.pkl so that the next
     build can commit them as reference-held vectors (data only).
Exit status 0 = everything equal, 1 = differences (listed), 2 = the reference environment is not importable here.
"""
import argparse
import io
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLUMNS = ["atomic_num", "chirality", "degree", "formal_charge", "implicit_valence", "numH", "radical_e", "hybridization", "is_aromatic",
           "numring", "ring3", "ring4", "ring5", "ring6", "ring7", "ring8"]


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--reference", required=True, help="checkout of LDeng0205/confidence-bootstrapping")
    ap.add_argument("--ligand", nargs="*", default=[os.path.join(ROOT, "tests", "golden", "1a0q", f) for f in ("1a0q_ligand.sdf", "1a0q_ligand.mol2")])
    ap.add_argument("--cache", default=None, help="an existing ligands.pkl / receptors*.pkl written by the reference")
    ap.add_argument("--write-fixtures", action="store_true")
    a = ap.parse_args()
    try:
        import rdkit  # noqa: F401
        import torch_geometric  # noqa: F401
    except Exception as e:
        print("this tool needs the reference's environment (rdkit, torch_geometric):", e)
        return 2
    import torch
    sys.path.insert(0, ROOT)
    from confidence_bootstrapping_amd.datasets import process_mols as mine
    from confidence_bootstrapping_amd.datasets.cache_reader import load_pyg_cache
    from confidence_bootstrapping_amd.hetero import HeteroData as MyHetero
    from confidence_bootstrapping_amd.torsion import get_transformation_mask as my_mask
    sys.path.insert(0, os.path.abspath(a.reference))
    from datasets import process_mols as ref            # the reference's module, imported from the user's checkout
    from torch_geometric.data import HeteroData
    from utils.torsion import get_transformation_mask as ref_mask

    bad = 0
    fixtures = {}
    for path in a.ligand:
        tag = os.path.basename(path)
        rmol = ref.read_molecule(path, remove_hs=True, sanitize=True)
        mmol = mine.read_molecule(path, remove_hs=True)
        if rmol is None:
            print(f"[{tag}] rdkit could not read the file: skipped")
            continue
        fr, fm = ref.lig_atom_featurizer(rmol).numpy(), mine.lig_atom_featurizer(mmol).numpy()
        print(f"[{tag}] {fr.shape[0]} atoms (reference) / {fm.shape[0]} (this package)")
        if fr.shape != fm.shape:
            print("  atom counts differ"); bad += 1
            continue
        for c, name in enumerate(COLUMNS):
            diff = np.flatnonzero(fr[:, c] != fm[:, c])
            status = "equal" if diff.size == 0 else f"{diff.size} atoms differ: " + ", ".join(f"{i}: ref {fr[i, c]} mine {fm[i, c]}" for i in diff[:12])
            print(f"  col {c:2d} {name:17s} {status}")
            bad += int(diff.size > 0)
        gr, gm = HeteroData(), MyHetero()
        ref.get_lig_graph(rmol, gr)
        mine.get_lig_graph(mmol, gm)
        ei_r, ei_m = gr["ligand", "lig_bond", "ligand"].edge_index.numpy(), gm["ligand", "lig_bond", "ligand"].edge_index.numpy()
        ea_r, ea_m = gr["ligand", "lig_bond", "ligand"].edge_attr.numpy(), gm["ligand", "lig_bond", "ligand"].edge_attr.numpy()
        same_edges = ei_r.shape == ei_m.shape and np.array_equal(ei_r, ei_m) and np.array_equal(ea_r, ea_m)
        print("  bond list + bond types:", "equal" if same_edges else "DIFFER")
        bad += int(not same_edges)
        mr, rr = ref_mask(gr)
        mm, rm = my_mask(gm)
        same_mask = np.array_equal(np.asarray(mr), np.asarray(mm)) and np.array_equal(np.asarray(rr), np.asarray(rm))
        print("  edge_mask / mask_rotate:", "equal" if same_mask else "DIFFER")
        bad += int(not same_mask)
        fixtures[tag] = dict(x=fr, edge_index=ei_r, edge_attr=ea_r, edge_mask=np.asarray(mr), mask_rotate=np.asarray(rr))
        if path == a.ligand[0]:
            # a REAL PyG pickle of this graph, read back through the restricted unpickler
            buf = io.BytesIO()
            pickle.dump([gr], buf)
            raw = buf.getvalue()
            got = load_pyg_cache(io.BytesIO(raw))[0]
            ok = True
            for store in ("ligand",):
                for k in gr[store].keys():
                    v, w = gr[store][k], got[store][k]
                    if torch.is_tensor(v):
                        ok &= torch.is_tensor(w) and torch.equal(v, w)
            e = ("ligand", "lig_bond", "ligand")
            ok &= torch.equal(gr[e].edge_index, got[e].edge_index) and torch.equal(gr[e].edge_attr, got[e].edge_attr)
            print("  real PyG pickle -> cache_reader.load_pyg_cache:", "equal" if ok else "DIFFERS")
            bad += int(not ok)
            fixtures["__pickle__"] = raw
    if a.cache:
        got = load_pyg_cache(a.cache)
        with open(a.cache, "rb") as f:
            want = pickle.load(f)           # the user's own file, in the user's own environment
        want = want if isinstance(want, (list, tuple)) else [want]
        got = got if isinstance(got, (list, tuple)) else [got]
        ok = len(want) == len(got)
        for gw, gg in zip(want, got):
            for store in gw.node_types:
                for k in gw[store].keys():
                    v = gw[store][k]
                    if torch.is_tensor(v):
                        ok &= torch.equal(v, gg[store][k])
            for e in gw.edge_types:
                for k in gw[e].keys():
                    v = gw[e][k]
                    if torch.is_tensor(v):
                        ok &= torch.equal(v, gg[e][k])
        print(f"[{os.path.basename(a.cache)}] {len(want)} graphs:", "equal" if ok else "DIFFER")
        bad += int(not ok)
    if a.write_fixtures and fixtures:
        out = os.path.join(ROOT, "tests", "golden")
        raw = fixtures.pop("__pickle__", None)
        np.savez(os.path.join(out, "g21_rdkit_ligand.npz"), **{f"{t}::{k}": v for t, d in fixtures.items() for k, v in d.items()})
        if raw is not None:
            open(os.path.join(out, "g21_pyg_cache.pkl"), "wb").write(raw)
        print("fixtures written under", out)
    print("RESULT:", "everything equal" if bad == 0 else f"{bad} difference(s)")
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
