"""Build tests/golden/c1_1a0q_pyg_cache.pkl: the reference's example complex data/1a0q in the FORMAT of the reference's dataset caches
(`datasets/moad.py:338-339,450-453`: pickled torch_geometric HeteroData graphs), for the cache-reader tests.

TEST INFRASTRUCTURE ONLY.  torch_geometric is not installable here, so the classes the pickle stream names are emulated in this script
with the state layout of torch_geometric 2.0.4 (the version the reference pins, environment.yml:188): `HeteroData.__dict__` =
{_global_store, _node_store_dict, _edge_store_dict}; every storage's state = its `__dict__` = {_mapping, _key, _parent}
(torch_geometric/data/storage.py `__getstate__` replaces the weak parent reference by the parent object).  The arrays come from
tests/golden/c1_1a0q.npz (built from data/1a0q by oracle/make_c1_fixture.py).  The file also holds a stand-in for the pickled rdkit
molecule of `rdkit_ligands.pkl` (an opaque blob to the reader).  Usage: python oracle/make_cache_fixture.py"""
import os
import pickle
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _fake_modules():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    for n in ("torch_geometric", "torch_geometric.data", "rdkit", "rdkit.Chem"):
        mod(n)
    st, hd, rd = mod("torch_geometric.data.storage"), mod("torch_geometric.data.hetero_data"), mod("rdkit.Chem.rdchem")

    class BaseStorage:
        def __init__(self, _mapping=None, **kw):
            self.__dict__["_mapping"] = dict(_mapping or {})
            for k, v in kw.items():
                self.__dict__[k] = v

        def __getstate__(self):
            return self.__dict__.copy()          # PyG: the weakref `_parent` is replaced by the parent object itself

        def __setstate__(self, state):
            self.__dict__.update(state)

    class NodeStorage(BaseStorage):
        pass

    class EdgeStorage(BaseStorage):
        pass

    class HeteroData:
        def __init__(self):
            self.__dict__["_global_store"] = BaseStorage(_parent=self)
            self.__dict__["_node_store_dict"] = {}
            self.__dict__["_edge_store_dict"] = {}

        def node(self, key):
            return self._node_store_dict.setdefault(key, NodeStorage(_parent=self, _key=key))

        def edge(self, key):
            return self._edge_store_dict.setdefault(key, EdgeStorage(_parent=self, _key=key))

    class Mol:
        def __init__(self, blob=b""):
            self.blob = blob

        def __reduce__(self):
            return (Mol, (self.blob,))
    for cls, m in ((BaseStorage, st), (NodeStorage, st), (EdgeStorage, st), (HeteroData, hd), (Mol, rd)):
        cls.__module__ = m.__name__
        cls.__qualname__ = cls.__name__
        setattr(m, cls.__name__, cls)
    return HeteroData, Mol


def main():
    HeteroData, Mol = _fake_modules()
    g = np.load(os.path.join(ROOT, "tests", "golden", "c1_1a0q.npz"))
    rec = HeteroData()
    rec._global_store._mapping["name"] = "1a0q"
    rec._global_store._mapping["original_center"] = torch.from_numpy(g["original_center"])[None]
    r = rec.node("receptor")
    r._mapping["x"] = torch.from_numpy(g["rec_type"]).float()[:, None]          # residue type only: the ESM block is attached from its own .pt
    r._mapping["pos"] = torch.from_numpy(g["rec_pos"])
    r._mapping["chain_ids"] = torch.zeros(len(g["rec_type"]), dtype=torch.long)
    rec.edge(("receptor", "rec_contact", "receptor"))._mapping["edge_index"] = torch.from_numpy(g["rec_edge_index"])
    lig = HeteroData()
    lig._global_store._mapping["name"] = "1a0q"
    l = lig.node("ligand")
    l._mapping["x"] = torch.from_numpy(g["lig_x"])
    l._mapping["pos"] = torch.from_numpy(g["lig_pos"]) + torch.from_numpy(g["original_center"])      # ligand caches hold un-centred poses (moad.py:205-209)
    l._mapping["orig_pos"] = (g["lig_pos"] + g["original_center"]).astype(np.float32)
    l._mapping["edge_mask"] = torch.from_numpy(g["edge_mask"])
    l._mapping["mask_rotate"] = g["mask_rotate"]
    e = lig.edge(("ligand", "lig_bond", "ligand"))
    e._mapping["edge_index"] = torch.from_numpy(g["edge_index"])
    e._mapping["edge_attr"] = torch.from_numpy(g["edge_attr"])
    out = os.path.join(ROOT, "tests", "golden", "c1_1a0q_pyg_cache.pkl")
    with open(out, "wb") as f:
        pickle.dump({"receptors": [rec], "ligands": {"1a0q": lig}, "rdkit_ligands": {"1a0q": Mol(b"\\x00opaque rdkit pickle\\x01")}}, f, protocol=4)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
