"""Minimal heterogeneous-graph containers with the slice of the torch_geometric API that the reference's
sampler touches (torch_geometric is not installed on the target image and the hot path does not need it).

Mirrors what `utils/sampling.py:78-91,231-233` and `utils/diffusion_utils.py:60-64,150-179` use:
  data['ligand'].pos / .x / .edge_mask / .mask_rotate / .num_nodes / .batch
  data['ligand','ligand'].edge_index / .edge_attr / .num_edges
  data['receptor'].x / .pos ; data['receptor','receptor'].edge_index
  Batch.from_data_list, .num_graphs, .to(device), DataLoader(list, batch_size)
A three-element key ('receptor','rec_contact','receptor') addresses the same store as ('receptor','receptor').
"""
from __future__ import annotations

import copy
from typing import Dict, Iterable, List

import numpy as np
import torch


class Store:
    """Attribute bag for one node type or edge type."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith("_")]

    def __contains__(self, k):
        return k in self.__dict__

    @property
    def num_nodes(self):
        for k in ("pos", "x"):
            if k in self.__dict__:
                return int(self.__dict__[k].shape[0])
        raise AttributeError("num_nodes")

    @property
    def num_edges(self):
        return int(self.__dict__["edge_index"].shape[1])

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
            elif isinstance(v, dict):
                self.__dict__[k] = {kk: (vv.to(device) if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
        return self


def _norm_key(key):
    if isinstance(key, tuple):
        if len(key) == 3:
            return (key[0], key[2])
        return tuple(key)
    return key


class HeteroData:
    def __init__(self):
        object.__setattr__(self, "_stores", {})

    def __getitem__(self, key) -> Store:
        key = _norm_key(key)
        st = self._stores.get(key)
        if st is None:
            st = self._stores[key] = Store()
        return st

    def __contains__(self, key):
        return _norm_key(key) in self._stores or key in self.__dict__

    @property
    def node_types(self):
        return [k for k in self._stores if isinstance(k, str)]

    @property
    def edge_types(self):
        return [k for k in self._stores if isinstance(k, tuple)]

    @property
    def num_graphs(self):
        return 1

    def to(self, device):
        for st in self._stores.values():
            st.to(device)
        for k, v in list(self.__dict__.items()):
            if k.startswith("_"):
                continue
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
            elif isinstance(v, dict):
                self.__dict__[k] = {kk: (vv.to(device) if torch.is_tensor(vv) else vv) for kk, vv in v.items()}
        return self

    def shallow_copy(self):
        """New graph object with its own attribute stores that SHARE the tensors of this one.  Everything on the sampling and
        fine-tuning paths re-binds attributes (`g['ligand'].pos = ...`, `g.complex_t = ...`) instead of writing into tensors, so
        a shallow copy isolates them at a fraction of a deepcopy's cost (the 1281-wide receptor features and the all-atom stores
        are 3 MB per complex)."""
        out = type(self)()
        for k, st in self._stores.items():
            out._stores[k] = Store(**st.__dict__)
        for k, v in self.__dict__.items():
            if not k.startswith("_"):
                out.__dict__[k] = dict(v) if isinstance(v, dict) else v
        if hasattr(self, "_num_graphs"):
            object.__setattr__(out, "_num_graphs", self._num_graphs)
        return out

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def clone(self):
        return copy.deepcopy(self)


_CAT_NODE = ("x", "pos")


class Batch(HeteroData):
    """Collation of a list of HeteroData: node tensors concatenated, edge_index offset per graph,
    `batch` vectors added, non-tensor attributes (mask_rotate, name) gathered into lists."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_num_graphs", 0)

    @property
    def num_graphs(self):
        return self._num_graphs

    @classmethod
    def from_data_list(cls, data_list: List[HeteroData]) -> "Batch":
        out = cls()
        object.__setattr__(out, "_num_graphs", len(data_list))
        first = data_list[0]
        node_offsets: Dict[str, List[int]] = {}
        for nt in first.node_types:
            sizes = [d[nt].num_nodes for d in data_list]
            offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
            node_offsets[nt] = offs
            st = out[nt]
            for k in first[nt].keys():
                vals = [getattr(d[nt], k) for d in data_list]
                if torch.is_tensor(vals[0]):
                    setattr(st, k, torch.cat(vals, dim=0))
                else:
                    setattr(st, k, list(vals))
            st.batch = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(sizes)])
        for et in first.edge_types:
            st = out[et]
            for k in first[et].keys():
                vals = [getattr(d[et], k) for d in data_list]
                if k == "edge_index":
                    so, do = node_offsets[et[0]], node_offsets[et[-1]]
                    vals = [v + torch.tensor([[so[i]], [do[i]]], dtype=v.dtype) for i, v in enumerate(vals)]
                    setattr(st, k, torch.cat(vals, dim=1))
                elif torch.is_tensor(vals[0]):
                    setattr(st, k, torch.cat(vals, dim=0))
                else:
                    setattr(st, k, list(vals))
        for k, v in first.__dict__.items():
            if k.startswith("_"):
                continue
            vals = [d.__dict__.get(k) for d in data_list]
            if torch.is_tensor(v):
                out.__dict__[k] = torch.cat([x if x.dim() > 0 else x[None] for x in vals], dim=0)
            elif isinstance(v, dict) and v and all(isinstance(x, dict) and x.keys() == v.keys() and all(torch.is_tensor(t) for t in x.values())
                                                   for x in vals):
                # a dict of tensors per graph (complex_t: the diffusion times): torch_geometric collates it key by key
                out.__dict__[k] = {kk: torch.cat([x[kk] if x[kk].dim() > 0 else x[kk][None] for x in vals], dim=0) for kk in v}
            else:
                out.__dict__[k] = list(vals)
        return out


    def to_data_list(self) -> List[HeteroData]:
        """Inverse of from_data_list (torch_geometric's Batch.to_data_list, which the reference's sampler uses to crop every graph of a
        batch on its own, utils/sampling.py:102-106): node tensors split by `batch`, edge lists split by the graph of their first node
        with the per-graph node offsets removed, list attributes handed back element by element."""
        n = self._num_graphs
        out = [HeteroData() for _ in range(n)]
        sizes, offsets = {}, {}
        for nt in self.node_types:
            cnt = torch.bincount(self._stores[nt].batch.cpu(), minlength=n).tolist()
            sizes[nt] = cnt
            offsets[nt] = np.concatenate([[0], np.cumsum(cnt)]).astype(int).tolist()
        edge_off = {}           # per edge type: cumulative edge counts per graph (edges are stored graph after graph)
        for et in self.edge_types:
            ei = self._stores[et].edge_index
            gid = torch.bucketize(ei[0].cpu(), torch.tensor(offsets[et[0]][1:]), right=True)
            edge_off[et] = np.concatenate([[0], np.cumsum(torch.bincount(gid, minlength=n).tolist())]).astype(int).tolist()
        for nt in self.node_types:
            st = self._stores[nt]
            for k, v in st.__dict__.items():
                if k == "batch":
                    continue
                cuts = None
                if torch.is_tensor(v) and v.dim() > 0:
                    if v.shape[0] == offsets[nt][-1]:
                        cuts = offsets[nt]
                    else:   # a per-EDGE attribute kept on the node store (ligand.edge_mask: one flag per bond of the ligand's edge list)
                        cuts = next((edge_off[et] for et in self.edge_types if et[0] == nt and edge_off[et][-1] == v.shape[0]), None)
                for i in range(n):
                    if cuts is not None:
                        setattr(out[i][nt], k, v[cuts[i]:cuts[i + 1]])
                    elif isinstance(v, list) and len(v) == n:
                        setattr(out[i][nt], k, v[i])
                    else:
                        setattr(out[i][nt], k, v)
        for et in self.edge_types:
            st = self._stores[et]
            ei = st.edge_index
            src_t, dst_t = et[0], et[-1]
            gid = torch.bucketize(ei[0].cpu(), torch.tensor(offsets[src_t][1:]), right=True)
            for i in range(n):
                sel = (gid == i).to(ei.device)
                for k, v in st.__dict__.items():
                    if k == "edge_index":
                        off = torch.tensor([[offsets[src_t][i]], [offsets[dst_t][i]]], dtype=ei.dtype, device=ei.device)
                        setattr(out[i][et], k, ei[:, sel] - off)
                    elif torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == ei.shape[1]:
                        setattr(out[i][et], k, v[sel])
                    elif isinstance(v, list) and len(v) == n:
                        setattr(out[i][et], k, v[i])
                    else:
                        setattr(out[i][et], k, v)
        for k, v in self.__dict__.items():
            if k.startswith("_"):
                continue
            for i in range(n):
                if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n:
                    out[i].__dict__[k] = v[i:i + 1]
                elif isinstance(v, list) and len(v) == n:
                    out[i].__dict__[k] = v[i]
                elif isinstance(v, dict):
                    out[i].__dict__[k] = {kk: (vv[i:i + 1] if torch.is_tensor(vv) and vv.dim() > 0 and vv.shape[0] == n else vv) for kk, vv in v.items()}
                else:
                    out[i].__dict__[k] = v
        return out


class DataLoader:
    """`torch_geometric.loader.DataLoader(data_list, batch_size)` without shuffling."""

    def __init__(self, data_list: Iterable[HeteroData], batch_size: int = 1):
        self.data_list = list(data_list)
        self.batch_size = int(batch_size)

    def __len__(self):
        return (len(self.data_list) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for i in range(0, len(self.data_list), self.batch_size):
            yield Batch.from_data_list(self.data_list[i:i + self.batch_size])
