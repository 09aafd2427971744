"""`CBBuffer`: the replay buffer of self-generated, confidence-filtered poses that the fine-tuning epochs draw from
(reference bootstrapping/buffer.py:9-116; filled by finetune_train.py:296-299, read through a DataListLoader at :300).

Same policy, names and argument meaning as the reference:
  * `add_complexes([(graph, confidence), ...])` stamps every graph with its confidence and the current iteration, resets its
    diffusion time to 0, extends (or with `reset_buffer` replaces) the pool, and -- if `max_complexes_per_couple` is set -- keeps
    per (receptor, ligand) couple (first 6 characters of `name[0]`) only the top-k by `confidence + buffer_decay * iteration`;
  * `get(idx)` returns a deep copy without the bookkeeping attributes: round-robin over the pool, or -- with `fixed_length` --
    a draw from softmax(temperature * confidence) on numpy's global generator;
  * `len()` = `len(pool) * multiplicity` or `fixed_length`.
Differences: `get` returns a copy that shares the (never written-to) tensors of the pooled graph instead of a deepcopy -- the
transform re-binds `pos`, times and scores; no torch_geometric `Dataset` base (absent here; `__len__`/`__getitem__` apply `transform` the way it does), and the
cluster -> ligand-names table can be passed in instead of being un-pickled from the hard-wired MOAD path.
"""
from __future__ import annotations

import copy
import os
import pickle

import numpy as np
import torch

_CLUSTER_PKL = "data/BindingMOAD_2020_processed/new_cluster_to_ligands.pkl"


def _name_of(graph):
    n = graph.name
    return n[0] if isinstance(n, (list, tuple)) else n


class CBBuffer:
    def __init__(self, cluster_name=None, root=None, transform=None, multiplicity=1, max_complexes_per_couple=None,
                 fixed_length=None, temperature=1.0, buffer_decay=0.2, reset_buffer=False, cluster_to_ligands=None):
        assert cluster_name is not None
        self.root, self.transform = root, transform
        self.multiplicity = multiplicity
        self.complexes = []
        self.iteration = 0
        self.max_complexes_per_couple = max_complexes_per_couple
        self.fixed_length = fixed_length
        self.temperature = temperature
        self.buffer_decay = buffer_decay
        self.reset_buffer = reset_buffer
        if cluster_to_ligands is None:
            if not os.path.exists(_CLUSTER_PKL):
                raise FileNotFoundError(f"{_CLUSTER_PKL} not found; pass cluster_to_ligands={{cluster: [ligand names]}}")
            with open(_CLUSTER_PKL, "rb") as f:
                cluster_to_ligands = pickle.load(f)
        self.cluster_to_ligands = cluster_to_ligands
        self.ligand_names = self.cluster_to_ligands[cluster_name]
        self.ligand_cnt = {name: 0 for name in self.ligand_names}

    # ---- dataset protocol
    def len(self):
        return len(self.complexes) * self.multiplicity if self.fixed_length is None else self.fixed_length

    __len__ = len

    def get(self, idx):
        if self.fixed_length is None:
            pick = idx % len(self.complexes)
        else:
            conf = np.asarray([float(c.confidence) for c in self.complexes])
            w = np.exp(conf * self.temperature)
            pick = np.random.choice(len(self.complexes), p=w / np.sum(w))
        g = self.complexes[pick].shallow_copy() if hasattr(self.complexes[pick], "shallow_copy") else copy.deepcopy(self.complexes[pick])
        for attr in ("confidence", "iteration"):
            g.__dict__.pop(attr, None)
            for nt in ("receptor", "ligand"):
                g[nt].__dict__.pop(attr, None)
        return g

    def __getitem__(self, idx):
        g = self.get(idx)
        return g if self.transform is None else self.transform(g)

    def statistics(self):
        return {"complexes": len(self.complexes), "per_ligand": dict(self.ligand_cnt)}

    # ---- policy
    def add_complexes(self, new_complex_list):
        for g, confidence in new_complex_list:
            g.confidence = confidence
            g.iteration = self.iteration
            nl, nr = g["ligand"].num_nodes, g["receptor"].num_nodes
            g.complex_t = {k: torch.zeros(1) for k in ("tr", "rot", "tor")}
            g["ligand"].node_t = {k: torch.zeros(nl) for k in ("tr", "rot", "tor")}
            g["receptor"].node_t = {k: torch.zeros(nr) for k in ("tr", "rot", "tor")}
            self.ligand_cnt[_name_of(g)] = self.ligand_cnt.get(_name_of(g), 0) + 1
            g.to("cpu")
        self.iteration += 1
        fresh = [g for g, _ in new_complex_list]
        self.complexes = fresh if self.reset_buffer else self.complexes + fresh
        if self.max_complexes_per_couple is not None:
            couples = {}
            for g in self.complexes:                    # insertion order of first appearance, like the reference's dict
                couples.setdefault(_name_of(g)[:6], []).append((float(g.confidence) + self.buffer_decay * g.iteration, g))
            kept = []
            for key, items in couples.items():
                if len(items) > self.max_complexes_per_couple:
                    items = sorted(items, key=lambda x: x[0], reverse=True)[:self.max_complexes_per_couple]   # stable, like sorted()
                kept.extend(g for _, g in items)
            self.complexes = kept
