"""Shared helpers for the parity tests (oracle <-> golden <-> HIP engine)."""
import copy

import numpy as np
import torch


def to_cx(d):
    """HeteroData (single complex) -> oracle ComplexData."""
    from oracle.score_ref import ComplexData
    mr = d["ligand"].mask_rotate
    if isinstance(mr, list):
        mr = mr[0]
    return ComplexData(d["ligand"].x, d["ligand", "ligand"].edge_index, d["ligand", "ligand"].edge_attr,
                       d["ligand"].edge_mask, np.asarray(mr), d["receptor"].x, d["receptor"].pos,
                       d["receptor", "receptor"].edge_index)


def rmsd(a, b):
    a, b = torch.as_tensor(a), torch.as_tensor(b)
    return torch.sqrt(((a - b) ** 2).sum(-1).mean(-1))


def rel_err(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / b.abs().max().clamp(min=1e-30))
