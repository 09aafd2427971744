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


def load_c1_complex():
    """BASELINE.json configs[0]: the reference's example complex data/1a0q as a plumbing fixture
    (tests/golden/c1_1a0q.npz, built by oracle/make_c1_fixture.py; ESM block = seeded placeholder)."""
    import os
    from confidence_bootstrapping_amd.hetero import HeteroData
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "c1_1a0q.npz"))
    rng = np.random.default_rng(0)
    Nr = g["rec_pos"].shape[0]
    rec_x = np.zeros((Nr, 1281), dtype=np.float32)
    rec_x[:, 0] = g["rec_type"]
    rec_x[:, 1:] = rng.normal(0, 0.5, size=(Nr, 1280))
    d = HeteroData()
    d["ligand"].x = torch.from_numpy(g["lig_x"])
    d["ligand"].pos = torch.from_numpy(g["lig_pos"])
    d["ligand"].edge_mask = torch.from_numpy(g["edge_mask"])
    d["ligand"].mask_rotate = g["mask_rotate"]
    d["ligand", "ligand"].edge_index = torch.from_numpy(g["edge_index"])
    d["ligand", "ligand"].edge_attr = torch.from_numpy(g["edge_attr"])
    d["receptor"].x = torch.from_numpy(rec_x)
    d["receptor"].pos = torch.from_numpy(g["rec_pos"])
    d["receptor", "receptor"].edge_index = torch.from_numpy(g["rec_edge_index"])
    d.original_center = torch.from_numpy(g["original_center"])[None]
    d.name = "1a0q"
    return d


def to_aacx(d):
    """HeteroData (single complex with the all-atom stores) -> oracle AllAtomComplex."""
    from oracle.confidence_ref import AllAtomComplex
    return AllAtomComplex(d["ligand"].x, d["ligand", "ligand"].edge_index, d["ligand", "ligand"].edge_attr,
                          d["receptor"].x, d["receptor"].pos, d["receptor", "receptor"].edge_index,
                          d["atom"].x, d["atom"].pos, d["atom", "atom"].edge_index, d["atom", "receptor"].edge_index[1])
