"""Closed forms of the real Wigner-3j tensors whose arithmetic is hard-wired in the HIP kernels, and the check of a real
checkpoint's e3nn code-gen constants against them.

The reference builds its heads with e3nn (`o3.FullyConnectedTensorProduct`, models/tensor_layers.py:185, and
`o3.FullTensorProduct`, models/score_model.py:265); e3nn's generated modules keep the Wigner-3j tensors they contract with
as buffers named `_w3j_{l1}_{l2}_{l3}`.  The engines do not read those buffers -- the contractions are
written out as dot / cross / (b b^T - I/3) products in csrc/kernels.hip, tp_conv.hip and fctp_conv.hip -- so a checkpoint
whose constants differ from the ones baked in here must not load silently: `check_w3j_buffers` raises.

Basis: e3nn's real spherical harmonics, l = 1 -> (x, y, z); l = 2 -> (xz, xy, y^2 - (x^2+z^2)/2, yz, (z^2-x^2)/2) up to the
component normalisation; every tensor has unit Frobenius norm.
"""
from __future__ import annotations

import math
import re

import numpy as np

_S3H = math.sqrt(3.0) / 2.0
# Y2_j(b) = sqrt(5) * b^T Q_j b for a unit vector b; <Q_i, Q_j> = 1.5 delta_ij
Q2 = np.zeros((5, 3, 3))
Q2[0, 0, 2] = Q2[0, 2, 0] = _S3H
Q2[1, 0, 1] = Q2[1, 1, 0] = _S3H
Q2[2] = np.diag([-0.5, 1.0, -0.5])
Q2[3, 1, 2] = Q2[3, 2, 1] = _S3H
Q2[4] = np.diag([-_S3H, 0.0, _S3H])

_EPS = np.zeros((3, 3, 3))
for _i, _j, _k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
    _EPS[_i, _j, _k] = 1.0
    _EPS[_j, _i, _k] = -1.0


def w3j_closed_form(l1: int, l2: int, l3: int):
    """Real Wigner-3j [2l1+1, 2l2+1, 2l3+1] for the triples the shipped architectures contract with, else None.
    (0,l,l) / (l,0,l) / (l,l,0): delta / sqrt(2l+1);  (1,1,1): epsilon / sqrt(6);
    (1,2,1)[i,j,k] = sqrt(2/15) Q_j[i,k] (and its index permutations) -- the tensor behind the torsion head's
    T1 = (3/sqrt2)(b b^T - I/3)(sqrt3 v), csrc/tp_conv.hip::bond_conv_kernel."""
    d = lambda l: np.eye(2 * l + 1) / math.sqrt(2 * l + 1)
    if l1 == 0 and l2 == l3:
        return d(l2)[None, :, :]
    if l2 == 0 and l1 == l3:
        return d(l1)[:, None, :]
    if l3 == 0 and l1 == l2:
        return d(l1)[:, :, None]
    if (l1, l2, l3) == (1, 1, 1):
        return _EPS / math.sqrt(6.0)
    c = math.sqrt(2.0 / 15.0)
    if (l1, l2, l3) == (1, 2, 1):
        return c * np.transpose(Q2, (1, 0, 2))
    if (l1, l2, l3) == (2, 1, 1):
        return c * Q2
    if (l1, l2, l3) == (1, 1, 2):
        return c * np.transpose(Q2, (1, 2, 0))
    return None


_W3J_KEY = re.compile(r"(?:^|\.)_w3j_(\d+)_(\d+)_(\d+)$")


def check_w3j_buffers(state_dict, atol: float = 1e-6):
    """Compare every `*_w3j_l1_l2_l3` tensor of a checkpoint with the constants the kernels hard-wire.  Returns the keys that were
    checked; raises RuntimeError on a mismatch (shape or value).  Triples without a closed form here are not used by the engines."""
    checked = []
    for k, v in state_dict.items():
        m = _W3J_KEY.search(k)
        if not m:
            continue
        ls = tuple(int(x) for x in m.groups())
        ref = w3j_closed_form(*ls)
        if ref is None:
            continue
        got = np.asarray(v.detach().cpu().double().numpy() if hasattr(v, "detach") else v, dtype=np.float64)
        if got.shape != ref.shape or not np.allclose(got, ref, atol=atol, rtol=0.0):
            raise RuntimeError(f"checkpoint tensor '{k}' differs from the Wigner-3j constants hard-wired in the MI355X kernels "
                               f"(l = {ls}); this checkpoint was written by an e3nn with a different basis / sign convention")
        checked.append(k)
    return checked
