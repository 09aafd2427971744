"""SO(3) score normaliser lookup (reference utils/so3.py:90-94) on the shipped 2000-entry table
(generated once from the reference's own series code by oracle/gen_tables.py; data/tables_meta.json)."""
import os

import numpy as np
import torch

MIN_EPS, MAX_EPS, N_EPS = 0.0005, 4, 2000
_exp_score_norms = np.load(os.path.join(os.path.dirname(__file__), "data", "so3_exp_score_norms.npy"))


def score_norm(eps):
    eps = eps.numpy() if torch.is_tensor(eps) else np.asarray(eps)
    eps_idx = (np.log10(eps) - np.log10(MIN_EPS)) / (np.log10(MAX_EPS) - np.log10(MIN_EPS)) * N_EPS
    eps_idx = np.clip(np.around(eps_idx).astype(int), a_min=0, a_max=N_EPS - 1)
    return torch.from_numpy(_exp_score_norms[eps_idx]).float()


# ---- sampling and score of IGSO(3) for the training-side NoiseTransform (reference utils/so3.py:17-88).  The reference tabulates
# 2000 x 2000 grids at import (8.5 minutes, 3 x 32 MB caches in the working directory); a training step touches one sigma row at a
# time, so rows are computed on demand by the same truncated series (L = 2000 terms, same summation order) and memoised.
from functools import lru_cache

X_N = 2000
_omegas_array = np.linspace(0, np.pi, X_N + 1)[1:]


def _eps_of(idx):
    return (10 ** np.linspace(np.log10(MIN_EPS), np.log10(MAX_EPS), N_EPS))[idx]


def _eps_idx(eps):
    i = (np.log10(eps) - np.log10(MIN_EPS)) / (np.log10(MAX_EPS) - np.log10(MIN_EPS)) * N_EPS
    return np.clip(np.around(i).astype(int), a_min=0, a_max=N_EPS - 1)


@lru_cache(maxsize=256)
def _rows(idx: int, L=2000):
    """(cdf row, score row) of the eps grid point `idx` over `_omegas_array`."""
    eps, om = _eps_of(idx), _omegas_array
    p, dsig = 0, 0
    lo, dlo = np.sin(om / 2), 1 / 2 * np.cos(om / 2)
    for l in range(L):
        c = (2 * l + 1) * np.exp(-l * (l + 1) * eps ** 2 / 2)
        hi = np.sin(om * (l + 1 / 2))
        p += c * hi / lo
        dsig += c * (lo * ((l + 1 / 2) * np.cos(om * (l + 1 / 2))) - hi * dlo) / lo ** 2
    pdf = p * (1 - np.cos(om)) / np.pi
    return pdf.cumsum() / X_N * np.pi, dsig / p


def sample(eps):
    x = np.random.rand()
    return np.interp(x, _rows(int(_eps_idx(eps)))[0], _omegas_array)


def sample_vec(eps):
    x = np.random.randn(3)
    x /= np.linalg.norm(x)
    return x * sample(eps)


def score_vec(eps, vec):
    om = np.linalg.norm(vec)
    return np.interp(om, _omegas_array, _rows(int(_eps_idx(eps)))[1]) * vec / om
