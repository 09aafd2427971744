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


@lru_cache(maxsize=1)
def _basis(L=2000):
    """sin((l+1/2) w) and (l+1/2) cos((l+1/2) w) for l < L on the omega grid: 2 x 32 MB, built once on first use (0.2 s)."""
    arg = (np.arange(L)[:, None] + 0.5) * _omegas_array[None, :]
    return np.sin(arg), (np.arange(L)[:, None] + 0.5) * np.cos(arg)


@lru_cache(maxsize=512)
def _rows(idx: int, L=2000):
    """(cdf row, score row) of the eps grid point `idx` over `_omegas_array`: the reference's truncated series
    sum_l (2l+1) exp(-l(l+1) eps^2/2) sin((l+1/2)w)/sin(w/2) and its derivative (utils/so3.py:23-45), evaluated as two
    matrix-vector products against the cached basis (a python loop over the 2000 terms costs 80 ms per noised complex)."""
    eps, om = _eps_of(idx), _omegas_array
    hi, dhi = _basis(L)
    l = np.arange(L)
    c = (2 * l + 1) * np.exp(-l * (l + 1) * eps ** 2 / 2)
    n = max(int(np.count_nonzero(c)), 1)      # the coefficients underflow to exactly 0 beyond l ~ 39 / eps: those terms add nothing
    lo, dlo = np.sin(om / 2), 1 / 2 * np.cos(om / 2)
    s_hi = c[:n] @ hi[:n]
    p = s_hi / lo
    dsig = (lo * (c[:n] @ dhi[:n]) - dlo * s_hi) / lo ** 2
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
