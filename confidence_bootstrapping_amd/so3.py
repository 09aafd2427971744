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
