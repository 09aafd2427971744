"""`NoiseTransform`: the forward-diffusion transform the fine-tuning DataLoader applies to every buffered pose
(reference datasets/pdbbind.py:25-133), with the reference's constructor arguments, RNG draw order and output fields:
  t ~ Beta(alpha, beta) (numpy global generator), sigma = t_to_sigma(t),
  tr_update ~ N(0, sigma_tr) (torch global generator, shape (1,3)), rot_update = so3.sample_vec(sigma_rot),
  torsion_updates ~ N(0, sigma_tor) (numpy, one per rotatable bond), pose moved by modify_conformer, and
  data.tr_score = -tr_update / sigma_tr^2, data.rot_score = so3.score_vec(...), data.tor_score = torus.score(...),
  data.tor_sigma_edge, complex_t / node_t set to t.
Host-side like the reference (it runs in the loader, one complex at a time, O(Nl) work); `time_independent`,
`crop_beyond_cutoff`, all-atom and asynchronous schedules are outside the hot path's scope and raise.
"""
from __future__ import annotations

import math
import random

import numpy as np
import torch

from .. import so3, torus
from ..diffusion_utils import set_time
from ..sampling import _mask_rotate_of, modify_conformer_torsion_angles


def axis_angle_to_matrix(aa: torch.Tensor) -> torch.Tensor:
    """Rotation matrix of an axis-angle vector through the unit quaternion (reference utils/geometry.py:43-86, incl. the
    small-angle series of sin(x/2)/x below 1e-6)."""
    ang = torch.linalg.vector_norm(aa, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - ang * ang / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    q = torch.cat([torch.cos(half), aa * k], dim=-1)
    r, i, j, kk = q.unbind(-1)
    two_s = 2.0 / (q * q).sum(-1)
    m = torch.stack([1 - two_s * (j * j + kk * kk), two_s * (i * j - kk * r), two_s * (i * kk + j * r),
                     two_s * (i * j + kk * r), 1 - two_s * (i * i + kk * kk), two_s * (j * kk - i * r),
                     two_s * (i * kk - j * r), two_s * (j * kk + i * r), 1 - two_s * (i * i + j * j)], dim=-1)
    return m.reshape(aa.shape[:-1] + (3, 3))


def kabsch(A: torch.Tensor, B: torch.Tensor):
    """R [3,3], t [3,1] minimising |R A + t - B| for 3xN point sets (reference utils/geometry.py:209-243).  The 3x3 SVD runs in
    numpy float64: torch's CPU LAPACK path spins up the whole thread pool for it (tens of ms per call on a many-core host)."""
    a, b_ = A.detach().cpu().numpy().astype(np.float64), B.detach().cpu().numpy().astype(np.float64)
    ca, cb = a.mean(axis=1, keepdims=True), b_.mean(axis=1, keepdims=True)
    U, S, Vt = np.linalg.svd((a - ca) @ (b_ - cb).T)
    R = Vt.T @ U.T
    if np.linalg.det(R) < 0:
        R = (Vt.T @ np.diag([1.0, 1.0, -1.0])) @ U.T
    assert math.fabs(np.linalg.det(R) - 1) < 3e-3
    t = -R @ ca + cb
    return torch.from_numpy(R).to(A.dtype), torch.from_numpy(t).to(A.dtype)


def modify_conformer(data, tr_update, rot_update, torsion_updates):
    """Rigid move about the centroid, torsion updates, Kabsch re-alignment onto the rigid pose
    (reference utils/diffusion_utils.py:33-58, pivot=None)."""
    pos = data["ligand"].pos
    center = torch.mean(pos, dim=0, keepdim=True)
    rot_mat = axis_angle_to_matrix(rot_update.squeeze())
    rigid = (pos - center) @ rot_mat.T + tr_update + center
    if torsion_updates is not None:
        ei = data["ligand", "ligand"].edge_index.T[data["ligand"].edge_mask]
        flex = modify_conformer_torsion_angles(rigid, ei, _mask_rotate_of(data), torsion_updates).to(rigid.device)
        R, t = kabsch(flex.T, rigid.T)
        data["ligand"].pos = flex @ R.T + t.T
    else:
        data["ligand"].pos = rigid
    return data


class NoiseTransform:
    def __init__(self, t_to_sigma, no_torsion, all_atom, alpha=1, beta=1, rot_alpha=1, rot_beta=1, tor_alpha=1, tor_beta=1,
                 separate_noise_schedule=False, asyncronous_noise_schedule=False, include_miscellaneous_atoms=False,
                 crop_beyond_cutoff=None, time_independent=False, rmsd_cutoff=0, minimum_t=0, sampling_mixing_coeff=0):
        if all_atom or asyncronous_noise_schedule or include_miscellaneous_atoms or time_independent or crop_beyond_cutoff is not None:
            raise NotImplementedError("all_atom / asynchronous / time_independent / crop_beyond_cutoff noise transforms are "
                                      "outside the score-model fine-tuning path")
        self.t_to_sigma, self.no_torsion, self.all_atom = t_to_sigma, no_torsion, all_atom
        self.minimum_t, self.mixing_coeff = minimum_t, sampling_mixing_coeff
        self.separate_noise_schedule = separate_noise_schedule
        self.alpha, self.beta = alpha, beta
        self.rot_alpha, self.rot_beta, self.tor_alpha, self.tor_beta = rot_alpha, rot_beta, tor_alpha, tor_beta

    def __call__(self, data):
        t_tr, t_rot, t_tor, t = self.get_time()
        return self.apply_noise(data, t_tr, t_rot, t_tor, t)

    def get_time(self):
        if self.separate_noise_schedule:
            return (np.random.beta(self.alpha, self.beta), np.random.beta(self.rot_alpha, self.rot_beta),
                    np.random.beta(self.tor_alpha, self.tor_beta), None)
        if self.mixing_coeff == 0:
            t = np.random.beta(self.alpha, self.beta)
            t = self.minimum_t + t * (1 - self.minimum_t)
        else:
            choice = np.random.binomial(1, self.mixing_coeff)
            t1 = np.random.beta(self.alpha, self.beta) * self.minimum_t
            t2 = self.minimum_t + np.random.beta(self.alpha, self.beta) * (1 - self.minimum_t)
            t = choice * t1 + (1 - choice) * t2
        return t, t, t, t

    def apply_noise(self, data, t_tr, t_rot, t_tor, t, tr_update=None, rot_update=None, torsion_updates=None):
        if not torch.is_tensor(data["ligand"].pos):
            data["ligand"].pos = random.choice(data["ligand"].pos)
        tr_sigma, rot_sigma, tor_sigma = self.t_to_sigma(t_tr, t_rot, t_tor)
        set_time(data, t, t_tr, t_rot, t_tor, 1, self.all_atom, False, device=None)
        tr_update = torch.normal(mean=0, std=tr_sigma, size=(1, 3)) if tr_update is None else tr_update
        rot_update = so3.sample_vec(eps=rot_sigma) if rot_update is None else rot_update
        n_tor = int(data["ligand"].edge_mask.sum())
        torsion_updates = np.random.normal(loc=0.0, scale=tor_sigma, size=n_tor) if torsion_updates is None else torsion_updates
        torsion_updates = None if self.no_torsion else torsion_updates
        modify_conformer(data, tr_update, torch.from_numpy(rot_update).float(), torsion_updates)
        data.tr_score = -tr_update / tr_sigma ** 2
        data.rot_score = torch.from_numpy(so3.score_vec(vec=rot_update, eps=rot_sigma)).float().unsqueeze(0)
        data.tor_score = None if self.no_torsion else torch.from_numpy(torus.score(torsion_updates, tor_sigma)).float()
        data.tor_sigma_edge = None if self.no_torsion else np.ones(n_tor) * tor_sigma
        if data["ligand"].pos.shape[0] == 1:
            data.rot_score = data.rot_score * 0   # a single atom has no orientation
        return data
