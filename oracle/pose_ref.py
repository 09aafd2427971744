"""CPU restatement of the reference's pose update and reverse-SDE step.

TEST INFRASTRUCTURE ONLY (oracle) -- see oracle/e3nn_ref.py header for the import rule.

Follows (file:line in /root/reference):
  axis_angle_to_matrix (via quaternion)      utils/geometry.py:7-86
  rigid_transform_Kabsch_3D_torch_batch      utils/geometry.py:246-276
  modify_conformer_torsion_angles_batch      utils/torsion.py:75-90
  modify_conformer_torsion_angles (numpy)    utils/torsion.py:48-72
  modify_conformer_batch                     utils/diffusion_utils.py:60-78
  get_t_schedule                             utils/diffusion_utils.py:138-143
  per-step SDE update in sampling()          utils/sampling.py:93-144, 221-223
  randomize_position                         utils/sampling.py:15-48
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.stats import beta
from scipy.spatial.transform import Rotation

from .score_ref import ScoreConfig, ComplexData, score_forward, receptor_embedding, t_to_sigma


def axis_angle_to_matrix(aa: torch.Tensor) -> torch.Tensor:
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * ang
    small = ang.abs() < 1e-6
    k = torch.empty_like(ang)
    k[~small] = torch.sin(half[~small]) / ang[~small]
    k[small] = 0.5 - (ang[small] * ang[small]) / 48
    q = torch.cat([torch.cos(half), aa * k], dim=-1)
    r, i, j, kk = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + kk * kk), two_s * (i * j - kk * r), two_s * (i * kk + j * r),
                     two_s * (i * j + kk * r), 1 - two_s * (i * i + kk * kk), two_s * (j * kk - i * r),
                     two_s * (i * kk - j * r), two_s * (j * kk + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def kabsch_batch(A: torch.Tensor, B: torch.Tensor):
    """R [b,3,3], t [b,3,1] minimising |R A + t - B| (A, B given as [b, N, 3])."""
    A, B = A.permute(0, 2, 1), B.permute(0, 2, 1)
    ca, cb = A.mean(dim=2, keepdim=True), B.mean(dim=2, keepdim=True)
    H = torch.bmm(A - ca, (B - cb).transpose(1, 2))
    U, S, Vt = torch.linalg.svd(H)
    R = torch.bmm(Vt.transpose(1, 2), U.transpose(1, 2))
    SS = torch.diag(torch.tensor([1., 1., -1.], dtype=A.dtype))
    Rm = torch.bmm(Vt.transpose(1, 2) @ SS, U.transpose(1, 2))
    R = torch.where(torch.linalg.det(R)[:, None, None] < 0, Rm, R)
    t = torch.bmm(-R, ca) + cb
    return R, t


def torsion_update_batch(pos: torch.Tensor, rot_edges: torch.Tensor, mask_rotate: torch.Tensor, tor: torch.Tensor):
    """pos [b,N,3]; rot_edges [R,2] (u,v); mask_rotate [R,N] bool; tor [b,R].  Sequential, order-dependent."""
    pos = pos + 0
    for r in range(rot_edges.shape[0]):
        u, v = int(rot_edges[r, 0]), int(rot_edges[r, 1])
        assert not mask_rotate[r, u] and mask_rotate[r, v]
        axis = pos[:, u] - pos[:, v]
        rm = axis_angle_to_matrix(axis / torch.linalg.norm(axis, dim=-1, keepdims=True) * tor[:, r:r + 1])
        m = mask_rotate[r]
        pos[:, m] = torch.bmm(pos[:, m] - pos[:, v:v + 1], rm.transpose(1, 2)) + pos[:, v:v + 1]
    return pos


def modify_conformer_batch(pos: torch.Tensor, cx: ComplexData, tr: torch.Tensor, rot: torch.Tensor, tor):
    """pos [b,N,3], tr/rot [b,3], tor [b*R] or None -> new pos [b,N,3]."""
    b = pos.shape[0]
    center = pos.mean(dim=1, keepdim=True)
    rm = axis_angle_to_matrix(rot)
    rigid = torch.bmm(pos - center, rm.permute(0, 2, 1)) + tr.unsqueeze(1) + center
    if tor is None:
        return rigid
    rot_edges = cx.lig_bond_index.T[cx.edge_mask]
    flex = torsion_update_batch(rigid, rot_edges, torch.from_numpy(np.asarray(cx.mask_rotate)), tor.reshape(b, -1))
    R, t = kabsch_batch(flex, rigid)
    return torch.bmm(flex, R.transpose(1, 2)) + t.transpose(1, 2)


def torsion_update_numpy(pos: np.ndarray, rot_edges: np.ndarray, mask_rotate: np.ndarray, tor: np.ndarray):
    """float64 numpy/scipy variant used by randomize_position (utils/torsion.py:48-72)."""
    pos = np.array(pos, dtype=np.float64, copy=True) if pos.dtype != np.float32 else np.array(pos, copy=True)
    for r, (u, v) in enumerate(rot_edges):
        if tor[r] == 0:
            continue
        axis = pos[u] - pos[v]
        axis = axis * tor[r] / np.linalg.norm(axis)
        rm = Rotation.from_rotvec(axis).as_matrix()
        pos[mask_rotate[r]] = (pos[mask_rotate[r]] - pos[v]) @ rm.T + pos[v]
    return pos


def get_t_schedule(inference_steps, alpha=1, beta_=1, t_max=1):
    lin_max = beta.cdf(t_max, a=alpha, b=beta_)
    c = np.linspace(lin_max, 0, inference_steps + 1)[:-1]
    return beta.ppf(c, a=alpha, b=beta_)


def sde_coefficients(t_idx, schedule, cfg: ScoreConfig, rot_schedule=None, tor_schedule=None):
    """(t, dt, sigma, g) for tr/rot/tor at step t_idx, scalars typed like the reference (sampling.py:94-135):
    t, dt, sigma are numpy float64; g is a 0-dim fp32 tensor.  With --different_schedules (inference.py:375-383) every component
    runs on its own time grid; t / dt are then 3-lists."""
    S = len(schedule)
    scheds = [schedule, schedule if rot_schedule is None else rot_schedule, schedule if tor_schedule is None else tor_schedule]
    ts = [sc[t_idx] for sc in scheds]
    dts = [sc[t_idx] - sc[t_idx + 1] if t_idx < S - 1 else sc[t_idx] for sc in scheds]
    sig = t_to_sigma(ts[0], ts[1], ts[2], cfg)
    lims = [(cfg.tr_sigma_min, cfg.tr_sigma_max), (cfg.rot_sigma_min, cfg.rot_sigma_max),
            (cfg.tor_sigma_min, cfg.tor_sigma_max)]
    g = [s * torch.sqrt(torch.tensor(2 * np.log(hi / lo))) for s, (lo, hi) in zip(sig, lims)]
    return ts, dts, sig, g


@torch.no_grad()
def sampling_ref(w, cx: ComplexData, pos0: torch.Tensor, schedule, cfg: ScoreConfig, so3_table, torus_table,
                 noise=None, no_final_step_noise=False, ode=False, record=False, temp_sampling=1.0, temp_psi=0.0,
                 temp_sigma_data=0.5, rot_schedule=None, tor_schedule=None, common_t_schedule=None):
    """Reverse diffusion for b poses of ONE complex: pos0 [b,N,3] -> final pos [b,N,3].
    common_t_schedule: the common time grid of a model with asyncronous_noise_schedule (utils/sampling.py:110-111).
    noise: dict of 'tr' [S,b,3], 'rot' [S,b,3], 'tor' [S,b*R] (explicit, lifted out of the reference's
    unseeded torch.normal calls, drawn in the reference's order) or None => zeros (no_random)."""
    S = len(schedule)
    b, R = pos0.shape[0], cx.R
    pos = pos0.clone().float()
    rec_cache = receptor_embedding(w, cx, cfg)
    trace = []
    for s in range(S):
        ts, dts, sig, g = sde_coefficients(s, schedule, cfg, rot_schedule, tor_schedule)
        out = score_forward(w, cx, pos, ts[0], ts[1], ts[2], cfg, so3_table, torus_table, rec_cache=rec_cache,
                            t_common=None if common_t_schedule is None else float(common_t_schedule[s]))
        tr_s, rot_s, tor_s = out["tr_pred"], out["rot_pred"], out["tor_pred"]
        last = (s == S - 1)
        zero = noise is None or (no_final_step_noise and last)
        z_tr = torch.zeros(b, 3) if zero else noise["tr"][s]
        z_rot = torch.zeros(b, 3) if zero else noise["rot"][s]
        if ode:
            tr_p = 0.5 * g[0] ** 2 * dts[0] * tr_s
            rot_p = 0.5 * rot_s * dts[1] * g[1] ** 2
        else:
            tr_p = g[0] ** 2 * dts[0] * tr_s + g[0] * np.sqrt(dts[0]) * z_tr
            rot_p = rot_s * dts[1] * g[1] ** 2 + g[1] * np.sqrt(dts[1]) * z_rot
        tor_p = None
        if not cfg.no_torsion and R > 0:
            z_tor = torch.zeros(b * R) if zero else noise["tor"][s]
            tor_p = 0.5 * g[2] ** 2 * dts[2] * tor_s if ode else g[2] ** 2 * dts[2] * tor_s + g[2] * np.sqrt(dts[2]) * z_tor
        # low-temperature sampling (utils/sampling.py:146-167): overrides the perturbation of every component whose
        # temperature differs from 1
        tsamp = list(temp_sampling) if np.iterable(temp_sampling) else [temp_sampling] * 3
        tp = list(temp_psi) if np.iterable(temp_psi) else [temp_psi] * 3
        lims = [(cfg.tr_sigma_min, cfg.tr_sigma_max), (cfg.rot_sigma_min, cfg.rot_sigma_max), (cfg.tor_sigma_min, cfg.tor_sigma_max)]
        scores, zs = [tr_s, rot_s, tor_s], [z_tr, z_rot, None if tor_p is None else z_tor]
        new = [tr_p, rot_p, tor_p]
        for k in range(3):
            if tsamp[k] != 1.0 and new[k] is not None and not ode:
                lo, hi = lims[k]
                sigma_data = np.exp(temp_sigma_data * np.log(hi) + (1 - temp_sigma_data) * np.log(lo))
                lam = (sigma_data + sig[k]) / (sigma_data + sig[k] / tsamp[k])
                new[k] = g[k] ** 2 * dts[k] * (lam + tsamp[k] * tp[k] / 2) * scores[k] + g[k] * np.sqrt(dts[k] * (1 + tp[k])) * zs[k]
        tr_p, rot_p, tor_p = new
        pos = modify_conformer_batch(pos, cx, tr_p.float(), rot_p.float(), None if tor_p is None else tor_p.float())
        if record:
            trace.append({"tr": tr_s.clone(), "rot": rot_s.clone(), "tor": tor_s.clone(), "pos": pos.clone()})
    return (pos, trace) if record else pos
