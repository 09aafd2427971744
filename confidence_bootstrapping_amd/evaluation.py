"""Evaluation of sampled poses (reference inference.py:500-548 per complex, :593-885 aggregate).

`pose_metrics`: per-pose symmetry-corrected RMSD (GPU kernel `cbd_symm_rmsd` via molecules_utils), centroid distance and smallest
intra-ligand distance.  `performance_metrics`: the reference's aggregate dictionary -- same keys, same rounding, and the same
selection rules, including the reference's own inconsistencies (e.g. the `reversefiltered_*centroid*` / `*self_intersect*`
entries index with the DESCENDING confidence order, inference.py:782-785, 812-819) so that numbers are comparable run to run.
Table-driven instead of the reference's 290 unrolled lines.
"""
from __future__ import annotations

import numpy as np
import torch

from .molecules_utils import get_symmetry_rmsd


def pose_metrics(ligand_pos, orig_ligand_pos, mol=None, device=None):
    """ligand_pos [N, Nl, 3] (heavy atoms), orig_ligand_pos [Nl, 3] or [K, Nl, 3] reference pose(s), same frame.
    Returns (rmsd [N], centroid_distance [N], min_self_distance [N]) like inference.py:505-548."""
    lp = np.asarray(ligand_pos, dtype=np.float32)
    ref = np.asarray(orig_ligand_pos, dtype=np.float32)
    ref = ref[None] if ref.ndim == 2 else ref
    if mol is not None:
        try:
            rmsd = np.min(np.asarray([get_symmetry_rmsd(mol, r, [l for l in lp], device=device) for r in ref]), axis=0)
        except Exception as e:
            print("Using non corrected RMSD because of the error:", e)
            mol = None
    if mol is None:
        rmsd = np.min(np.sqrt(((lp[None] - ref[:, None]) ** 2).sum(axis=3).mean(axis=2)), axis=0)
    centroid = np.min(np.linalg.norm(lp.mean(axis=1)[None] - ref.mean(axis=1)[:, None], axis=2), axis=0)
    t = torch.from_numpy(lp)
    d = torch.cdist(t, t)
    d = d + torch.diag_embed(torch.full((lp.shape[1],), float("inf")))[None]
    return rmsd, centroid, d.flatten(1).min(dim=1).values.numpy()


def _block(prefix, rmsd, centroid, self_dist=None):
    """the reference's standard group of entries for one selection of poses (one value per complex)"""
    pct = lambda a, thr: (100 * (a < thr).sum() / len(a)).__round__(2)
    out = {}
    if self_dist is not None:
        out[f"{prefix}self_intersect_fraction"] = pct(self_dist, 0.4)
    out.update({f"{prefix}rmsds_below_2": pct(rmsd, 2), f"{prefix}rmsds_below_5": pct(rmsd, 5)})
    out.update({f"{prefix}rmsds_percentile_{q}": np.percentile(rmsd, q).round(2) for q in (25, 50, 75)})
    out.update({f"{prefix}centroid_below_2": pct(centroid, 2), f"{prefix}centroid_below_5": pct(centroid, 5)})
    out.update({f"{prefix}centroid_percentile_{q}": np.percentile(centroid, q).round(2) for q in (25, 50, 75)})
    return out


def performance_metrics(rmsds, centroid_distances, min_self_distances, confidences=None, run_times=None, without_rec_overlap=None):
    """Aggregate metrics over complexes.  rmsds / centroid_distances / min_self_distances / confidences: [C, N] (N poses per
    complex, in sampling order); without_rec_overlap: optional [C] bool -> the `no_overlap_` copy of every entry."""
    R, Cd, S = (np.asarray(x, dtype=np.float64) for x in (rmsds, centroid_distances, min_self_distances))
    conf = None if confidences is None else np.asarray(confidences, dtype=np.float64)
    rt = np.asarray(run_times if run_times is not None else [0.0], dtype=np.float64)
    out = {}
    for overlap in ("", "no_overlap_"):
        if overlap:
            if without_rec_overlap is None or np.asarray(without_rec_overlap, dtype=bool).sum() == 0:
                continue
            m = np.asarray(without_rec_overlap, dtype=bool)
            r, c, s, cf = R[m], Cd[m], S[m], (None if conf is None else conf[m])
        else:
            r, c, s, cf = R, Cd, S, conf
        n_c, N = r.shape
        rows = np.arange(n_c)[:, None]
        out.update({f"{overlap}run_times_std": rt.std().__round__(2), f"{overlap}run_times_mean": rt.mean().__round__(2),
                    f"{overlap}mean_rmsd": r.mean(),
                    f"{overlap}rmsds_below_2": (100 * (r < 2).sum() / len(r) / N), f"{overlap}rmsds_below_5": (100 * (r < 5).sum() / len(r) / N)})
        out.update({f"{overlap}rmsds_percentile_{q}": np.percentile(r, q).round(2) for q in (25, 50, 75)})
        out.update({f"{overlap}min_rmsds_below_2": (100 * (np.min(r, axis=1) < 2).sum() / len(r)),
                    f"{overlap}min_rmsds_below_5": (100 * (np.min(r, axis=1) < 5).sum() / len(r)),
                    f"{overlap}mean_centroid": c.mean().__round__(2),
                    f"{overlap}centroid_below_2": (100 * (c < 2).sum() / len(c) / N).__round__(2),
                    f"{overlap}centroid_below_5": (100 * (c < 5).sum() / len(c) / N).__round__(2)})
        out.update({f"{overlap}centroid_percentile_{q}": np.percentile(c, q).round(2) for q in (25, 50, 75)})

        def best_of(order, k, r_order=None):
            """per complex: among the first k poses of `order`, the RMSD-best one (the RMSD values may come from another order:
            the reference's reverse-filtered entries take RMSDs from the ascending and everything else from the descending one)"""
            rr = r[rows, order][:, :k]
            pick = np.argsort(rr, axis=1)
            rm = np.min((r[rows, r_order] if r_order is not None else r[rows, order])[:, :k], axis=1)
            return rm, c[rows, order][:, :k][rows, pick][:, 0], s[rows, order][:, :k][rows, pick][:, 0]

        ident = np.tile(np.arange(N), (n_c, 1))
        for k in (5, 10):
            if N >= k:
                out.update(_block(f"{overlap}top{k}_", *best_of(ident, k)))
        if cf is not None:
            desc = np.argsort(cf, axis=1)[:, ::-1]
            asc = np.argsort(cf, axis=1)
            out.update(_block(f"{overlap}filtered_", r[rows, desc][:, 0], c[rows, desc][:, 0], s[rows, desc][:, 0]))
            for k in (5, 10):
                if N >= k:
                    rm, cc, _ = best_of(desc, k)
                    out.update(_block(f"{overlap}top{k}_filtered_", rm, cc))
            out.update(_block(f"{overlap}reversefiltered_", r[rows, asc][:, 0], c[rows, desc][:, 0], s[rows, desc][:, 0]))
            for k in (5, 10):
                if N >= k:
                    rm, cc, _ = best_of(desc, k, r_order=asc)
                    out.update(_block(f"{overlap}top{k}_reversefiltered_", rm, cc))
    return out
