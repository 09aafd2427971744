"""`sampling()` and `randomize_position()` with the reference's signatures and semantics
(reference utils/sampling.py:15-48, 59-274), driving the MI355X engine.

`batch_size` keeps its meaning for the order and sizes of the random draws, but consecutive loader batches of the same
complex are executed as ONE engine batch (up to the engine capacity, 64 poses): pose samples never interact, results are
unchanged, and e.g. the reference's default of 10 poses per batch no longer turns into four small GPU launches per layer.

What stays on the host (as in the reference): batching the data list, the schedule scalars of every step
(engine.make_steps), drawing the N(0,1) noise in the reference's order, writing poses back into `data_list`.
What moves to the GPU as ONE call per batch: the whole step loop (score model + perturbation + pose update),
`cbd_sample` in include/cbdock.h -- no per-step host round trips, no `.item()` syncs.

Differences from the reference that are deliberate and documented:
  * noise for a partial last batch is drawn with the batch's real size b (the reference draws min(batch_size, N)
    rows and fails on the shape mismatch, relying on the caller's halve-and-retry);
  * the noise is drawn on the CPU generator (torch.normal, same call order and sizes as the reference: tr (b,3),
    rot (b,3), tor (b*R) per step), so a seed reproduces the reference's CPU path draw for draw;
  * SVGD, pivot, return_full_trajectory, return_features raise NotImplementedError (the last three also raise in the reference);
    crop_beyond of the SCORE model (not in the shipped yml) runs through _sample_cropped (a set-up per distinct crop mask and step).
Confidence scoring (reference utils/sampling.py:240-261): with `confidence_model` set, the final poses of every batch
are scored by the all-atom confidence engine (cbd_conf_score) -- crop_beyond per pose, t = 0 -- on the all-atom graphs
of `filtering_data_list` (or of `data_list` itself when it carries the 'atom' stores); returns the concatenated
confidences with NaN -> -1000 like the reference.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from .engine import DockEngine, make_steps, complex_fingerprint, _single_complex, _single_all_atom_complex
from .hostcfg import with_glue_threads



def _mask_rotate_of(graph):
    mr = graph["ligand"].mask_rotate
    while isinstance(mr, (list, tuple)):
        mr = mr[0]
    return np.asarray(mr)


def modify_conformer_torsion_angles(pos, edge_index, mask_rotate, torsion_updates, as_numpy=False):
    """Sequential torsion rotations on the host (reference utils/torsion.py:48-72); used by randomize_position only."""
    pos = pos.cpu().numpy().copy() if torch.is_tensor(pos) else np.array(pos, copy=True)
    edge_index = edge_index.cpu().numpy() if torch.is_tensor(edge_index) else np.asarray(edge_index)
    for idx_edge, (u, v) in enumerate(edge_index):
        if torsion_updates[idx_edge] == 0:
            continue
        rot_vec = pos[u] - pos[v]
        rot_vec = rot_vec * torsion_updates[idx_edge] / np.linalg.norm(rot_vec)
        rot_mat = R.from_rotvec(rot_vec).as_matrix()
        pos[mask_rotate[idx_edge]] = (pos[mask_rotate[idx_edge]] - pos[v]) @ rot_mat.T + pos[v]
    return pos if as_numpy else torch.from_numpy(pos.astype(np.float32))


def randomize_position(data_list, no_torsion, no_random, tr_sigma_max, pocket_knowledge=False, pocket_cutoff=7):
    """In-place initial pose randomisation; RNG use identical to the reference (numpy global for torsions,
    scipy Rotation.random, torch global for the translation)."""
    center_pocket = data_list[0]["receptor"].pos.mean(dim=0)
    if pocket_knowledge:
        cg = data_list[0]
        orig = cg["ligand"].orig_pos
        orig = orig[0] if isinstance(orig, (list, tuple)) else orig
        d = torch.cdist(cg["receptor"].pos, torch.from_numpy(np.asarray(orig)).float() - cg.original_center)
        label = torch.any(d < pocket_cutoff, dim=1)
        if torch.any(label):
            center_pocket = cg["receptor"].pos[label].mean(dim=0)
        else:
            center_pocket = cg["receptor"].pos[torch.argmin(torch.min(d, dim=1)[0])]
    if not no_torsion:
        for g in data_list:
            n_tor = int(g["ligand"].edge_mask.sum())
            torsion_updates = np.random.uniform(low=-np.pi, high=np.pi, size=n_tor)
            g["ligand"].pos = modify_conformer_torsion_angles(
                g["ligand"].pos, g["ligand", "ligand"].edge_index.T[g["ligand"].edge_mask], _mask_rotate_of(g), torsion_updates)
    for g in data_list:
        molecule_center = torch.mean(g["ligand"].pos, dim=0, keepdim=True)
        random_rotation = torch.from_numpy(R.random().as_matrix()).float()
        g["ligand"].pos = (g["ligand"].pos - molecule_center) @ random_rotation.T + center_pocket
        if not no_random:
            g["ligand"].pos += torch.normal(mean=0, std=tr_sigma_max, size=(1, 3))


def _draw_chunk_noise(b, R_, S, no_final_step_noise=False):
    """N(0,1) draws of one loader batch of b poses: per step tr (b,3), rot (b,3), tor (b*R) from the global CPU generator -- the
    order and sizes of the reference's torch.normal calls (utils/sampling.py:122-141)."""
    tr_l, rot_l, tor_l = [], [], []
    for s in range(S):
        last_quiet = no_final_step_noise and s == S - 1
        tr_l.append(torch.zeros(b, 3) if last_quiet else torch.normal(mean=0, std=1, size=(b, 3)))
        rot_l.append(torch.zeros(b, 3) if last_quiet else torch.normal(mean=0, std=1, size=(b, 3)))
        if R_ > 0:
            tor_l.append(torch.zeros(b * R_) if last_quiet else torch.normal(mean=0, std=1, size=(b * R_,)))
    return torch.stack(tr_l), torch.stack(rot_l), (torch.stack(tor_l) if R_ > 0 else None)


def draw_noise_like_reference(N, R_, S, batch_size, no_final_step_noise=False):
    """The noise `sampling()` would draw for N poses of one complex (R_ rotatable bonds) walked in loader batches of `batch_size`,
    as a dict of 'tr' [S,N,3], 'rot' [S,N,3], 'tor' [S,N*R_] (None if R_ == 0) CPU tensors: `sampling(..., noise=this)` then equals
    `sampling(...)` under the same seed.  `sampling_distributed` draws it on every rank and slices, so results do not depend
    on the number of ranks."""
    parts = [_draw_chunk_noise(min(int(batch_size), N - start), R_, S, no_final_step_noise) for start in range(0, N, max(int(batch_size), 1))]
    return {"tr": torch.cat([p[0] for p in parts], 1), "rot": torch.cat([p[1] for p in parts], 1),
            "tor": torch.cat([p[2] for p in parts], 1) if R_ > 0 else None}


def _sample_cropped(eng, cplx, key, pos, steps, noise, crop):
    """Reverse diffusion with the score model's `crop_beyond` (reference utils/sampling.py:101-108 with utils/utils.py:395-420): before
    every step each pose's receptor is cropped to the residues within 3 sigma_tr(t) + crop_beyond of a ligand atom, and the model sees
    that smaller complex -- other residues, other C-alpha edges, another receptor embedding.  Not in the shipped yml; implemented on the
    existing entry points: per step the poses are grouped by their crop mask (early steps: everything is kept, late steps: the poses
    of a converged run share a pocket), each group's cropped complex goes through cbd_set_complex and ONE step of cbd_sample.  The
    arithmetic is the engine's; what this path adds is host orchestration (a set-up per distinct mask and step)."""
    from .utils import crop_beyond as crop_graph
    z_tr, z_rot, z_tor = noise
    B, Nl = pos.shape[0], pos.shape[1]
    R = eng.R
    rec_pos = cplx["receptor"].pos.float().cpu()
    base_R = int(cplx["ligand"].edge_mask.sum())
    eng.set_option("graph", 0)
    try:
        for i in range(len(steps)):
            cutoff = steps[i].tr_sigma * 3 + crop
            host = pos.detach().cpu()
            d2 = torch.sum((host.unsqueeze(1) - rec_pos.view(1, -1, 1, 3)) ** 2, -1)            # [B, Nr, Nl], the reference's formula
            keep = torch.any(d2 < cutoff ** 2, dim=2)
            groups = {}
            for b in range(B):
                groups.setdefault(keep[b].numpy().tobytes(), []).append(b)
            for mask_bytes, idx in groups.items():
                m = keep[idx[0]]
                if not bool(m.any()):
                    raise RuntimeError("crop_beyond left a pose without any receptor residue (the reference's model fails on an empty "
                                       "receptor graph as well)")
                g = cplx.shallow_copy()
                g["ligand"].pos = host[idx[0]]
                crop_graph(g, cutoff, False)
                eng.set_complex(g, (key, "crop", hash(mask_bytes)))
                sel = torch.as_tensor(idx, device=pos.device)
                p = pos.index_select(0, sel).contiguous()
                cols = (sel[:, None] * base_R + torch.arange(base_R, device=pos.device)[None, :]).reshape(-1)
                take = lambda z, c=None: None if z is None else z[i:i + 1].to(pos.device).index_select(1, sel if c is None else c).contiguous()
                eng.sample(p, (type(steps[i]) * 1)(steps[i]), take(z_tr), take(z_rot), take(z_tor, cols) if (R > 0 and z_tor is not None) else None)
                pos.index_copy_(0, sel, p)
    finally:
        eng.set_option("graph", 1)
        eng.complex_key = None          # the engine holds a cropped complex now


@with_glue_threads
def sampling(data_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device, t_to_sigma, model_args,
             no_random=False, ode=False, visualization_list=None, confidence_model=None, filtering_data_list=None,
             filtering_model_args=None, asyncronous_noise_schedule=False, t_schedule=None, batch_size=32,
             no_final_step_noise=False, pivot=None, return_full_trajectory=False, temp_sampling=1.0, temp_psi=0.0,
             temp_sigma_data=0.5, return_features=False, svgd_weight_log_0=None, svgd_repulsive_weight_log_0=None,
             svgd_weight_log_1=None, svgd_repulsive_weight_log_1=None, svgd_kernel_size_log_0=None,
             svgd_kernel_size_log_1=None, svgd_langevin_weight_log_0=None, svgd_langevin_weight_log_1=None,
             svgd_rot_log_rel_weight=0.0, svgd_tor_log_rel_weight=0.0, svgd_use_x0=False, noise=None, n_streams=1, co_schedule=None):
    """Reverse diffusion of every pose in `data_list`; returns (data_list, confidence) like the reference.
    `noise` (optional, extension): dict of pre-drawn 'tr' [S,N,3], 'rot' [S,N,3], 'tor' [S,N*R] CPU tensors.
    `n_streams` (extension): each batch is split over this many concurrent HIP streams (identical results).
    `co_schedule` (extension): when `data_list` holds poses of several complexes, up to this many (<= 8) complexes are advanced in
    lockstep with merged tensor-product launches (identical results; 1 = one complex at a time like the reference; default: as many
    as bring a launch to ~160 poses)."""
    N = len(data_list)
    assert not (return_full_trajectory or return_features or pivot), "Not implemented yet in new inference version"
    if svgd_weight_log_0 is not None and svgd_weight_log_1 is not None:
        raise NotImplementedError("SVGD sampling (O(B^2) host loop in the reference) is outside the MI355X hot path")
    # asyncronous_noise_schedule (inference.py:384-388, utils/diffusion_utils.py:172-175): the caller passes the common time grid as
    # `t_schedule`, the three component schedules are its beta-quantile images; what the flag changes is which time the MODEL embeds,
    # and that is a property of the model (score_model.py:85) -- a model built without it ignores complex_t['t'] in the reference too
    model_async = bool(getattr(getattr(model, "module", model), "asyncronous_noise_schedule", False))
    if model_async and (t_schedule is None or not asyncronous_noise_schedule):
        raise KeyError("'t': the model embeds the common diffusion time (asyncronous_noise_schedule) -- pass t_schedule and "
                       "asyncronous_noise_schedule=True, as inference.py does")
    if t_schedule is not None and len(t_schedule) != inference_steps:
        raise ValueError("t_schedule length != inference_steps")
    score_crop = getattr(model_args, "crop_beyond", None)
    if score_crop is not None and getattr(model_args, "all_atoms", False):
        raise NotImplementedError("score-model crop_beyond is supported for the C-alpha score model (the shipped architecture)")
    conf_model = getattr(confidence_model, "module", confidence_model)
    if conf_model is not None and not hasattr(conf_model, "atom_confidence_predictor"):
        raise NotImplementedError("confidence scoring runs on the all-atom confidence engine (all_atom_score_model)")
    if filtering_data_list is not None and len(filtering_data_list) != N:
        raise ValueError("filtering_data_list must have one graph per pose")
    confidence = []
    tr_schedule, rot_schedule, tor_schedule = (np.asarray(s, dtype=np.float64) for s in (tr_schedule, rot_schedule, tor_schedule))
    if not (len(tr_schedule) == len(rot_schedule) == len(tor_schedule) == inference_steps):
        raise ValueError("schedule length != inference_steps")
    device = torch.device(device)
    model = getattr(model, "module", model)
    eng = model.engine_pool(n_streams=n_streams, max_batch=max(int(batch_size), 1)) if n_streams > 1 else model.engine()
    # --different_schedules (inference.py:375-383): rot / tor run on their own time grids; everything schedule-dependent is a host
    # scalar of the step (engine.make_steps), the engine itself is agnostic
    steps = make_steps(tr_schedule, model_args, model.timestep_emb_func, ode=ode, no_random=no_random,
                       no_final_step_noise=no_final_step_noise, temp_sampling=temp_sampling, temp_psi=temp_psi,
                       temp_sigma_data=temp_sigma_data, rot_schedule=rot_schedule, tor_schedule=tor_schedule,
                       common_t_schedule=t_schedule if model_async else None)
    S = inference_steps
    use_noise = not (no_random or ode)
    if co_schedule is None:
        # aim at ~160 poses per merged launch -- 4 complexes of 40 samples, 8 of 8 (measured: 8-way is +6 % over 4-way at 8 samples per
        # complex and -1 % at 40).  Poses per complex = the leading run of equal names (the loader batches of a complex are merged).
        name0 = getattr(data_list[0], "name", None) if N else None
        per = next((i for i, d in enumerate(data_list) if getattr(d, "name", None) != name0), N) if name0 is not None else int(batch_size)
        co_schedule = -(-160 // max(per, 1))
    n_co = max(1, min(int(co_schedule), 8)) if n_streams == 1 else 1
    if score_crop is not None:
        n_co = 1            # the cropped receptor differs from pose to pose and step to step: one complex at a time (_sample_cropped)
    offset = 0
    pending = []          # (first pose index, b, pos [b,Nl,3] CPU, z_tr, z_rot, z_tor, loader batch)
    pending_key = None

    groups = []           # closed groups (one complex each) waiting to be advanced together
    conf_engines_used = set()

    def flush():
        """close the pending group (consecutive loader batches of one complex); run when `co_schedule` groups are waiting"""
        nonlocal pending, pending_key
        if not pending:
            return
        groups.append((pending, pending_key))
        pending, pending_key = [], None
        if len(groups) >= n_co:
            run_groups()

    def run_groups():
        """Up to eight complexes advance in lockstep (cbd_sample_multi: their tensor-product launches are merged, so a launch
        carries several times the waves -- +24 % poses/s at 8 samples per complex); one group runs on cbd_sample.  Results are
        bitwise those of separate calls (tests/test_gpu_parity.py::test_sample_pair_equals_two_samples)."""
        if not groups:
            return
        engines = [eng] + (model.co_engines(len(groups) - 1, eng) if len(groups) > 1 else [])
        work = []
        for (pend, key), e in zip(groups, engines):
            batch0 = pend[0][6]           # first graph of the group: all of them are poses of the same complex
            g, _, _ = _single_complex(batch0)
            if e.complex_key != key:
                e.set_complex(g, key)
            R_ = e.R if not model_args.no_torsion else 0
            pos = torch.cat([p[2] for p in pend], dim=0).to(device, torch.float32).contiguous()
            cat = lambda k, dim: None if pend[0][k] is None else torch.cat([p[k] for p in pend], dim=dim)
            work.append((pend, e, pos, (cat(3, 1), cat(4, 1), cat(5, 1) if R_ > 0 else None), batch0))
        if score_crop is not None:
            for (pend, key), (_, e, pos, nz, batch0) in zip(groups, work):
                _sample_cropped(e, _single_complex(batch0)[0], key, pos, steps, nz, float(score_crop))
        elif len(work) == 1:
            work[0][1].sample(work[0][2], steps, *work[0][3])
        else:
            DockEngine.sample_multi([w[1] for w in work], [w[2] for w in work], steps, [w[3] for w in work])
        for pend, e, pos, _, batch0 in work:
            first, B, Nl = pend[0][0], pos.shape[0], pos.shape[1]
            flat = pos.reshape(B * Nl, 3)
            for i in range(B):
                data_list[first + i]["ligand"].pos = flat[i * Nl:(i + 1) * Nl]
            if conf_model is not None:
                if filtering_data_list is not None:
                    fbatch = filtering_data_list[first]
                    crop = getattr(filtering_model_args, "crop_beyond", None)
                else:
                    fbatch, crop = batch0, None
                fg, _, fNl = _single_all_atom_complex(fbatch)
                if fNl != Nl:
                    raise RuntimeError("filtering graphs hold a different ligand than the sampled ones")
                ceng = conf_model.engine(max_batch=eng.max_batch)
                ckey = complex_fingerprint(fbatch) + (fg["atom"].pos.shape[0],)
                if ceng.complex_key != ckey:
                    ceng.set_complex(fg, ckey)
                # no host sync per complex: the capacity flag of the confidence engine is sticky, checked once at the end
                confidence.append(ceng.score(pos, crop, check=False)[0])
                conf_engines_used.add(ceng)
        groups.clear()

    def drop_stale_capacity_flags(exc_type):
        """The confidence engine's capacity flag is sticky and lives on the (cached) engine: if this call ends by an exception
        before the final check(), the flag must not survive into the next, unrelated call."""
        if exc_type is not None:
            for ceng in conf_engines_used:
                try:
                    ceng.check()
                except RuntimeError:
                    pass

    class _Guard:
        def __enter__(self):
            return self

        def __exit__(self, exc_type, exc, tb):
            drop_stale_capacity_flags(exc_type)
            return False

    with torch.no_grad(), _Guard():
        # The reference collates every chunk with torch_geometric's Batch (utils/sampling.py:78); only the ligand coordinates of
        # the poses and ONE copy of the complex are needed here, so the chunks are walked without collating (for a 40-pose
        # complex the collation of the 1281-wide receptor features alone costs more host time than the GPU work).
        for start in range(0, N, max(int(batch_size), 1)):
            chunk = data_list[start:start + max(int(batch_size), 1)]
            b = len(chunk)
            if b > eng.max_batch:
                raise RuntimeError(f"batch of {b} exceeds the engine capacity {eng.max_batch}")
            if any(getattr(d, "num_graphs", 1) != 1 for d in chunk):
                raise ValueError("data_list elements must be single graphs (as inference.py builds them)")
            batch = chunk[0]
            Nl = batch["ligand"].pos.shape[0]
            key = complex_fingerprint(batch)
            for d in chunk[1:]:
                if complex_fingerprint(d) != key:
                    raise NotImplementedError("all poses of one batch must belong to the same complex (as modify_conformer_batch assumes)")
            if pending and (key != pending_key or sum(p[1] for p in pending) + b > eng.max_batch):
                flush()
            R_ = int(batch["ligand"].edge_mask.sum()) if not model_args.no_torsion else 0
            z_tr = z_rot = z_tor = None
            if use_noise:
                if noise is not None:
                    z_tr = noise["tr"][:, offset:offset + b]
                    z_rot = noise["rot"][:, offset:offset + b]
                    z_tor = noise["tor"][:, offset * R_:(offset + b) * R_] if R_ > 0 else None
                else:
                    z_tr, z_rot, z_tor = _draw_chunk_noise(b, R_, S, no_final_step_noise)
            lig_pos = torch.stack([d["ligand"].pos.detach().cpu().float().reshape(Nl, 3) for d in chunk])
            pending.append((offset, b, lig_pos, z_tr, z_rot, z_tor, batch))
            pending_key = key
            offset += b
        flush()
        run_groups()
        if visualization_list is not None:
            for idx, visualization in enumerate(visualization_list):
                visualization.add((data_list[idx]["ligand"].pos.detach().cpu() + data_list[idx].original_center.detach().cpu()),
                                  part=1, order=2)
    if conf_model is not None:
        for ceng in conf_engines_used:
            ceng.check()          # raises if any of the batches above exceeded a per-atom edge capacity
        confidence = torch.nan_to_num(torch.cat(confidence, dim=0), nan=-1000)
        return data_list, confidence
    return data_list, None
