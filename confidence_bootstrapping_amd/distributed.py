"""Multi-GPU layer: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm, "gloo" for
the CPU tests).  The reference has no distributed code on this path (inference is single-GPU, one complex at a time,
SURVEY.md section 5); pose samples and complexes are independent, so the work is sharded with NO collective inside the
step loop, and a single gather of the final poses with their confidences to one rank is the only exchange.

Entry points:
  * `sampling_distributed(...)`  -- BASELINE.json north star for ONE complex: the N pose samples are split round-robin over the
    ranks, every rank runs `sampling()` (+ the confidence model) on its share, one confidence-ranked gather
    (reference inference.py:537-547 ranks the samples of a complex by confidence) brings everything to `dst`.
  * `run_complex_set(...)`       -- configs[2] (a set of complexes): longest-processing-time partition of the complexes by
    Nl*Nr over the ranks, each rank samples its complexes in co-scheduled groups, one object gather of the per-complex results.
Both take the sampler as an argument (default: the MI355X `sampling()`), so the world_size-2 gloo tests drive them end to end
on CPU with a stand-in sampler while the GPU runs use the engine.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist
from .hostcfg import with_glue_threads


def world_rank(world: Optional[int] = None, rank: Optional[int] = None):
    if world is None or rank is None:
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(), dist.get_rank()
        return 1, 0
    return world, rank


def shard_round_robin(n_items: int, world: int, rank: int) -> List[int]:
    """Indices of the pose samples of ONE complex owned by `rank` (40 samples -> 5 per GPU at 8)."""
    return list(range(rank, n_items, world))


def shard_lpt(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time greedy partition of complexes by cost (e.g. Nl*Nr) -> per-rank index lists.
    Deterministic: ties broken by index."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        parts[r].append(i)
        load[r] += costs[i]
    return [sorted(p) for p in parts]


def gather_poses(pos: torch.Tensor, world: int, rank: int, dst: int = 0):
    """Gather equally-shaped pose tensors [b, Nl, 3] to `dst` (one small message per rank, latency bound)."""
    if world == 1 or not dist.is_initialized():
        return [pos]
    out = [torch.empty_like(pos) for _ in range(world)] if rank == dst else None
    dist.gather(pos, out, dst=dst)
    return out


def gather_ranked(pos: torch.Tensor, confidence: torch.Tensor, world: int, rank: int, dst: int = 0, ids: Optional[torch.Tensor] = None,
                  rows: Optional[int] = None):
    """Final confidence-ranked gather: all ranks send (confidence [b], pos [b,Nl,3]); `dst` returns the poses of all
    ranks sorted by descending confidence (inference.py:537-547 ranks the samples of a complex the same way).
    `ids` [b] (optional): global sample indices, returned in ranked order as a third value.  `rows`: rows every rank sends (ranks
    holding fewer -- an uneven round-robin split -- are padded and the padding is dropped on `dst`); default: b on every rank."""
    b = pos.shape[0]
    if ids is None:
        # no global sample indices given: number the rows rank-major (rank * rows + local row), so that the tie-break "by sample
        # index" is still a total order over ALL gathered rows and does not number every rank's rows 0..b-1
        off = rank * (b if rows is None else rows) if (world > 1 and dist.is_initialized()) else 0
        idv = torch.arange(b, dtype=torch.float64) + off
    else:
        idv = torch.as_tensor(ids, dtype=torch.float64)
    if world == 1 or not dist.is_initialized():
        order = torch.argsort(confidence, descending=True, stable=True)
        out = (pos[order], confidence[order])
        return out + (idv[order.cpu()].long(),) if ids is not None else out
    rows = b if rows is None else rows
    width = 2 + pos[0].numel() if b else 2 + int(pos.shape[1]) * 3
    payload = torch.zeros(rows, width, dtype=torch.float64, device=pos.device)
    payload[:, 0] = -1.0                                   # id -1 marks padding
    if b:
        payload[:b, 0] = idv.to(pos.device)
        payload[:b, 1] = confidence.reshape(-1).to(torch.float64)
        payload[:b, 2:] = pos.reshape(b, -1).to(torch.float64)
    out = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, out, dst=dst)
    if rank != dst:
        return (None, None, None) if ids is not None else (None, None)
    allp = torch.cat(out, dim=0)
    allp = allp[allp[:, 0] >= 0]
    gid, conf, flat = allp[:, 0].long(), allp[:, 1].to(confidence.dtype), allp[:, 2:].to(pos.dtype)
    # ties are broken by the global sample index, so the order does not depend on how the samples were split over ranks
    by_id = torch.argsort(gid, stable=True)
    gid, conf, flat = gid[by_id], conf[by_id], flat[by_id]
    order = torch.argsort(conf, descending=True, stable=True)
    res = (flat[order].reshape(-1, pos.shape[1], 3), conf[order])
    return res + (gid[order].cpu(),) if ids is not None else res


@with_glue_threads
def sampling_distributed(data_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device, t_to_sigma, model_args,
                         confidence_model=None, filtering_data_list=None, filtering_model_args=None, batch_size=32,
                         no_random=False, ode=False, no_final_step_noise=False, noise=None, world=None, rank=None, dst=0,
                         sampler: Optional[Callable] = None, **sampling_kw):
    """Reverse diffusion of the N pose samples of ONE complex over all ranks (BASELINE.json north star / SURVEY.md 8e).

    Every rank passes the SAME `data_list` (N randomised copies of the complex, same seed on every rank).  The samples are split
    round-robin, each rank calls `sampling()` on its share -- the N(0,1) noise is drawn for all N samples in the reference's order
    on every rank and sliced, so the poses do not depend on the number of ranks -- and the final poses are gathered to `dst`
    ranked by confidence (by sample index without a confidence model).  Returns on `dst` a dict
        {"pos": [N,Nl,3] ranked, "confidence": [N] ranked (or None), "index": [N] sample index of every ranked row};
    other ranks get None.  The only collective is that gather (RCCL over xGMI with the "nccl" backend)."""
    from .sampling import draw_noise_like_reference
    if sampler is None:
        from .sampling import sampling as sampler
    world, rank = world_rank(world, rank)
    if world > 1 and not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError(f"sampling_distributed(world={world}) needs an initialised torch.distributed process group "
                           "(init_process_group('nccl' | 'gloo') before the call)")
    N = len(data_list)
    if N == 0:
        raise ValueError("empty data_list")
    mine = shard_round_robin(N, world, rank)
    no_torsion = bool(getattr(model_args, "no_torsion", False))
    R = 0 if no_torsion else int(data_list[0]["ligand"].edge_mask.sum())
    if noise is None and not (no_random or ode):
        noise = draw_noise_like_reference(N, R, inference_steps, batch_size, no_final_step_noise)
    my_noise = None
    if noise is not None:
        idx = torch.as_tensor(mine, dtype=torch.long)
        cols = (idx[:, None] * R + torch.arange(R)[None, :]).reshape(-1) if R > 0 else None
        my_noise = {"tr": noise["tr"][:, idx], "rot": noise["rot"][:, idx], "tor": noise["tor"][:, cols] if R > 0 else None}
    Nl = int(data_list[0]["ligand"].pos.shape[0])
    if mine:
        my_list = [data_list[i] for i in mine]
        my_filt = None if filtering_data_list is None else [filtering_data_list[i] for i in mine]
        out_list, conf = sampler(my_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device, t_to_sigma, model_args,
                                 no_random=no_random, ode=ode, confidence_model=confidence_model, filtering_data_list=my_filt,
                                 filtering_model_args=filtering_model_args, batch_size=batch_size,
                                 no_final_step_noise=no_final_step_noise, noise=my_noise, **sampling_kw)
        pos = torch.stack([g["ligand"].pos.reshape(Nl, 3) for g in out_list]).float()
        for i, g in zip(mine, out_list):
            data_list[i] = g
    else:
        pos, conf = torch.zeros(0, Nl, 3), None
    ids = torch.as_tensor(mine, dtype=torch.long)
    has_conf = confidence_model is not None
    if has_conf and conf is not None and conf.dim() > 1:
        conf = conf[:, 0]            # rmsd_classification_cutoff lists: the first output ranks (inference.py:541-542)
    if has_conf:
        c = conf.reshape(-1).float().to(pos.device) if conf is not None else torch.zeros(0, device=pos.device)
    else:
        c = -ids.float().to(pos.device)                      # no confidence model: keep sample order
    if world > 1:
        # RCCL gathers device tensors, gloo host tensors (two ranks may share one GPU under gloo: tests/test_gpu_distributed.py)
        pos, c = (pos.to(device), c.to(device)) if dist.get_backend() == "nccl" else (pos.cpu(), c.cpu())
    rpos, rconf, rid = gather_ranked(pos, c, world, rank, dst, ids=ids, rows=-(-N // world))
    if rank != dst:
        return None
    return {"pos": rpos, "confidence": rconf if has_conf else None, "index": rid}


@with_glue_threads
def run_complex_set(complexes: Sequence, sample_group: Callable, world=None, rank=None, dst=0, group: int = 4,
                    cost: Optional[Callable] = None):
    """A set of complexes over all ranks (BASELINE.json configs[2]): LPT partition by `cost(complex)` (default Nl * Nr), every rank
    runs `sample_group([(index, complex), ...])` on consecutive groups of up to `group` of ITS complexes (the engine co-schedules a
    group in merged launches) and gets back one picklable result per complex; `dst` receives the results of all ranks ordered by
    complex index, other ranks None.  No collective besides the final object gather."""
    world, rank = world_rank(world, rank)
    if cost is None:
        cost = lambda c: float(c["ligand"].pos.shape[0]) * float(c["receptor"].pos.shape[0])
    parts = shard_lpt([cost(c) for c in complexes], world)
    mine = parts[rank]
    results = []
    g = max(int(group), 1)
    try:
        import inspect
        ahead = "next_items" in inspect.signature(sample_group).parameters      # a callee that can set the next group up meanwhile
    except (TypeError, ValueError):
        ahead = False
    for k in range(0, len(mine), g):
        idx = mine[k:k + g]
        items = [(i, complexes[i]) for i in idx]
        if ahead:
            out = sample_group(items, next_items=[(i, complexes[i]) for i in mine[k + g:k + 2 * g]] or None)
        else:
            out = sample_group(items)
        if len(out) != len(idx):
            raise RuntimeError("sample_group must return one result per complex")
        results.extend(zip(idx, out))
    if world == 1 or not dist.is_initialized():
        return [r for _, r in sorted(results, key=lambda t: t[0])]
    gathered = [None] * world if rank == dst else None
    dist.gather_object(results, gathered, dst=dst)
    if rank != dst:
        return None
    flat = sorted((t for part in gathered for t in part), key=lambda t: t[0])
    if [i for i, _ in flat] != list(range(len(complexes))):
        raise RuntimeError("complex set gather lost or duplicated entries")
    return [r for _, r in flat]
