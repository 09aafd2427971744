"""Multi-GPU layer: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm, "gloo" for
the CPU tests).  The reference has no distributed code on this path (inference is single-GPU, one complex at a time,
SURVEY.md section 5); pose samples and complexes are independent, so the work is sharded with NO collective inside the
step loop, and a single gather of the final poses (and, once the confidence model exists, their confidences) to
rank 0 is the only exchange.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


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


def gather_ranked(pos: torch.Tensor, confidence: torch.Tensor, world: int, rank: int, dst: int = 0):
    """Final confidence-ranked gather: all ranks send (confidence [b], pos [b,Nl,3]); `dst` returns the poses of all
    ranks sorted by descending confidence (inference.py:537-547 ranks the samples of a complex the same way)."""
    if world == 1 or not dist.is_initialized():
        order = torch.argsort(confidence, descending=True)
        return pos[order], confidence[order]
    payload = torch.cat([confidence.reshape(-1, 1), pos.reshape(pos.shape[0], -1)], dim=1).contiguous()
    out = [torch.empty_like(payload) for _ in range(world)] if rank == dst else None
    dist.gather(payload, out, dst=dst)
    if rank != dst:
        return None, None
    allp = torch.cat(out, dim=0)
    conf, flat = allp[:, 0], allp[:, 1:]
    order = torch.argsort(conf, descending=True, stable=True)
    return flat[order].reshape(-1, pos.shape[1], 3), conf[order]
