"""world_size-2 gloo tests of the sharding + gather layer (no GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import multiprocessing as mp

from confidence_bootstrapping_amd.distributed import shard_round_robin, shard_lpt, gather_poses, gather_ranked


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_round_robin(10, world, rank)
        pos = torch.arange(len(mine) * 4 * 3, dtype=torch.float32).reshape(len(mine), 4, 3) + 1000 * rank
        conf = torch.tensor([float(i) for i in mine])
        got = gather_poses(pos, world, rank)
        ranked, c = gather_ranked(pos, conf, world, rank)
        if rank == 0:
            q.put((mine, [g.numpy().copy() for g in got], ranked.numpy().copy(), c.numpy().copy()))
        else:
            q.put((mine, None, None, None))
    finally:
        dist.destroy_process_group()


def test_shards_cover_everything():
    for n, w in ((40, 8), (10, 3), (1, 4)):
        parts = [shard_round_robin(n, w, r) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    costs = [5, 1, 9, 3, 3, 7, 2, 8]
    parts = shard_lpt(costs, 3)
    assert sorted(sum(parts, [])) == list(range(8))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)


def test_gather_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    root = [r for r in res if r[1] is not None][0]
    _, got, ranked, conf = root
    assert len(got) == 2 and got[0].shape == (5, 4, 3) and float(got[1][0, 0, 0]) == 1000.0
    assert list(conf) == list(range(9, -1, -1))   # all 10 samples, descending confidence
    assert ranked.shape == (10, 4, 3)
    # sample with confidence 9 belongs to rank 1 (indices 1,3,5,7,9 -> local 4)
    assert float(ranked[0, 0, 0]) == 1000.0 + 4 * 12


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from confidence_bootstrapping_amd.training import allreduce_gradients
        torch.manual_seed(0)
        net = torch.nn.Sequential(torch.nn.Linear(5, 4), torch.nn.Linear(4, 2))
        x = torch.full((3, 5), float(rank + 1))
        net(x).sum().backward()
        net[1].bias.grad = None                      # a parameter without gradient on this rank counts as zero
        local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
        allreduce_gradients(net)
        q.put((rank, [None if g is None else g.numpy().copy() for g in local], [p.grad.numpy().copy() for p in net.parameters()]))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    """training.allreduce_gradients: one flat all-reduce, every rank ends with the mean of the per-rank gradients."""
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for k in range(4):
        a, b = res[0][1][k], res[1][1][k]
        mean = (np.zeros_like(res[0][2][k]) if a is None else a) / 2 + (np.zeros_like(res[0][2][k]) if b is None else b) / 2
        np.testing.assert_allclose(res[0][2][k], mean, rtol=1e-6)
        np.testing.assert_allclose(res[1][2][k], mean, rtol=1e-6)
