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


# ---------------------------------------------------------------------------------------------------------------------
# The entry points themselves (sampling_distributed, run_complex_set) end to end under world_size 2 / gloo.  The MI355X sampler
# cannot run here, so a stand-in with sampling()'s signature moves every pose by a function of ITS OWN noise columns and
# scores it: any mistake in the sample split, the noise slicing, the padding of uneven shares or the ranked gather shows up as
# a difference from the single-process result.
def _standin_sampling(data_list, model, inference_steps, tr_schedule, rot_schedule, tor_schedule, device, t_to_sigma, model_args,
                      no_random=False, ode=False, confidence_model=None, filtering_data_list=None, filtering_model_args=None,
                      batch_size=32, no_final_step_noise=False, noise=None, **kw):
    out, conf = [], []
    R = int(data_list[0]["ligand"].edge_mask.sum())
    for i, g in enumerate(data_list):
        shift = noise["tr"][:, i].sum(0) + 0.1 * noise["rot"][:, i].sum(0) + 0.01 * noise["tor"][:, i * R:(i + 1) * R].sum()
        g["ligand"].pos = g["ligand"].pos + shift
        out.append(g)
        conf.append(-g["ligand"].pos.norm())
    return out, (torch.stack(conf) if confidence_model is not None else None)


def _make_list(n):
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd import Batch
    import copy
    import numpy as np
    from confidence_bootstrapping_amd.sampling import randomize_position
    c = make_workload("tiny")
    torch.manual_seed(3)
    np.random.seed(3)
    dl = [Batch.from_data_list([copy.deepcopy(c)]) for _ in range(n)]
    randomize_position(dl, False, False, 5.0)
    return dl


def _sd_call(world, rank, n, with_conf):
    from argparse import Namespace
    from confidence_bootstrapping_amd.distributed import sampling_distributed
    import numpy as np
    dl = _make_list(n)
    sched = np.linspace(1, 0, 5)[:-1]
    torch.manual_seed(11)
    return sampling_distributed(dl, None, len(sched), sched, sched, sched, torch.device("cpu"), None, Namespace(no_torsion=False),
                                confidence_model=object() if with_conf else None, batch_size=3, world=world, rank=rank,
                                sampler=_standin_sampling)


def _sd_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = []
        for n, with_conf in ((7, True), (7, False), (1, True)):       # uneven split; no confidence model; one rank without samples
            r = _sd_call(None, None, n, with_conf)
            res.append(None if r is None else {k: (None if v is None else v.numpy().copy()) for k, v in r.items()})
        # complex set: LPT split of 5 complexes, groups of 2, per-complex result = (index, size)
        from confidence_bootstrapping_amd.distributed import run_complex_set
        from confidence_bootstrapping_amd.synthetic import make_complex
        cps = [make_complex(Nl=6 + i, Nr=20 + 3 * i, R=1, knn=6, seed=60 + i) for i in range(5)]
        seen = []

        def sample_group(items):
            seen.append([i for i, _ in items])
            return [{"i": i, "nl": int(c["ligand"].pos.shape[0]), "rank": rank} for i, c in items]
        rs = run_complex_set(cps, sample_group, group=2)
        q.put((rank, res, rs, seen))
    finally:
        dist.destroy_process_group()


def test_sampling_distributed_and_complex_set_world2_gloo():
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sd_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r is None for r in res[1][1]) and res[1][2] is None           # only dst returns
    for k, (n, with_conf) in enumerate(((7, True), (7, False), (1, True))):
        single = _sd_call(1, 0, n, with_conf)                                 # the same job in one process
        got = res[0][1][k]
        assert got["pos"].shape == (n, 12, 3)
        np.testing.assert_array_equal(got["index"], single["index"].numpy())
        np.testing.assert_allclose(got["pos"], single["pos"].numpy(), rtol=0, atol=0)
        if with_conf:
            np.testing.assert_array_equal(got["confidence"], single["confidence"].numpy())
            assert np.all(np.diff(got["confidence"]) <= 0)
        else:
            assert got["confidence"] is None and list(got["index"]) == list(range(n))
    rs = res[0][2]
    assert [r["i"] for r in rs] == list(range(5)) and [r["nl"] for r in rs] == [6, 7, 8, 9, 10]
    assert {r["rank"] for r in rs} == {0, 1}
    assert all(len(g) <= 2 for r in res for g in r[3])
    assert sorted(i for r in res for g in r[3] for i in g) == list(range(5))


# ---------------------------------------------------------------------------------------------------------------------
# Multi-rank fine-tuning must not hang when a rank skips a step: NaN loss on one rank, a batch of one on one rank, loaders of
# different lengths (each rank's CBBuffer has its own size).  train_epoch / train_step make the skip decision inside the gradient
# all-reduce; BatchNorm running statistics are averaged at the end of the epoch.
class _TinyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(4, 3)
        self.bn = torch.nn.BatchNorm1d(3)

    def forward(self, x):
        return self.bn(self.lin(x))


def _train_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from confidence_bootstrapping_amd.training import train_epoch
        torch.manual_seed(0)
        net = _TinyNet()
        opt = torch.optim.SGD(net.parameters(), lr=0.1)
        g = torch.Generator().manual_seed(100 + rank)
        mk = lambda n, tag: [{"x": torch.randn(4, generator=g), "tag": tag} for _ in range(n)]
        # rank 0: 4 batches (the 3rd has ONE graph); rank 1: 5 batches (the 2nd yields a NaN loss) -> 4 common steps, 2 of them skipped
        loader = [mk(3, "ok"), mk(3, "ok"), mk(1, "ok"), mk(3, "ok")] if rank == 0 else \
                 [mk(3, "ok"), mk(3, "nan"), mk(3, "ok"), mk(3, "ok"), mk(3, "ok")]
        steps = []

        def forward_fn(model, data):
            y = model(torch.stack([d["x"] for d in data]))
            return y, y, y[:, 0], None

        def loss_fn(tr, rot, tor, sc, data, t_to_sigma, device):
            loss = (tr ** 2).mean() * (float("nan") if data[0]["tag"] == "nan" else 1.0)
            steps.append(float(loss.detach()))
            z = torch.zeros(1)
            return (loss.reshape(1),) + (z,) * 10
        summary = train_epoch(net, loader, opt, torch.device("cpu"), None, loss_fn, None, forward_fn=forward_fn)
        q.put((rank, [p.detach().numpy().copy() for p in net.parameters()], net.bn.running_mean.numpy().copy(),
               net.bn.running_var.numpy().copy(), len(steps), summary["loss"]))
    finally:
        dist.destroy_process_group()


def test_train_epoch_skips_collectively_world2_gloo():
    import numpy as np
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_train_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])     # a hang shows up as queue.Empty
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(res[0][1], res[1][1]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-7)          # same averaged gradients applied on both ranks, skipped steps on both
    np.testing.assert_allclose(res[0][2], res[1][2], atol=1e-7)      # BatchNorm running statistics synchronised
    np.testing.assert_allclose(res[0][3], res[1][3], atol=1e-7)
    assert res[0][4] == 3 and res[1][4] == 4                         # rank 0 never evaluated its batch of one; rank 1 stopped after 4 steps
    assert np.isfinite(res[0][5]) and np.isfinite(res[1][5])
    torch.manual_seed(0)
    fresh = _TinyNet()
    assert not np.allclose(res[0][1][0], fresh.lin.weight.detach().numpy())      # the two good steps did update the weights


# ---------------------------------------------------------------------------------------------------------------------
# World 8 (the node the SCALE run uses), uneven shares: 37 samples over 8 ranks (5 ranks with five samples, 3 with four: padded rows in
# the ranked gather), 13 complexes of different cost over 8 ranks (LPT: ranks with one and with two complexes), groups of 2.
def _w8_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        r = _sd_call(None, None, 37, True)
        res = None if r is None else {k: (None if v is None else v.numpy().copy()) for k, v in r.items()}
        from confidence_bootstrapping_amd.distributed import run_complex_set, gather_ranked, shard_round_robin
        from confidence_bootstrapping_amd.synthetic import make_complex
        cps = [make_complex(Nl=5 + (i * 7) % 11, Nr=18 + (i * 5) % 13, R=1, knn=6, seed=80 + i) for i in range(13)]
        seen = []

        def sample_group(items):
            seen.append([i for i, _ in items])
            return [{"i": i, "nl": int(c["ligand"].pos.shape[0]), "rank": rank} for i, c in items]
        rs = run_complex_set(cps, sample_group, group=2)
        # the bare ranked gather with ties: every confidence equal -> the order must be the global sample order
        mine = shard_round_robin(37, world, rank)
        ids = torch.as_tensor(mine, dtype=torch.long)
        pos = ids.float()[:, None, None].expand(len(mine), 4, 3).contiguous()
        tied = gather_ranked(pos, torch.zeros(len(mine)), world, rank, 0, ids=ids, rows=-(-37 // world))
        q.put((rank, res, rs, seen, None if tied[0] is None else (tied[0].numpy().copy(), tied[2].numpy().copy())))
    finally:
        dist.destroy_process_group()


def test_world8_uneven_shares_gloo():
    import numpy as np
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_w8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] is None and r[2] is None and r[4] is None for r in res[1:])          # only dst returns
    single = _sd_call(1, 0, 37, True)
    got = res[0][1]
    assert got["pos"].shape == (37, 12, 3)
    np.testing.assert_array_equal(got["index"], single["index"].numpy())
    np.testing.assert_allclose(got["pos"], single["pos"].numpy(), rtol=0, atol=0)
    np.testing.assert_array_equal(got["confidence"], single["confidence"].numpy())
    rs = res[0][2]
    assert [r["i"] for r in rs] == list(range(13))
    per_rank = {k: sum(1 for r in rs if r["rank"] == k) for k in range(8)}
    assert sorted(per_rank.values()) == [1, 1, 1, 2, 2, 2, 2, 2]                          # LPT of 13 over 8
    assert all(len(g) <= 2 for r in res for g in r[3])
    tied_pos, tied_idx = res[0][4]
    assert tied_idx.tolist() == list(range(37)) and tied_pos[:, 0, 0].tolist() == [float(i) for i in range(37)]
