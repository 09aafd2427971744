"""Training op (csrc/tp_train.hip via cbd_tp_forward / cbd_tp_backward) against autograd through the oracle's restatement of the
reference FasterTensorProduct + last FCBlock Linear (oracle/score_ref.py::faster_tensor_product, models/tensor_layers.py:66-117)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LEVELS = [(0, 1), (1, 2), (2, 3), (3, 3)]


@pytest.mark.parametrize("lv", LEVELS)
@pytest.mark.parametrize("E", [1, 37, 300])
def test_tp_op_forward_backward_matches_oracle(lv, E):
    from confidence_bootstrapping_amd.score_model import FCBlock, faster_tp_weight_numel, get_irrep_seq
    from confidence_bootstrapping_amd.train_ops import stream_map, LEVEL_DIMS
    from experiments.train_ops_reference import tensor_product
    from oracle import score_ref as sr
    IN, OUT = lv
    seq = get_irrep_seq(32, 6, False, True)
    W = faster_tp_weight_numel(seq[IN], seq[OUT])
    torch.manual_seed(100 * IN + E)
    fc = FCBlock(96, 96, W, 0.0)
    din, dout = LEVEL_DIMS[IN], LEVEL_DIMS[OUT]
    x = torch.randn(E, din)
    vec = F.normalize(torch.randn(E, 3), dim=-1)
    h = torch.relu(torch.randn(E, 96))
    gout = torch.randn(E, dout)

    # oracle (CPU autograd)
    xo, ho = x.clone().requires_grad_(), h.clone().requires_grad_()
    w2, b2 = fc[3].weight.detach().clone().requires_grad_(), fc[3].bias.detach().clone().requires_grad_()
    sh = torch.cat([torch.ones(E, 1), np.sqrt(3.0) * vec], 1)
    ref = sr.faster_tensor_product(xo, sh, F.linear(ho, w2, b2), sr.IRREP_SEQ[IN], sr.IRREP_SEQ[OUT])
    (ref * gout).sum().backward()

    dev = torch.device("cuda:0")
    fcd = FCBlock(96, 96, W, 0.0).to(dev)
    fcd.load_state_dict(fc.state_dict())
    xd = F.pad(x, (0, 80 - din)).to(dev).requires_grad_()
    hd = h.to(dev).requires_grad_()
    vd = F.pad(vec, (0, 1)).to(dev)
    msg = tensor_product(xd, vd, hd, stream_map(IN, OUT).stream(fcd), IN, OUT)
    assert float(msg.detach()[:, dout:].abs().max()) == 0.0
    (msg[:, :dout] * gout.to(dev)).sum().backward()

    def close(a, b, what):
        a, b = a.cpu(), b.cpu()
        tol = 2e-5 * float(b.abs().max()) + 1e-6
        assert float((a - b).abs().max()) <= tol, (what, float((a - b).abs().max()), tol)

    close(msg[:, :dout].detach(), ref.detach(), "msg")
    close(xd.grad[:, :din], xo.grad, "gx")
    assert float(xd.grad[:, din:].abs().max()) == 0.0
    close(hd.grad, ho.grad, "gh")
    close(fcd[3].weight.grad, w2.grad, "gW2")
    close(fcd[3].bias.grad, b2.grad, "gb2")
    assert fcd[0].weight.grad is None or float(fcd[0].weight.grad.abs().max()) == 0.0


def test_tp_op_edge_groups_in_one_launch():
    """Three edge groups with their own FCBlocks (sizes 45 / 0 is skipped by the caller / 7 / 100) in one launch == per-group oracle."""
    from confidence_bootstrapping_amd.score_model import FCBlock, faster_tp_weight_numel, get_irrep_seq
    from confidence_bootstrapping_amd.train_ops import stream_map
    from experiments.train_ops_reference import tensor_product
    from oracle import score_ref as sr
    IN, OUT = 3, 3
    seq = get_irrep_seq(32, 6, False, True)
    W = faster_tp_weight_numel(seq[IN], seq[OUT])
    torch.manual_seed(5)
    sizes = [45, 7, 100]
    fcs = [FCBlock(96, 96, W, 0.0) for _ in sizes]
    E = sum(sizes)
    x = torch.randn(E, 74)
    vec = F.normalize(torch.randn(E, 3), dim=-1)
    h = torch.relu(torch.randn(E, 96))
    gout = torch.randn(E, 74)
    sh = torch.cat([torch.ones(E, 1), np.sqrt(3.0) * vec], 1)
    xo, ho = x.clone().requires_grad_(), h.clone().requires_grad_()
    refs, lo = [], 0
    for fc, n in zip(fcs, sizes):
        refs.append(sr.faster_tensor_product(xo[lo:lo + n], sh[lo:lo + n], fc[3](ho[lo:lo + n]), sr.IRREP_SEQ[IN], sr.IRREP_SEQ[OUT]))
        lo += n
    ref = torch.cat(refs)
    (ref * gout).sum().backward()
    dev = torch.device("cuda:0")
    fcd = [FCBlock(96, 96, W, 0.0).to(dev) for _ in sizes]
    for a, b in zip(fcd, fcs):
        a.load_state_dict(b.state_dict())
    xd = F.pad(x, (0, 6)).to(dev).requires_grad_()
    hd = h.to(dev).requires_grad_()
    sm = stream_map(IN, OUT)
    msg = tensor_product(xd, F.pad(vec, (0, 1)).to(dev), hd, [sm.stream(f) for f in fcd], IN, OUT, sizes)
    (msg[:, :74] * gout.to(dev)).sum().backward()

    def close(a, b, what):
        a, b = a.detach().cpu(), b.detach().cpu()
        assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max()) + 1e-6, what

    close(msg[:, :74], ref, "msg")
    close(xd.grad[:, :74], xo.grad, "gx")
    close(hd.grad, ho.grad, "gh")
    for a, b in zip(fcd, fcs):
        close(a[3].weight.grad, b[3].weight.grad, "gW2")
        close(a[3].bias.grad, b[3].bias.grad, "gb2")


@pytest.mark.parametrize("E", [1, 63, 5001, 70000])
def test_first_linear_weight_gradient_kernel(E):
    """cbd_outer_accum (dW = G^T X, db = column sums of G over E edges) through FirstLinearFn against torch autograd in fp64."""
    from experiments.train_ops_reference import first_linear
    dev = torch.device("cuda:0")
    torch.manual_seed(E)
    lin = torch.nn.Linear(96, 96).to(dev)
    x = torch.randn(E, 96, device=dev, requires_grad=True)
    gout = torch.randn(E, 96, device=dev)
    y = first_linear(x, lin)
    (y * gout).sum().backward()
    lin64 = torch.nn.Linear(96, 96).double().to(dev)
    lin64.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    x64 = x.detach().double().requires_grad_()
    y64 = lin64(x64)
    (y64 * gout.double()).sum().backward()
    rel = lambda a, b: float((a.detach().double() - b.detach()).abs().max() / b.detach().abs().max())
    assert rel(y, y64) < 1e-5 and rel(x.grad, x64.grad) < 1e-5
    assert rel(lin.weight.grad, lin64.weight.grad) < 2e-5 and rel(lin.bias.grad, lin64.bias.grad) < 2e-5


def test_stream_hub_path_equals_the_per_block_streams():
    """The training forward packs the tile streams of all FCBlocks once per step (train_ops.StreamHub) and returns their gradients
    through one buffer; the per-block form (`StreamMap.stream` + TensorProductFn, checked against the oracle above) must give the same
    messages and the same gradients for x, h and every fc[3] parameter -- blocks of two different irreps levels, one of them unused
    in the step (its gradient must be zero, not garbage)."""
    from confidence_bootstrapping_amd.score_model import FCBlock, faster_tp_weight_numel, get_irrep_seq
    from confidence_bootstrapping_amd.train_ops import stream_map, StreamHub, TensorProductHubFn
    from experiments.train_ops_reference import tensor_product
    seq = get_irrep_seq(32, 6, False, True)
    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    levels = [(3, 3), (3, 3), (1, 2), (3, 3)]                 # block 3 takes no part in the step
    fcs = [FCBlock(96, 96, faster_tp_weight_numel(seq[i], seq[o]), 0.0).to(dev) for i, o in levels]
    hub = StreamHub([(fc, i, o) for fc, (i, o) in zip(fcs, levels)], dev)
    calls = [((3, 3), [0, 1], [300, 41]), ((1, 2), [2], [77])]
    data = []
    for (i, o), blocks, sizes in calls:
        E = sum(sizes)
        data.append((torch.randn(E, 80, device=dev), F.pad(F.normalize(torch.randn(E, 3, device=dev), dim=-1), (0, 1)),
                     torch.relu(torch.randn(E, 96, device=dev)), torch.randn(E, 80, device=dev)))

    def run(use_hub):
        for fc in fcs:
            fc.zero_grad(set_to_none=True)
        if use_hub:
            hub.pack()
        outs = []
        for ((i, o), blocks, sizes), (x, v, h, gout) in zip(calls, data):
            x, h = x.clone().requires_grad_(), h.clone().requires_grad_()
            if use_hub:
                msg = TensorProductHubFn.apply(x, v, h, hub.big, hub, i, o, tuple(sizes), tuple(blocks))
            else:
                msg = tensor_product(x, v, h, [stream_map(i, o).stream(fcs[b]) for b in blocks], i, o, sizes)
            outs.append((msg, x, h, gout))
        sum((m * g).sum() for m, _, _, g in outs).backward()
        return ([m.detach() for m, _, _, _ in outs], [x.grad for _, x, _, _ in outs], [h.grad for _, _, h, _ in outs],
                [(fc[3].weight.grad.clone() if fc[3].weight.grad is not None else None, fc[3].bias.grad.clone() if fc[3].bias.grad is not None else None)
                 for fc in fcs])

    a, b = run(True), run(False)
    for k in range(2):
        for t, u in zip(a[k], b[k]):
            assert torch.equal(t, u)
    # g_h: the hub path re-forms g_w tile by tile and multiplies with the transposed weight tiles on the matrix cores
    # (cbd_tp_backward_gh); the per-block path is a library GEMM on the stored g_w -- same sum, another order
    for t, u in zip(a[2], b[2]):
        assert float((t - u).abs().max()) <= 2e-6 * float(u.abs().max()) + 1e-7
    for blk, ((wa, ba), (wb, bb)) in enumerate(zip(a[3], b[3])):
        if blk == 3:
            assert float(wa.abs().max()) == 0.0 and float(ba.abs().max()) == 0.0 and wb is None
            continue
        assert float((wa - wb).abs().max()) <= 2e-6 * float(wb.abs().max()) and float((ba - bb).abs().max()) <= 2e-6 * float(bb.abs().max()), blk


def test_fused_score_loss_equals_the_torch_form():
    """cbd_score_loss (training.loss_from_targets with train_ops.FUSED_LOSS) against the torch-op form of the same function (reference
    utils/training.py:17-126, apply_mean=True): the 11 values to 1e-6 relative, the gradients of the three predictions to 1e-6 of their
    largest entry; a batch without rotatable bonds gives a NaN loss in both; no_torsion models get a zero torsion term."""
    import confidence_bootstrapping_amd.train_ops as to
    from confidence_bootstrapping_amd.training import loss_from_targets
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    B, T = 7, 43
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    tg = {"tr_score": r(B, 3), "tr_sigma": (r(B, 1).abs() + 0.3), "rot_score": r(B, 3), "rot_score_norm": (r(B, 1).abs() + 0.5),
          "tor_score": r(T), "tor_score_norm2": (r(T).abs() + 0.2)}

    def run(fused, T_, no_torsion=False):
        preds = [r(B, 3).requires_grad_(), r(B, 3).requires_grad_(), r(T_).requires_grad_()]
        t2 = dict(tg, tor_score=tg["tor_score"][:T_], tor_score_norm2=tg["tor_score_norm2"][:T_])
        to.FUSED_LOSS = fused
        try:
            out = loss_from_targets(preds[0], preds[1], None if no_torsion else preds[2], t2, 0.4, 0.35, 0.25, True, no_torsion)
        finally:
            to.FUSED_LOSS = True
        (out[0] * 1.7).sum().backward()
        return [float(o.detach()) for o in out], [p.grad for p in preds]
    for T_, no_tor in ((T, False), (T, True), (0, False)):
        g.manual_seed(11)
        va, ga = run(True, T_, no_tor)
        g.manual_seed(11)
        vb, gb = run(False, T_, no_tor)
        for x, y in zip(va, vb):
            assert (np.isnan(x) and np.isnan(y)) or abs(x - y) <= 1e-6 * max(abs(y), 1e-3), (T_, no_tor, va, vb)
        if T_ > 0:
            for k, (x, y) in enumerate(zip(ga, gb)):
                if k == 2 and no_tor:
                    assert x is None and y is None
                    continue
                assert float((x - y).abs().max()) <= 1e-6 * float(y.abs().max()), (T_, no_tor, k)
        else:
            assert np.isnan(va[0]) and np.isnan(vb[0])


def test_fused_heads_equal_the_torch_forms():
    """cbd_center_tp_* / cbd_bond_tp_* (train_ops.CenterTpFn / BondTpFn) against the torch-op forms of the two e3nn heads
    (train_forward.center_tensor_product / bond_tensor_product, which the reference's training step g11 pins): outputs and the gradients
    with respect to the node rows and the per-edge weights to 2e-6 of their largest entry; the padding columns of wider rows get zero
    gradient; a zero direction vector is handled like F.normalize does (no NaN)."""
    import confidence_bootstrapping_amd.train_ops as to
    import confidence_bootstrapping_amd.train_forward as tf
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)

    def both(fn, make):
        res = []
        for fused in (True, False):
            args = make()
            to.FUSED_HEADS = fused
            try:
                out = fn(*args)
            finally:
                to.FUSED_HEADS = True
            gout = make.gout
            (out * gout).sum().backward()
            res.append((out.detach(), [a.grad for a in args if a.requires_grad]))
        (oa, ga), (ob, gb) = res
        assert float((oa - ob).abs().max()) <= 2e-6 * float(ob.abs().max())
        for x, y in zip(ga, gb):
            assert torch.isfinite(x).all() and float((x - y).abs().max()) <= 2e-6 * float(y.abs().max())
        return ga

    for n, ldx in ((133, 74), (5, 80)):
        x0, vec, w0 = r(n, ldx), r(n, 3), r(n, 124)
        vec[0] = 0.0                                   # F.normalize: x / max(|x|, 1e-12) -> zero direction

        def make():
            return [x0.clone().requires_grad_(), vec, w0.clone().requires_grad_()]
        make.gout = r(n, 12)
        ga = both(tf.center_tensor_product, make)
        if ldx > 74:
            assert float(ga[0][:, 74:].abs().max()) == 0.0
    for n in (257, 3):
        x0, ev, bv, w0 = r(n, 74), r(n, 3), r(n, 3), r(n, 384)

        def make():
            return [x0.clone().requires_grad_(), ev, bv, w0.clone().requires_grad_()]
        make.gout = r(n, 64)
        both(tf.bond_tensor_product, make)
