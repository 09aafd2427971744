"""Fine-tuning step of the confidence-bootstrapping loop and its validation loops: `loss_function`, `train_epoch`, `AverageMeter`,
`test_epoch`, `inference_epoch_fix` (reference utils/training.py:17-126, 129-181, 184-233, 236-289, 292-373), with the reference's
signatures and return values.

MI355X design: the batch (a list of HeteroData, as the reference's DataListLoader yields on CUDA) is collated once, the forward
and backward passes run on the device (train_forward.py + the HIP tensor-product kernels), the loss is evaluated on the device
too (the reference moves every prediction to the CPU first), and -- one process per GPU -- gradients are averaged across ranks
with ONE flat RCCL all-reduce per step (`allreduce_gradients`, 16.3 MB for the shipped model) instead of DataParallel's
scatter/gather.
"""
from __future__ import annotations

import numpy as np
import torch

from .hostcfg import canonical_device, dev_key

from . import so3, torus
from .hostcfg import with_glue_threads


def _cat(data, name):
    if isinstance(data, (list, tuple)):
        return torch.cat([getattr(d, name) for d in data], dim=0)
    return getattr(data, name)


def loss_targets(data, t_to_sigma, device, no_torsion=False):
    """Everything of the loss that depends on the batch alone (reference utils/training.py:17-126): scores, sigmas and the score-norm
    table look-ups (host tables), uploaded to `device` -- on a GPU through the side stream, so that the host does not stall on the
    compute stream.  -> dict of device tensors; the hipGraph-captured step (train_graph.py) copies them into its static inputs."""
    dev = torch.device(device)
    if dev.type == "cuda":          # host -> device copies that do not stall the host on the compute stream (train_forward.upload)
        from .train_forward import upload
        up = lambda t: upload(t, dev)
    else:
        up = lambda t: t.to(dev)
    lst = isinstance(data, (list, tuple))
    ct = {k: (torch.cat([torch.as_tensor(d.complex_t[k], dtype=torch.float32).reshape(-1) for d in data]) if lst else data.complex_t[k])
          for k in ("tr", "rot", "tor")}
    tr_sigma, rot_sigma, tor_sigma = t_to_sigma(ct["tr"], ct["rot"], ct["tor"])
    tg = {"tr_score": up(_cat(data, "tr_score")), "tr_sigma": up(tr_sigma).unsqueeze(-1), "rot_score": up(_cat(data, "rot_score")),
          "rot_score_norm": up(so3.score_norm(rot_sigma.cpu()).unsqueeze(-1))}
    if not no_torsion:
        sig = [d.tor_sigma_edge for d in data] if lst else data.tor_sigma_edge
        edge_tor_sigma = np.concatenate(sig) if isinstance(sig, (list, tuple)) else np.asarray(sig)
        tg["tor_score"] = up(_cat(data, "tor_score"))
        tg["tor_score_norm2"] = up(torch.tensor(torus.score_norm(edge_tor_sigma)).float())
    return tg


def loss_from_targets(tr_pred, rot_pred, tor_pred, tg, tr_weight=1, rot_weight=1, tor_weight=1, apply_mean=True, no_torsion=False,
                      data=None):
    """The arithmetic of the loss on device tensors (`tg` from loss_targets) -> the reference's 11-tuple."""
    dev = tr_pred.device
    if apply_mean and dev.type == "cuda":
        from . import train_ops
        if train_ops.FUSED_LOSS and tr_pred.dtype == torch.float32 and (no_torsion or tor_pred is not None):
            # one launch for the 11 values and the gradients of the three predictions (csrc/train_loss.hip)
            o = train_ops.ScoreLossFn.apply(tr_pred, rot_pred, None if no_torsion else tor_pred, tg["tr_score"], tg["tr_sigma"].reshape(-1),
                                            tg["rot_score"], tg["rot_score_norm"].reshape(-1), None if no_torsion else tg["tor_score"],
                                            None if no_torsion else tg["tor_score_norm2"], (tr_weight, rot_weight, tor_weight), not no_torsion)
            d = o.detach()
            return (o[0:1],) + tuple(d[k:k + 1] for k in range(1, 11))
    mean_dims = (0, 1) if apply_mean else 1
    zeros = lambda: torch.zeros(1 if apply_mean else tr_pred.shape[0], dtype=torch.float, device=dev)
    tr_score, tr_sigma = tg["tr_score"], tg["tr_sigma"]
    tr_loss = ((tr_pred - tr_score) ** 2 * tr_sigma ** 2).mean(dim=mean_dims)
    tr_base_loss = (tr_score ** 2 * tr_sigma ** 2).mean(dim=mean_dims).detach()
    rot_score, rot_score_norm = tg["rot_score"], tg["rot_score_norm"]
    rot_loss = (((rot_pred - rot_score) / rot_score_norm) ** 2).mean(dim=mean_dims)
    rot_base_loss = ((rot_score / rot_score_norm) ** 2).mean(dim=mean_dims).detach()
    if not no_torsion:
        tor_score, tor_score_norm2 = tg["tor_score"], tg["tor_score_norm2"]
        tor_loss = (tor_pred - tor_score) ** 2 / tor_score_norm2
        tor_base_loss = (tor_score ** 2 / tor_score_norm2).detach()
        if apply_mean:
            tor_loss, tor_base_loss = tor_loss.mean() * torch.ones(1, device=dev), tor_base_loss.mean() * torch.ones(1, device=dev)
        else:
            lst = isinstance(data, (list, tuple))
            if lst:
                index = torch.cat([torch.full((int(d["ligand"].edge_mask.sum()),), i, dtype=torch.long) for i, d in enumerate(data)]).to(dev)
                n = len(data)
            else:
                index = data["ligand"].batch[data["ligand", "ligand"].edge_index[0][data["ligand"].edge_mask]].to(dev)
                n = data.num_graphs
            c = torch.zeros(n, device=dev).index_add_(0, index, torch.ones_like(tor_loss)) + 0.0001
            tor_loss = torch.zeros(n, device=dev).index_add(0, index, tor_loss) / c
            tor_base_loss = torch.zeros(n, device=dev).index_add(0, index, tor_base_loss) / c
    else:
        tor_loss, tor_base_loss = zeros(), zeros()
    backbone_loss, backbone_base_loss, sidechain_loss, sidechain_base_loss = zeros(), zeros(), zeros(), zeros()
    loss = tr_loss * tr_weight + rot_loss * rot_weight + tor_loss * tor_weight
    return (loss, tr_loss.detach(), rot_loss.detach(), tor_loss.detach(), backbone_loss, sidechain_loss,
            tr_base_loss, rot_base_loss, tor_base_loss, backbone_base_loss, sidechain_base_loss)


def loss_function(tr_pred, rot_pred, tor_pred, sidechain_pred, data, t_to_sigma, device, tr_weight=1, rot_weight=1, tor_weight=1,
                  backbone_weight=0, sidechain_weight=0, apply_mean=True, no_torsion=False):
    """Denoising score-matching loss (reference utils/training.py:17-126).  `data` is the list of noised graphs (or their
    collation); returns the reference's 11-tuple (loss, tr, rot, tor, backbone, sidechain, and the five base losses)."""
    if backbone_weight > 0 or sidechain_weight > 0:
        raise NotImplementedError("side-chain / backbone losses are outside the score-model fine-tuning path")
    tg = loss_targets(data, t_to_sigma, tr_pred.device, no_torsion=no_torsion)
    return loss_from_targets(tr_pred, rot_pred, tor_pred, tg, tr_weight, rot_weight, tor_weight, apply_mean, no_torsion, data=data)


class AverageMeter:
    """Running means of named values (reference utils/training.py:129-181).  Pooled scalars (train_epoch) are accumulated on the values'
    device -- no read-back per step, one in summary(); `unpooled_metrics` (per-complex vectors, test_epoch: the count advances by the
    number of complexes) and `intervals` > 1 (per-noise-level bins selected by `interval_idx`) follow the reference's host arithmetic."""

    def __init__(self, types, unpooled_metrics=False, intervals=1):
        self.types = types
        self.intervals = intervals
        self.unpooled_metrics = unpooled_metrics
        self.acc = None if intervals == 1 else torch.zeros(len(types), intervals)
        self.count = 0 if intervals == 1 else torch.zeros(len(types), intervals)

    def add(self, vals, interval_idx=None):
        if self.intervals > 1:
            for k, v in enumerate(vals):
                v = torch.as_tensor(v).detach().float().cpu().reshape(-1)
                idx = torch.as_tensor(interval_idx[k]).long().cpu().reshape(-1)
                self.count[k].index_add_(0, idx, torch.ones(len(v)))
                if not torch.allclose(v, torch.tensor(0.0)):
                    self.acc[k].index_add_(0, idx, v)
            return
        dev = next((v.device for v in vals if torch.is_tensor(v)), torch.device("cpu"))
        if self.unpooled_metrics:
            v0 = torch.as_tensor(vals[0])
            self.count += 1 if v0.dim() == 0 else len(v0)
            row = torch.stack([torch.as_tensor(v, device=dev).detach().float().sum() for v in vals])
        else:
            self.count += 1
            if all(torch.is_tensor(v) and v.numel() == 1 and v.dtype == torch.float32 and v.device == dev for v in vals):
                row = torch.cat([v.detach().reshape(1) for v in vals])          # the usual case (apply_mean=True): one launch
            else:
                row = torch.stack([torch.as_tensor(v, device=dev).detach().float().mean() for v in vals])
        self.acc = row if self.acc is None else self.acc + row

    def summary(self):
        if self.intervals > 1:
            return {f"int{i}_{t}": (self.acc[k][i] / self.count[k][i]).item() for i in range(self.intervals) for k, t in enumerate(self.types)}
        if self.acc is None:
            return {t: 0.0 for t in self.types}
        return {t: v / max(self.count, 1) for t, v in zip(self.types, self.acc.tolist())}


_METRICS = ["loss", "tr_loss", "rot_loss", "tor_loss", "backbone_loss", "sidechain_loss", "tr_base_loss", "rot_base_loss",
            "tor_base_loss", "backbone_base_loss", "sidechain_base_loss"]


def _dist_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


def allreduce_gradients(model, world_size=None, skip=False):
    """Average the gradients over ranks with one flat all-reduce (RCCL when the process group is 'nccl'; 4 084 564 floats =
    16.3 MB for the shipped model -- a single bucket, so the ring runs once per step at full message size).

    The buffer carries one extra element, this rank's `skip` flag: a rank whose batch is unusable (NaN loss, batch of one)
    still ENTERS the collective with zero gradients, and every rank learns whether anybody skipped.  Returns True when the
    step is good on all ranks.  (Deciding to skip on one rank alone would leave the others blocked in the all-reduce.)"""
    import torch.distributed as dist
    world_size = world_size or _dist_world()
    if world_size == 1:
        return not skip
    params = [p for p in model.parameters() if p.requires_grad]
    dev = params[0].device
    grads = [torch.zeros_like(p).reshape(-1) if (skip or p.grad is None) else p.grad.reshape(-1) for p in params]
    flat = torch.cat(grads + [torch.full((1,), 1.0 if skip else 0.0, device=dev, dtype=grads[0].dtype)])
    if flat.is_cuda and dist.get_backend() != "nccl":     # a host-memory backend (gloo in the tests): reduce a host copy
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    n_skipped = float(flat[-1])
    if n_skipped > 0 or not bool(torch.isfinite(flat).all()):      # non-finite gradients anywhere count as a skipped step everywhere
        return False
    flat /= world_size
    off = 0
    for p in params:
        n = p.numel()
        p.grad = flat[off:off + n].view_as(p).clone() if p.grad is None else p.grad.copy_(flat[off:off + n].view_as(p))
        off += n
    return True


def sync_batchnorm_buffers(model):
    """Average the BatchNorm running statistics over ranks (one flat all-reduce).  Each rank's train-mode forward updates its
    own running_mean / running_var from its own batches (DataParallel in the reference keeps a single copy on GPU 0); without
    this the inference engines of the ranks drift apart.  Called at the end of every epoch."""
    import torch.distributed as dist
    world = _dist_world()
    if world == 1:
        return
    bufs = [b for n, b in model.named_buffers() if n.endswith(("running_mean", "running_var")) and b.numel() > 0]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    if flat.is_cuda and dist.get_backend() != "nccl":
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        flat.copy_(host)
    else:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= world
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    if hasattr(model, "invalidate_engine"):
        model.invalidate_engine()


_NAN_FLAGS = {}


def _async_any_nan(t):
    """-> callable returning bool(isnan(t).any()); for a device tensor the flag travels through a pinned host buffer behind an event,
    so calling it later waits for that copy only (not for work enqueued after it)"""
    bad = torch.isnan(t).any()
    if not bad.is_cuda:
        return lambda: bool(bad)
    key = (dev_key(t.device), __import__("threading").get_ident())     # one pinned flag per device AND host thread: two models trained from
    buf = _NAN_FLAGS.get(key)                                       # two threads on one GPU do not share it
    if buf is None:
        buf = _NAN_FLAGS[key] = torch.zeros(1, dtype=torch.bool, pin_memory=True)
    buf.copy_(bad.reshape(1), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(t.device))

    def read():
        ev.synchronize()
        return bool(buf[0])
    return read


_PARAM_LISTS = __import__("weakref").WeakKeyDictionary()


def _parameter_list(model):
    """list(model.parameters()), built once per model: the generator walks the whole module tree (2 ms per call for the 215 tensors of
    the shipped model -- 5 % of a training step at batch 8).  The parameters of a model keep their identity through load_state_dict,
    .to() and optimizer steps; a model whose modules are REPLACED after its first step needs a fresh entry (`_PARAM_LISTS.pop`)."""
    lst = _PARAM_LISTS.get(model)
    if lst is None:
        lst = _PARAM_LISTS[model] = list(model.parameters())
    return lst


@with_glue_threads
def train_step(model, data, optimizer, device, t_to_sigma, loss_fn, ema_weights=None, forward_fn=None, skip=False, prepared=None,
               before_backward=None):
    """One optimisation step on a list of noised graphs (body of the reference loop, utils/training.py:195-211).
    Returns None when the step was skipped -- on ANY rank: the skip decision (NaN loss, `skip=True` for an unusable batch) is
    made collectively inside the gradient all-reduce, so no rank is left waiting in a collective the others never enter.
    `prepared`: the batch after train_forward.prepare_batch (done ahead of the step); `before_backward`: called once, right before the
    backward pass is enqueued -- the moment at which the host has idle time to fetch and prepare the NEXT batch (train_epoch)."""
    if forward_fn is None:
        from .train_forward import forward as forward_fn
    optimizer.zero_grad()
    loss_tuple = None
    if not skip:
        tr_pred, rot_pred, tor_pred, sc = forward_fn(model, data if prepared is None else prepared)
        loss_tuple = loss_fn(tr_pred, rot_pred, tor_pred, sc, data=data, t_to_sigma=t_to_sigma, device=device)
        loss = loss_tuple[0]
        # NaN check without a pipeline flush.  The reference tests the loss between forward and backward (utils/training.py:201); on an
        # asynchronous device that read-back drains the GPU and the host then enqueues the whole backward pass with the GPU idle.  Here
        # the flag is COPIED to pinned host memory asynchronously, the backward pass is enqueued, and only then is the copy waited for:
        # by that time the forward pass has long finished, so the wait is over the copy alone, not over the backward pass.  A NaN
        # loss gives NaN gradients, which are discarded below exactly as if backward had not run.
        nan_flag = _async_any_nan(loss.detach())
        if before_backward is not None:
            before_backward()
            before_backward = None
        # INVARIANT the custom backward functions keep: nothing they write outlives the step unless the step is applied -- hub.grads is
        # re-zeroed by pack(), train_ops._DW_SCRATCH is overwritten on its next use, parameter .grad is dropped by zero_grad() below --
        # so a backward pass over a NaN loss leaves no trace (the reference skips BEFORE backward, utils/training.py:201)
        loss.backward()
        if nan_flag():
            skip = True
    if before_backward is not None:         # skipped before the backward pass: the caller's look-ahead still has to start
        before_backward()
    if not allreduce_gradients(model, skip=skip):
        optimizer.zero_grad()
        return None
    optimizer.step()
    if ema_weights is not None:
        ema_weights.update(_parameter_list(model))
    return (loss.detach(),) + tuple(loss_tuple[1:])


def uniform_step_count(loader):
    """Number of batches every rank runs this epoch: the minimum of the ranks' loader lengths (each rank's buffer / loader has its
    own length; the surplus batches of longer loaders are dropped, like drop_last over ranks)."""
    import torch.distributed as dist
    try:
        n = len(loader)
    except TypeError:                       # an iterable without __len__: materialise it once to count its batches
        raise TypeError("train_epoch under torch.distributed needs a loader with __len__ (the ranks agree on a step count before "
                        "the epoch starts); wrap the iterable in a list") from None
    if _dist_world() == 1:
        return n
    t = torch.tensor([n], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


_GRAPHED = __import__("weakref").WeakKeyDictionary()       # model -> GraphedStep, which holds its model WEAKLY: the graphs go when the model goes


def release_graphs(model=None):
    """Drops the captured graphs (and their private memory pools) of `model`, or of every model."""
    if model is None:
        _GRAPHED.clear()
    else:
        _GRAPHED.pop(model, None)


def graphed_step_for(model, optimizer, device, t_to_sigma, loss_fn, ema_weights, **kw):
    """The model's train_graph.GraphedStep (hipGraph-captured forward + loss + backward), built once and re-used across epochs as long
    as optimiser, EMA and loss weights are the same objects / values.  `loss_fn` must be `loss_function` or a functools.partial of it."""
    import functools
    from .train_graph import GraphedStep
    if isinstance(loss_fn, functools.partial) and loss_fn.func is loss_function and not loss_fn.args:
        lw = dict(loss_fn.keywords)
    elif loss_fn is loss_function:
        lw = {}
    else:
        raise TypeError("the hipGraph-captured step evaluates training.loss_function on the device; pass it (or a partial of it)")
    if not lw.get("apply_mean", True):
        raise NotImplementedError("the captured step computes the batch-mean loss (apply_mean=True)")
    lw.pop("apply_mean", None)
    cur = _GRAPHED.get(model)
    sig = (id(optimizer), id(ema_weights), dev_key(device), tuple(sorted(lw.items())))
    if cur is None or cur[0] != sig:
        cur = _GRAPHED[model] = (sig, GraphedStep(model, optimizer, device, t_to_sigma, lw, ema_weights, **kw))
    return cur[1]


def _train_epoch_graphed(model, loader, optimizer, device, t_to_sigma, loss_fn, ema_weights, meter, n_steps, distributed):
    """train_epoch on the hipGraph-captured step: while the graph of step k runs on the GPU the host fetches, collates and prepares
    batch k + 1 (side stream); then it reads step k's NaN flag, steps the optimiser and launches k + 1."""
    trainer = graphed_step_for(model, optimizer, device, t_to_sigma, loss_fn, ema_weights)

    def close(item):
        out = trainer.finish(item)
        if out is None:
            print("Nan loss, skipping batch" + (" (on some rank)" if distributed else ""))
        else:
            meter.add(out)
    pending, i = None, 0
    import os
    import time
    prof = trainer.stats.setdefault("host_s", {"prepare": 0.0, "finish": 0.0, "launch": 0.0, "steps": 0}) if os.environ.get("CBD_TRAIN_PROF") else None
    for data in loader:
        if n_steps is not None and i >= n_steps:
            break
        i += 1
        n = len(data) if isinstance(data, (list, tuple)) else data.num_graphs
        if n == 1:
            print("Skipping batch of size 1 since otherwise batchnorm would not work.")
            if pending is not None:
                close(pending)
                pending = None
            if distributed:     # the other ranks are inside this step's all-reduce: take part with zero gradients
                allreduce_gradients(model, skip=True)
                optimizer.zero_grad()
            continue
        t0 = time.perf_counter()
        item = trainer.prepare(data if isinstance(data, (list, tuple)) else data.to_data_list())
        t1 = time.perf_counter()
        if pending is not None:
            close(pending)
        t2 = time.perf_counter()
        pending = trainer.launch(item)
        if prof is not None:
            t3 = time.perf_counter()
            prof["prepare"] += t1 - t0
            prof["finish"] += t2 - t1
            prof["launch"] += t3 - t2
            prof["steps"] += 1
            prof.setdefault("trace", []).append((round((t1 - t0) * 1e3, 1), round((t2 - t1) * 1e3, 1), round((t3 - t2) * 1e3, 1)))
    if pending is not None:
        close(pending)


@with_glue_threads
def train_epoch(model, loader, optimizer, device, t_to_sigma, loss_fn, ema_weights, torsional=False, forward_fn=None, look_ahead=False,
                hip_graph=False):
    """One epoch (reference utils/training.py:184-233).  `hip_graph=True`: forward + loss + backward of every step as ONE hipGraph launch
    (train_graph.py: capacity-padded batches, a graph per batch shape, captured at the shape's second sighting).  With `look_ahead=True` (HIP forward on a GPU only) the NEXT batch is fetched
    from the loader -- which is where the reference's DataLoader workers run NoiseTransform -- collated and taken through the
    input-only part of the forward pass (train_forward.prepare_batch: radius graphs, edge groupings, ...) on a second host thread
    while the current step's backward pass is being enqueued; the loader is only ever advanced by one thread at a time, in order, and
    the results are identical to the sequential loop (tests/test_gpu_train_step.py::test_train_epoch_with_look_ahead_...).  Off by default: with batches that are already
    noised (bench.py) the two threads compete for the interpreter lock and the step time does not change (36.0 / 39.8 vs 38.3 / 36.0 ms
    at batch 8, round 3); it pays when the loader itself does host work per batch."""
    if torsional:
        raise NotImplementedError("torsional-only training is outside the score-model fine-tuning path")
    model.train()
    meter = AverageMeter(_METRICS)
    distributed = _dist_world() > 1
    n_steps = uniform_step_count(loader) if distributed else None
    dev = device = canonical_device(device)      # one spelling for every per-device cache key downstream ('cuda' == 'cuda:0')
    if hip_graph:
        if forward_fn is not None or dev.type != "cuda":
            raise RuntimeError("hip_graph=True runs the package's own HIP forward on a GPU")
        _train_epoch_graphed(model, loader, optimizer, device, t_to_sigma, loss_fn, ema_weights, meter, n_steps, distributed)
        sync_batchnorm_buffers(model)
        return meter.summary()
    ahead = look_ahead and forward_fn is None and dev.type == "cuda"
    it = iter(loader)

    def fetch():
        data = next(it, None)
        if data is None:
            return None
        n = len(data) if isinstance(data, (list, tuple)) else data.num_graphs
        prepared = None
        if ahead and n > 1:
            from .train_forward import prepare_batch
            prepared = prepare_batch(model, data, dev)
        return data, prepared, n

    pool = None
    if ahead:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="cbd-look-ahead")
    try:
        pending = pool.submit(fetch) if pool else None
        i = 0
        while True:
            item = pending.result() if pool else fetch()
            pending = None
            if item is None or (n_steps is not None and i >= n_steps):
                break
            data, prepared, n = item
            i += 1
            nxt = {}

            def start_next():
                if pool and "f" not in nxt:
                    nxt["f"] = pool.submit(fetch)

            if n == 1:
                print("Skipping batch of size 1 since otherwise batchnorm would not work.")
                if distributed:     # the other ranks are inside this step's all-reduce: take part with zero gradients
                    train_step(model, data, optimizer, device, t_to_sigma, loss_fn, ema_weights, forward_fn=forward_fn, skip=True)
                start_next()
                pending = nxt.get("f")
                continue
            out = train_step(model, data, optimizer, device, t_to_sigma, loss_fn, ema_weights, forward_fn=forward_fn, prepared=prepared,
                             before_backward=start_next if pool else None)
            start_next()
            pending = nxt.get("f")
            if out is None:
                print("Nan loss, skipping batch" + (" (on some rank)" if distributed else ""))
                continue
            meter.add(out)
    finally:
        if pool:
            pool.shutdown(wait=True)
    sync_batchnorm_buffers(model)
    return meter.summary()


@with_glue_threads
def test_epoch(model, loader, device, t_to_sigma, loss_fn, test_sigma_intervals=False, torsional=False):
    """Validation pass (reference utils/training.py:236-289): eval mode, no gradients, per-complex losses (`apply_mean=False`) pooled over
    the loader, and with `test_sigma_intervals` the same losses binned into ten noise levels of every component (`int{i}_{name}` keys).
    The batches carry a diffusion time per complex, so the forward is the batched HIP path of train_forward.py in eval mode (the fused
    sampling engine advances one time for a whole batch); tests/test_gpu_train_step.py checks it against the engine complex by complex."""
    if torsional:
        raise NotImplementedError("the torsional-only model is outside the score-model fine-tuning path")
    model.eval()
    meter = AverageMeter(_METRICS, unpooled_metrics=True)
    meter_all = AverageMeter(_METRICS, unpooled_metrics=True, intervals=10) if test_sigma_intervals else None
    net = getattr(model, "module", model)
    for data in loader:
        with torch.no_grad():
            tr_pred, rot_pred, tor_pred, sidechain_pred = net.forward_train(data)
        loss_tuple = loss_fn(tr_pred, rot_pred, tor_pred, sidechain_pred, data=data, t_to_sigma=t_to_sigma, apply_mean=False, device=device)
        if loss_tuple is None:
            continue
        vals = [v.detach().cpu() for v in loss_tuple]
        meter.add(vals)
        if meter_all is not None:
            lst = isinstance(data, (list, tuple))
            ct = {k: (torch.cat([torch.as_tensor(d.complex_t[k], dtype=torch.float32).reshape(-1) for d in data]) if lst
                      else data.complex_t[k]).cpu() for k in ("tr", "rot", "tor")}
            i_tr, i_rot, i_tor = (torch.round(ct[k] * (10 - 1)).long() for k in ("tr", "rot", "tor"))
            meter_all.add(vals, [i_tr, i_tr, i_rot, i_tor, i_tr, i_tr, i_tr, i_rot, i_tor, i_tr, i_tr])
    out = meter.summary()
    if meter_all is not None:
        out.update(meter_all.summary())
    return out


@with_glue_threads
def inference_epoch_fix(model, complex_graphs, device, t_to_sigma, args):
    """Validation by docking (reference utils/training.py:292-373): `args.inference_samples` poses per complex through sampling() without a
    confidence model, symmetry-corrected RMSD to the crystal pose(s); a complex whose sampling failed six times counts as 100 A.
    -> {'rmsds_lt2', 'rmsds_lt5', 'min_rmsds_lt2', 'min_rmsds_lt5'} in percent."""
    import copy
    from .diffusion_utils import get_inverse_schedule, get_t_schedule
    from .hetero import Batch
    from .molecules_utils import get_symmetry_rmsd, remove_all_hs
    from .sampling import randomize_position, sampling
    t_schedule = get_t_schedule(sigma_schedule="expbeta", inference_steps=args.inference_steps, inf_sched_alpha=1, inf_sched_beta=1)
    asyn = bool(getattr(args, "asyncronous_noise_schedule", False))
    if asyn:
        tr_schedule = get_inverse_schedule(t_schedule, args.sampling_alpha, args.sampling_beta)
        rot_schedule = get_inverse_schedule(t_schedule, args.rot_alpha, args.rot_beta)
        tor_schedule = get_inverse_schedule(t_schedule, args.tor_alpha, args.tor_beta)
    else:
        tr_schedule = rot_schedule = tor_schedule = t_schedule
    net = getattr(model, "module", model)
    net.eval()
    rmsds, min_rmsds = [], []
    for orig in complex_graphs:
        orig = orig if isinstance(orig, Batch) else Batch.from_data_list([orig])
        data_list = [orig.shallow_copy() if hasattr(orig, "shallow_copy") else copy.deepcopy(orig) for _ in range(args.inference_samples)]
        randomize_position(data_list, args.no_torsion, False, args.tr_sigma_max, pocket_knowledge=getattr(args, "inf_pocket_knowledge", False),
                           pocket_cutoff=getattr(args, "inf_pocket_cutoff", 7))
        predictions_list, failed = None, 0
        while predictions_list is None and failed <= 5:
            try:
                predictions_list, _ = sampling(data_list=data_list, model=net, inference_steps=args.inference_steps, tr_schedule=tr_schedule,
                                               rot_schedule=rot_schedule, tor_schedule=tor_schedule, device=device, t_to_sigma=t_to_sigma,
                                               model_args=args, asyncronous_noise_schedule=asyn, t_schedule=t_schedule)
            except Exception as e:
                failed += 1
                print("failed 5 times - skipping the complex" if failed > 5 else f"Exception while running inference on complex: {e}")
        if predictions_list is None:
            rmsds.extend([100] * args.inference_samples)
            min_rmsds.append(100)
            continue
        center = orig.original_center.cpu().numpy()
        orig_pos = orig["ligand"].pos.cpu().numpy() + center if args.no_torsion else orig["ligand"].orig_pos
        if isinstance(orig_pos, list):
            orig_pos = orig_pos[0]
        orig_pos = np.asarray(orig_pos, dtype=np.float32)
        orig_pos = orig_pos[None] if orig_pos.ndim == 2 else orig_pos
        filterHs = torch.not_equal(predictions_list[0]["ligand"].x[:, 0], 0).cpu().numpy()
        ligand_pos = np.asarray([g["ligand"].pos.cpu().numpy()[filterHs] for g in predictions_list])
        ref = orig_pos[:, filterHs] - center
        mol = getattr(orig, "mol", None)
        mol = mol[0] if isinstance(mol, (list, tuple)) else mol
        mol = remove_all_hs(mol)          # RemoveAllHs(orig_complex_graph.mol[0]) in the reference; the coordinates are filtered with filterHs
        per_ref = []
        for r in ref:
            try:
                per_ref.append(np.asarray(get_symmetry_rmsd(mol, r, [l for l in ligand_pos], device=device)))
            except Exception as e:
                print("Using non corrected RMSD because of the error:", e)
                per_ref.append(np.sqrt(((ligand_pos - r) ** 2).sum(axis=2).mean(axis=1)))
        rmsd = np.min(np.asarray(per_ref), axis=0)
        rmsds.extend(rmsd.tolist())
        min_rmsds.append(rmsd.min())
    rmsds, min_rmsds = np.asarray(rmsds, dtype=np.float64), np.asarray(min_rmsds, dtype=np.float64)
    return {"rmsds_lt2": 100 * (rmsds < 2).sum() / len(rmsds), "rmsds_lt5": 100 * (rmsds < 5).sum() / len(rmsds),
            "min_rmsds_lt2": 100 * (min_rmsds < 2).sum() / len(min_rmsds), "min_rmsds_lt5": 100 * (min_rmsds < 5).sum() / len(min_rmsds)}
