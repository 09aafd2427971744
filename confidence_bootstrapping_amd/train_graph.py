"""The fine-tuning step as a hipGraph (BASELINE.json configs[4]; reference utils/training.py:195-211 is the step, torch + DataParallel).

At the reference's batch sizes the eager step is HOST-bound on the MI355X: ~970 launches and ~10 800 ATen / autograd calls per step cost
~32 ms of host time against ~23.5 ms of GPU kernels.  Here the whole forward + loss + backward of a step is ONE
`hipGraphLaunch`, the way the sampler's step loop is:

  * capacity padding (`train_forward.prepare_batch(pad=...)`): a tiny filler graph is appended to the batch and the three radius graphs
    of the step (ligand-ligand, ligand-receptor, torsion) -- the only tensors whose size depends on the noise -- are padded with edges
    INSIDE the filler up to the next multiple of a bucket.  The filler is a connected component of its own: no message reaches a real
    node, its rows are excluded from every BatchNorm statistic (`cbd_irreps_bn_*` exclusion ranges) and from the loss, and autograd
    sends it exact zeros -- the parameters' gradients are those of the unpadded step up to the association of sums;
  * a step's inputs (`PreparedBatch` + loss targets: ~100 device tensors in ~70 storages) are produced eagerly on the side stream as
    before -- the radius searches read their edge counts back there -- and COPIED into a static twin with the same storage layout; the
    graph was captured on that twin.  Key of a graph = every shape / stride / offset and every host scalar of the structure; a new key is
    run eagerly once (warm-up) and captured at its second sighting; at most `max_graphs` graphs are kept (least recently used);
  * Adam / EMA (and the gradient all-reduce of a multi-rank run) stay outside the graph: the NaN check of the reference (skip the
    step, utils/training.py:201) is a host decision between backward and the optimiser.

`GraphedStep.prepare()` is pure host + side-stream work and may run while the previous step's graph executes; `launch()` enqueues the
copy-in and the graph; `finish()` reads the NaN flag (this is where the host waits for the GPU), reduces the gradients over ranks and
steps the optimiser.  training.train_epoch(hip_graph=True) pipelines the three.
"""
from __future__ import annotations

import collections
from typing import List

import numpy as np
import torch

from .hostcfg import canonical_device, dev_key

from . import train_forward as tf
from . import train_ops as to
from .hetero import Batch, HeteroData, Store
from .training import allreduce_gradients, loss_from_targets, loss_targets, _parameter_list


# ---- walking a prepared step --------------------------------------------------------------------------------------------------------
def _walk(obj, fn, memo):
    """structure-preserving copy of `obj` with every CUDA tensor replaced by fn(tensor); shared sub-objects stay shared"""
    if torch.is_tensor(obj):
        return fn(obj) if obj.is_cuda else obj
    if obj is None or isinstance(obj, (int, float, str, bool, np.ndarray, np.generic, torch.cuda.Event)):
        return obj
    oid = id(obj)
    if oid in memo:
        return memo[oid]
    if isinstance(obj, dict):
        out = memo[oid] = {}
        for k, v in obj.items():
            out[k] = _walk(v, fn, memo)
        return out
    if isinstance(obj, (list, tuple)):
        out = [_walk(v, fn, memo) for v in obj]
        out = memo[oid] = (tuple(out) if isinstance(obj, tuple) else out)
        return out
    if isinstance(obj, to.Csr):
        out = memo[oid] = to.Csr.from_parts(_walk(obj.index, fn, memo), obj.n_rows, _walk(obj.perm, fn, memo), _walk(obj.rowptr, fn, memo))
        return out
    if isinstance(obj, Store):
        out = memo[oid] = Store()
        for k, v in obj.__dict__.items():
            out.__dict__[k] = _walk(v, fn, memo)
        return out
    if isinstance(obj, HeteroData):
        out = memo[oid] = type(obj)()
        for k, st in obj._stores.items():
            out._stores[k] = _walk(st, fn, memo)
        for k, v in obj.__dict__.items():
            if not k.startswith("_"):
                out.__dict__[k] = _walk(v, fn, memo)
        if hasattr(obj, "_num_graphs"):
            object.__setattr__(out, "_num_graphs", obj._num_graphs)
        return out
    if isinstance(obj, tf._Prepared):
        out = memo[oid] = tf._Prepared()
        for k, v in obj.__dict__.items():
            out.__dict__[k] = _walk(v, fn, memo)
        return out
    return obj


def _structure(prep, targets):
    """the parts of a prepared step the graph reads (the keep-alive list and the event are not among them)"""
    return {"batch": prep.batch, "g": prep.g, "csr": [prep.csr[k] for k in prep.csr], "targets": targets,
            "pad": {k: v for k, v in (prep.pad or {}).items() if k != "edges_real"}}


def _signature(struct):
    """hashable description of every tensor (dtype, shape, stride, offset, storage size) and host scalar of the structure"""
    sig = []

    def rec(t):
        st = t.untyped_storage()
        sig.append((str(t.dtype), tuple(t.shape), tuple(t.stride()), int(t.storage_offset()), int(st.nbytes())))
        return t

    def scalars(obj, depth=0):
        if isinstance(obj, (int, float, str, bool)):
            sig.append(obj)
        elif isinstance(obj, (list, tuple)) and depth < 3 and all(isinstance(v, (int, float, str, bool, tuple, list)) for v in obj):
            sig.append(("seq", len(obj)))
            for v in obj:
                scalars(v, depth + 1)
    _walk(struct, rec, {})
    pad = struct.get("pad") or {}
    for k in ("B_real", "T_real", "ex_lig", "ex_rec", "ex_joint", "ex_graph", "ex_bond"):
        sig.append((k, pad.get(k)))
    g = struct["g"]
    for k in sorted(g.__dict__):
        v = g.__dict__[k]
        if not torch.is_tensor(v) and not isinstance(v, to.Csr):
            sig.append(k)
            scalars(v)
    return tuple(sig)


class _Twin:
    """static copy of one prepared step: the same structure on storages of its own, filled per step by copy_in()"""

    def __init__(self, prep, targets):
        self.clones = collections.OrderedDict()          # source storage address -> static flat uint8 tensor

        def static_of(t):
            st = t.untyped_storage()
            flat = self.clones.get(st.data_ptr())
            if flat is None:
                flat = self.clones[st.data_ptr()] = torch.empty(max(int(st.nbytes()), 1), dtype=torch.uint8, device=t.device)
            return torch.empty(0, dtype=t.dtype, device=t.device).set_(flat.untyped_storage(), t.storage_offset(), t.size(), t.stride())
        s = _walk(_structure(prep, targets), static_of, {})
        self.sizes = [int(f.numel()) for f in self.clones.values()]
        self.statics = list(self.clones.values())
        self.clones = None
        self.copy_in(prep, targets)
        csr = {}
        for c in s["csr"]:
            csr[to._csr_key(c.index, c.n_rows)] = c
        self.targets = s["targets"]
        self.prep = tf.PreparedBatch(s["batch"], s["g"], csr, None, None, prep.num_graphs, pad=prep.pad)

    @staticmethod
    def _flat_sources(prep, targets):
        seen, out = set(), []

        def rec(t):
            st = t.untyped_storage()
            if st.data_ptr() not in seen:
                seen.add(st.data_ptr())
                n = int(st.nbytes())
                out.append(torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (max(n, 0),), (1,)) if n else None)
            return t
        _walk(_structure(prep, targets), rec, {})
        return out

    def copy_in(self, prep, targets):
        src = self._flat_sources(prep, targets)
        if len(src) != len(self.statics):
            raise RuntimeError("prepared step does not match the captured structure")
        live_dst, live_src = [], []
        for d, s_ in zip(self.statics, src):
            if s_ is None:
                continue
            if s_.numel() != d.numel():
                raise RuntimeError("prepared step does not match the captured storage sizes")
            live_dst.append(d)
            live_src.append(s_)
        torch._foreach_copy_(live_dst, live_src)


class _Captured:
    def __init__(self, graph, twin, grads, loss_tuple):
        self.graph, self.twin, self.grads, self.loss_tuple = graph, twin, grads, loss_tuple


class GraphedStep:
    """forward + loss + backward of the fine-tuning step as one hipGraph launch per step (module docstring)."""

    @property
    def model(self):
        m = self._model_ref()
        if m is None:
            raise RuntimeError("the model of this GraphedStep has been garbage-collected")
        return m

    def __init__(self, model, optimizer, device, t_to_sigma, loss_kwargs=None, ema_weights=None, pad=True, max_graphs=12, capture_after=1):
        # the model is held weakly: training._GRAPHED maps model -> this object, and a value that kept its key alive would never be collected
        self._model_ref = __import__("weakref").ref(model)
        self.opt, self.dev, self.t2s = optimizer, canonical_device(device), t_to_sigma
        self.lw = dict(loss_kwargs or {})
        for k in ("backbone_weight", "sidechain_weight"):
            if self.lw.pop(k, 0):
                raise NotImplementedError("side-chain / backbone losses are outside the score-model fine-tuning path")
        self.no_torsion = bool(self.lw.get("no_torsion", False))
        self.ema, self.pad = ema_weights, pad
        self.graphs = collections.OrderedDict()
        self.seen = collections.Counter()
        self.max_graphs, self.capture_after = int(max_graphs), int(capture_after)
        self.params = _parameter_list(model)
        self.stats = {"replays": 0, "eager": 0, "captures": 0}

    # ---- stage 1: host + side stream
    def prepare(self, data: List[HeteroData]):
        was = tf._SIDE_PRIORITY[0]
        tf.side_priority(False)         # normal priority next to a running graph (train_forward.side_priority)
        try:
            prep = tf.prepare_batch(self.model, data, self.dev, pad=self.pad)
            targets = loss_targets(data, self.t2s, self.dev, no_torsion=self.no_torsion)
        finally:
            tf._SIDE_PRIORITY[0] = was
        return {"data": data, "prep": prep, "targets": targets, "key": _signature(_structure(prep, targets))}

    # ---- stage 2: copy-in + graph launch (or the eager step for a shape not captured yet)
    def _eager(self, item):
        self.model.zero_grad(set_to_none=True)
        tr, rot, tor, _ = tf.forward(self.model, item["prep"])
        lt = loss_from_targets(tr, rot, tor, item["targets"], **self.lw)
        lt[0].backward()
        self.stats["eager"] += 1
        return tuple(t.detach() for t in lt)       # no reference to the autograd graph survives the step (see _capture)

    def _capture(self, item):
        dev = self.dev
        main = torch.cuda.current_stream(dev)
        main.wait_event(item["prep"].event)
        twin = _Twin(item["prep"], item["targets"])
        tf._keep_until_main_passes(item["prep"].keep, dev)
        self.model.zero_grad(set_to_none=True)
        # An autograd graph of an EARLIER step that is still alive keeps its AccumulateGrad nodes alive, and those are bound to the
        # stream they were created on (the default stream): re-used under capture they synchronise across streams and break it.  The
        # only long-lived holder is the stream hub (its packed streams are autograd-connected to the parameters): drop them.
        hub = tf._HUBS.get(self.model)
        if hub is not None:
            hub.big = hub.w2p_all = hub.big_t = hub.grads = None
        graph = torch.cuda.CUDAGraph()
        eager_scratch = to._DW_SCRATCH.pop(dev_key(dev), None)       # the graph gets a scratch buffer of its own, from its private pool
        try:
            with torch.cuda.graph(graph):
                tr, rot, tor, _ = tf.forward(self.model, twin.prep)
                lt = loss_from_targets(tr, rot, tor, twin.targets, **self.lw)
                lt[0].backward()
        finally:
            to._DW_SCRATCH.pop(dev_key(dev), None)
            if eager_scratch is not None:
                to._DW_SCRATCH[dev_key(dev)] = eager_scratch
        grads = [p.grad for p in self.params]
        cap = _Captured(graph, twin, grads, tuple(t.detach() for t in lt))
        self.stats["captures"] += 1
        return cap

    def launch(self, item):
        dev = self.dev
        tf._rotate_keep(dev)
        key = item["key"]
        cap = self.graphs.get(key)
        if cap is None and self.seen[key] >= self.capture_after:
            cap = self._capture(item)
            self.graphs[key] = cap
            while len(self.graphs) > self.max_graphs:
                self.graphs.popitem(last=False)
        elif cap is not None:
            torch.cuda.current_stream(dev).wait_event(item["prep"].event)
            cap.twin.copy_in(item["prep"], item["targets"])
            tf._keep_until_main_passes(item["prep"].keep, dev)
        self.seen[key] += 1
        if cap is None:
            lt = self._eager(item)
        else:
            self.graphs.move_to_end(key)
            cap.graph.replay()
            for p, g in zip(self.params, cap.grads):
                p.grad = g
            lt = cap.loss_tuple
            self.stats["replays"] += 1
        from .training import _async_any_nan
        item["loss_tuple"] = lt
        item["nan"] = _async_any_nan(lt[0].detach())
        torch.cuda.current_stream(dev).query()      # push the enqueued work to the GPU now: the host goes on to prepare the next batch
        return item

    # ---- stage 3: the host decision and the optimiser
    def finish(self, item, skip=False):
        bad = bool(item["nan"]()) or skip
        if not allreduce_gradients(self.model, skip=bad):
            self.model.zero_grad(set_to_none=True)
            return None
        self.opt.step()
        if self.ema is not None:
            self.ema.update(self.params)
        lt = item["loss_tuple"]
        # a replayed graph overwrites its outputs at the next launch: hand out copies
        return tuple(t.detach().clone() for t in lt)

    def step(self, data):
        return self.finish(self.launch(self.prepare(data)))
