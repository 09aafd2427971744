"""Differentiable forward pass of `TensorProductScoreModel` for the fine-tuning step (reference
models/score_model.py:282-449 under `model.train()`, called from utils/training.py:198 `model(data)`).

Heterogeneous batches (different complexes, one diffusion time each), training-mode e3nn BatchNorm (batch statistics, running
averages updated), Dropout where the reference has it.  Every FasterTensorProduct layer -- > 99 % of the FLOPs -- runs on the
hand-written gfx950 kernels of csrc/tp_train.hip through `train_ops.TensorProductHubFn`; the surrounding small ops (embeddings,
radius graphs, scatter-mean, BatchNorm, the two e3nn heads) are autograd-visible torch ops on the same device.  The inference
path (`model.eval()` + sampling) does not come through here: it runs on the fused engine (engine.py).

Works in eval mode too (running statistics, no dropout), which the tests use to cross-check this path against the engine.
"""
from __future__ import annotations

import contextlib
import math
import os
import weakref
from typing import List

import numpy as np
import torch

from .hostcfg import canonical_device, dev_key
import torch.nn.functional as F

from . import so3, torus, train_ops
from .hetero import Batch, HeteroData
from .score_model import parse_irreps
from .train_ops import (LEVEL_DIMS, NODE_STRIDE, Csr, IrrepsBatchNormFn, RadiusQuery, ScatterSumFn, csr_build_many, edge_geometry, radius_queries, use_csr_cache, StreamHub, TensorProductHubFn, csr_of, edge_cat, gather_pad,
                        fc_first_stage, linear as _linear, mlp as _mlp,
                        gather_rows, scatter_mean as _scatter_mean_op,
                        scatter_sum, stream_map)

SQ3 = math.sqrt(3.0)


# ----------------------------------------------------------------------------- batching
def collate(data_list: List[HeteroData], device, keep=None) -> Batch:
    """Collation of the fields the score model reads (what PyG's Batch gives the reference): node tensors concatenated,
    edge_index offset per graph, `batch` vectors, per-graph diffusion times (utils/diffusion_utils.py:150-179) as tensors.
    Per-graph host-to-device copies on a side stream, concatenated on the device; the input graphs are not copied or modified."""
    if isinstance(data_list, Batch):
        return data_list.to(device)
    b = Batch()
    object.__setattr__(b, "_num_graphs", len(data_list))
    nl = [d["ligand"].num_nodes for d in data_list]
    nr = [d["receptor"].num_nodes for d in data_list]
    offs = lambda sizes: np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
    lo, ro = offs(nl), offs(nr)
    # host-to-device copies go through their own HIP stream: on the compute stream a copy from pageable memory would block the host
    # until the previous step's kernels have drained (26 ms per step in the first profile)
    side = _copy_stream(device)
    keep_alive = []

    names = [getattr(d, "name", None) for d in data_list]
    names = [n if isinstance(n, str) else None for n in names]

    def dv(parts, dim=0, static=False, role=None):
        """per-graph host-to-device copies, concatenated on the device (a host-side cat of the 2 MB/complex language-model
        features costs more than the whole GPU step on a many-core host).  static=True: tensors that belong to the COMPLEX, not to the
        noised sample (receptor features / trace / graph, ligand atom types / bonds) -- the buffer hands out shallow copies that share
        them, so their device copies are cached across steps (_dev_cached) instead of being uploaded again for every sample."""
        with torch.cuda.stream(side):
            up = [(_dev_cached(p, device, (n, role) if n is not None and role else None) if static else p.to(device, non_blocking=True))
                  for p, n in zip(parts, names)]
            out = torch.cat(up, dim) if len(up) > 1 else (up[0].clone() if static else up[0])
        keep_alive.append(out)
        return out

    # Everything that changes from step to step (noised ligand positions, diffusion times, and the host-built index vectors) travels in
    # TWO pinned staging buffers -- one float32, one int64 -- i.e. two truly asynchronous copies per step instead of ~25 blocking
    # copies from pageable memory (4.6 ms of host time per step in the profile of round 3); the device tensors are views of the two.
    ne_l = [d["ligand", "ligand"].edge_index.shape[1] for d in data_list]
    ne_r = [d["receptor", "receptor"].edge_index.shape[1] for d in data_list]
    t_host = {k: torch.cat([torch.as_tensor(d.complex_t[k], dtype=torch.float32).reshape(-1) for d in data_list]) for k in ("tr", "rot", "tor")}
    eo = np.concatenate([[0], np.cumsum(ne_l)[:-1]])
    rot = np.concatenate([np.flatnonzero(d["ligand"].edge_mask.numpy()) + o for d, o in zip(data_list, eo)]).astype(np.int64)
    fparts = [d["ligand"].pos.reshape(-1).float() for d in data_list] + [t_host["tr"], t_host["rot"], t_host["tor"]]
    iparts = [torch.from_numpy(a) for a in (np.repeat(np.arange(len(nl)), nl), np.repeat(np.arange(len(nr)), nr), rot,
                                            np.repeat(np.asarray(lo, dtype=np.int64), ne_l), np.repeat(np.asarray(ro, dtype=np.int64), ne_r),
                                            np.concatenate([[0], np.cumsum(nl)]).astype(np.int64), np.concatenate([[0], np.cumsum(nr)]).astype(np.int64))]

    def stage(parts, dtype):
        n = sum(int(p.numel()) for p in parts)
        host = torch.empty(n, dtype=dtype, pin_memory=True)
        torch.cat([p.to(dtype) for p in parts], out=host)
        with torch.cuda.stream(side):
            dev_t = host.to(device, non_blocking=True)
        keep_alive.append(dev_t)
        out, o = [], 0
        for p in parts:
            out.append(dev_t[o:o + p.numel()])
            o += p.numel()
        return out, dev_t

    (fdev, fall), (idev, _) = stage(fparts, torch.float32), stage(iparts, torch.int64)
    B = len(data_list)

    def edges(parts, per_edge_offset, role):
        """edge_index tensors: cached raw copies concatenated on the device, per-graph node offsets added with one op"""
        with torch.cuda.stream(side):
            up = [_dev_cached(p, device, (n, role) if n is not None else None) for p, n in zip(parts, names)]
            out = (torch.cat(up, 1) if len(up) > 1 else up[0]) + per_edge_offset
        keep_alive.append(out)
        return out

    b["ligand"].x = dv([d["ligand"].x for d in data_list], static=True, role="lig_x")
    b["ligand"].pos = fall[:3 * sum(nl)].view(-1, 3)
    b["ligand"].edge_mask = dv([d["ligand"].edge_mask for d in data_list], static=True, role="lig_edge_mask")
    b["ligand"].batch = idev[0]
    b["ligand", "ligand"].edge_index = edges([d["ligand", "ligand"].edge_index for d in data_list], idev[3], "lig_edge_index")
    b["ligand", "ligand"].edge_attr = dv([d["ligand", "ligand"].edge_attr for d in data_list], static=True, role="lig_edge_attr")
    b["receptor"].x = dv([d["receptor"].x for d in data_list], static=True, role="rec_x")
    b["receptor"].pos = dv([d["receptor"].pos for d in data_list], static=True, role="rec_pos")
    b["receptor"].batch = idev[1]
    b["receptor", "receptor"].edge_index = edges([d["receptor", "receptor"].edge_index for d in data_list], idev[4], "rec_edge_index")
    b.complex_t = {"tr": fdev[B], "rot": fdev[B + 1], "tor": fdev[B + 2]}
    # host copies of what the forward pass would otherwise read back from the device (each read-back is a pipeline bubble)
    b.host = {"t": t_host, "n_rot": [int(d["ligand"].edge_mask.sum()) for d in data_list], "nl": nl, "nr": nr}
    # columns of the batch's bond list that are rotatable bonds (the mask is host data: no boolean indexing on the device, which would
    # read the count back)
    b.rot_bond_cols = idev[2]
    b.lig_ptr, b.rec_ptr = idev[5], idev[6]        # node offsets of the graphs: the batched radius searches scan [ptr[b], ptr[b + 1])
    if keep is None:            # stand-alone use: the compute stream waits here; prepare_batch() hands an event to forward() instead
        torch.cuda.current_stream(device).wait_stream(side)
        _keep_until_main_passes(keep_alive, device)
    else:
        keep.extend(keep_alive)
    return b


_COPY_STREAMS = {}

# Device copies of host tensors that do not change between training steps (receptor features and positions, ligand features, bonds).
#   * keyed on the COMPLEX, not on the host tensor: (complex name, role, shape, dtype, device).  The bootstrapping buffer hands out
#     shallow copies that share a complex's tensors, the reference's loaders deep-copy them; either way the same complex comes back
#     under the same name, and a key that does not hold a storage address neither pins host memory nor depends on the allocator
#     recycling addresses.  Graphs without a string `name` are uploaded every step (no entry);
#   * every hit compares a fingerprint of the host tensor's CONTENTS (whole tensor below 4096 elements -- index / mask tensors, bond
#     lists --, else 32 strided elements + the last, ~5 us) with the one taken at upload time: two complexes that share a name, or a
#     tensor edited in place (re-centring, a numpy view, `.data`), re-upload and replace the entry instead of training on a stale
#     copy.  Still a guard, not a proof, for edits of the large feature matrices that miss all 33 probes (`dev_cache_clear()`);
#   * an entry remembers the stream it was uploaded on and an event behind the upload: a hit from another stream (the side stream changes
#     priority between eager and hipGraph-captured steps) waits for that event, and an entry that is evicted or replaced is kept alive
#     until the compute stream has passed (its last reader may still be queued);
#   * least-recently-used entries are evicted by bytes (default limit 4 GB of device copies), never the whole cache at once;
#   * `CBD_TRAIN_DEV_CACHE=0` or `dev_cache_configure(enabled=False)` turns it off (every step uploads its inputs).
_DEV_CACHE = __import__("collections").OrderedDict()      # key -> (device copy, fingerprint)
_DEV_CACHE_BYTES = [0]
_DEV_CACHE_CFG = {"enabled": os.environ.get("CBD_TRAIN_DEV_CACHE", "1") != "0", "limit": 4 << 30}


def dev_cache_configure(enabled=None, limit_bytes=None):
    if enabled is not None:
        _DEV_CACHE_CFG["enabled"] = bool(enabled)
        if not enabled:
            dev_cache_clear()
    if limit_bytes is not None:
        _DEV_CACHE_CFG["limit"] = int(limit_bytes)


def dev_cache_clear():
    _DEV_CACHE.clear()
    _DEV_CACHE_BYTES[0] = 0


def dev_cache_stats():
    return {"entries": len(_DEV_CACHE), "bytes": _DEV_CACHE_BYTES[0]}


_FP_FULL_BELOW = 4096      # elements: index / mask tensors and small feature blocks are compared in full


def _fingerprint(t: torch.Tensor):
    """Contents probe of a host tensor.  Tensors below _FP_FULL_BELOW elements (index and mask tensors, small feature blocks: where a
    same-shape ligand or bond list under a re-used name would differ in few places) are compared WHOLE; the large feature matrices by 32
    strided elements + the last (~5 us).  Deliberately NOT the storage address or the version counter: the reference's loaders deep-copy
    their graphs, and a deep copy of a cached complex must hit (tests/test_finetune_host.py::test_static_tensor_cache_policy)."""
    n = t.numel()
    if n == 0:
        return b""
    a = t.numpy().reshape(-1) if t.is_contiguous() else t.reshape(-1).numpy()       # a view of the host storage
    return a.tobytes() if n <= _FP_FULL_BELOW else a[::max(1, n // 32)][:32].tobytes() + a[-1:].tobytes()


def _dev_cached(t: torch.Tensor, device, ident=None):
    """Device copy of a host tensor that belongs to a complex (`ident` = (complex name, role)) and does not change between training
    steps, uploaded once (policy: comment above).  Device tensors pass through; without an identity the tensor is just uploaded."""
    if t.is_cuda:
        return t
    if ident is None or not _DEV_CACHE_CFG["enabled"]:
        return t.to(device, non_blocking=True)
    key = (ident, tuple(t.shape), t.dtype, dev_key(device))
    fp = _fingerprint(t)
    gpu = torch.device(device).type == "cuda"
    cur = torch.cuda.current_stream(device) if gpu else None
    hit = _DEV_CACHE.get(key)
    if hit is not None:
        if hit[1] == fp:
            _DEV_CACHE.move_to_end(key)
            if gpu and hit[3] != cur.cuda_stream:          # uploaded on another stream: order this stream behind the upload
                cur.wait_event(hit[2])
            return hit[0]
        _DEV_CACHE_BYTES[0] -= hit[0].numel() * hit[0].element_size()       # another complex under this name, or edited in place
        if gpu:
            _keep_until_main_passes([hit[0]], device)
        del _DEV_CACHE[key]
    d = t.to(device, non_blocking=True)
    nbytes = d.numel() * d.element_size()
    if nbytes > _DEV_CACHE_CFG["limit"]:
        return d
    while _DEV_CACHE and _DEV_CACHE_BYTES[0] + nbytes > _DEV_CACHE_CFG["limit"]:
        _, old = _DEV_CACHE.popitem(last=False)
        _DEV_CACHE_BYTES[0] -= old[0].numel() * old[0].element_size()
        if gpu:
            _keep_until_main_passes([old[0]], device)
    ev = None
    if gpu:
        ev = torch.cuda.Event()
        ev.record(cur)
    _DEV_CACHE[key] = (d, fp, ev, cur.cuda_stream if gpu else 0)
    _DEV_CACHE_BYTES[0] += nbytes
    return d


_SIDE_PRIORITY = [-1 if os.environ.get("CBD_TRAIN_SIDE_PRIO", "1") == "1" else 0]


def side_priority(high: bool):
    """Priority of the side stream that carries a step's input-only work.  Eager steps: HIGH -- its small kernels and copies (and the host
    read-backs that wait for them) must not queue behind the compute stream's millisecond kernels of the previous step.  hipGraph-captured
    steps (train_graph.py): NORMAL -- a high-priority queue dribbling ~350 tiny launches starves the graph running on the compute stream
    (a replay with two prepare() calls behind it took 96.7 ms instead of 23.5; tools/train_graph_check.py)."""
    _SIDE_PRIORITY[0] = -1 if high else 0


def _copy_stream(device):
    key = (dev_key(device), _SIDE_PRIORITY[0])
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=device, priority=_SIDE_PRIORITY[0])
    return _COPY_STREAMS[key]


# Tensors allocated on the side stream and read by the compute stream.  The caching allocator would hand their memory back to the side
# stream as soon as the last Python reference dies, possibly while the compute stream still reads it.  Tensor.record_stream() is the
# textbook answer, but with ~300 such tensors per step it leaves the side stream's pool permanently short of reusable blocks (every block
# waits for a compute-stream event) and the allocator falls back to hipMalloc: steps of 36..65 ms with the same kernels.  Instead the
# tensors of a step are kept alive here until an event recorded on the compute stream at the START OF THE NEXT STEP has completed -- all
# their uses (forward, loss, backward) are enqueued before that point -- and are then freed to the side pool in one go.
_KEEP_NOW = {}          # device -> tensors of the running step
_KEEP_OLD = {}          # device -> deque of (event, tensors)


def _keep_until_main_passes(tensors, device):
    if os.environ.get("CBD_TRAIN_RECORD_STREAM", "0") == "1":       # measurement switch: the textbook scheme
        main = torch.cuda.current_stream(device)
        for t in tensors:
            t.record_stream(main)
        return
    _KEEP_NOW.setdefault(dev_key(device), []).extend(tensors)


def _rotate_keep(device):
    """called at the start of every step: closes the previous step's set behind an event, frees the sets whose event has completed"""
    import collections
    key = dev_key(device)
    old = _KEEP_OLD.setdefault(key, collections.deque())
    cur = _KEEP_NOW.get(key)
    if cur:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        old.append((ev, cur))
        _KEEP_NOW[key] = []
    while old and old[0][0].query():
        old.popleft()
    while len(old) > 8:         # a host that runs eight steps ahead of the GPU: wait rather than grow
        old[0][0].synchronize()
        old.popleft()


def upload(t: torch.Tensor, device):
    """Host tensor -> device WITHOUT stalling the host on the compute stream.  A copy from pageable memory blocks the host until the
    stream it is issued on has drained; issued on the compute stream in the middle of a step that is a full pipeline flush (the
    host then enqueues the rest of the step with the GPU idle).  Here the copy goes through the side stream -- idle but for such
    copies -- and the compute stream waits for it on the device."""
    if t.is_cuda:
        return t
    side, main = _copy_stream(device), torch.cuda.current_stream(device)
    with torch.cuda.stream(side):
        d = t.to(device, non_blocking=True)
    main.wait_stream(side)
    _keep_until_main_passes([d], device)
    return d


# ----------------------------------------------------------------------------- graph ops (torch_cluster / torch_scatter semantics)
def radius_mask(x, y, r, batch_x, batch_y, max_num_neighbors=32):
    """[len(y), len(x)] mask of torch_cluster.radius: for every y the x's of the same graph with |x - y|^2 < r^2, first
    `max_num_neighbors` in index order."""
    d2 = torch.zeros(y.shape[0], x.shape[0], dtype=x.dtype, device=x.device)
    for k in range(x.shape[1]):
        diff = x[None, :, k] - y[:, None, k]
        d2 = d2 + diff * diff
    ok = (d2 < r * r) & (batch_y[:, None] == batch_x[None, :])
    rank = torch.cumsum(ok.to(torch.int32), dim=1)
    return ok & (rank <= max_num_neighbors)


def mask_edges(ok, count=None):
    """[2, E] (row 0 = index into y, row 1 = index into x) of a radius mask, row-major order.  With `count` (the number of set
    entries, already on the host) there is no device->host read-back here."""
    nz = torch.nonzero(ok) if count is None else torch.nonzero_static(ok, size=int(count))
    return nz.t().contiguous()


def radius(x, y, r, batch_x, batch_y, max_num_neighbors=32):
    """torch_cluster.radius (row 0 = index into y, row 1 = index into x)."""
    return mask_edges(radius_mask(x, y, r, batch_x, batch_y, max_num_neighbors))


def radius_graph_mask(x, r, batch, max_num_neighbors=32):
    """mask [centre, neighbour] of torch_cluster.radius_graph: the capped scan counts the centre itself, which is then dropped"""
    ok = radius_mask(x, x, r, batch, batch, max_num_neighbors + 1)
    return ok & ~torch.eye(x.shape[0], dtype=torch.bool, device=x.device)


def radius_graph(x, r, batch, max_num_neighbors=32):
    ei = mask_edges(radius_graph_mask(x, r, batch, max_num_neighbors))
    return torch.stack([ei[1], ei[0]], dim=0)   # [neighbour; centre]


def scatter_mean(src, index, dim_size):
    """torch_scatter.scatter(src, index, dim=0, dim_size=dim_size, reduce='mean') with the sum in a fixed order and the division by the
    clamped count inside the kernel (train_ops.scatter_mean: `cbd_segment_mean`, no atomics)."""
    return _scatter_mean_op(src, index, dim_size)


def take(x, idx):
    """x[idx] along dim 0 whose backward is a fixed-order segmented sum (train_ops.gather_rows) -- autograd's own backward for an
    index is an atomic index_add (non-deterministic), torch's fancy-index backward additionally sorts (30 % of the step's GPU time in
    the first profile, profiles/r01_h_train_*)."""
    return gather_rows(x, idx)


def gaussian_smearing(mod, dist):
    d = dist.view(-1, 1) - mod.offset.view(1, -1)
    return torch.exp(mod.coeff * torch.pow(d, 2))


def unit4(vec):
    """unit edge vector padded to 4 floats: the kernels' `vec` operand (sh = [1, sqrt3 * unit])."""
    return F.pad(F.normalize(vec, dim=-1), (0, 1))


def atom_encoder_index(enc, x_cat):
    """flat row index into the concatenation of the encoder's embedding tables, [N * n_tables] (input-only: built in `_prepare`)"""
    tables = enc.atom_embedding_list
    nf = len(tables)
    # keyed by the table sizes themselves: an id(enc) key can be handed to ANOTHER encoder once a model has been garbage-collected (the
    # ligand encoder of a new model then got the receptor encoder's offsets, or the other way round: wrong rows, or indices beyond
    # the table and a GPU memory fault -- seen as a 1-in-10 flake of the training tests in round 3)
    sizes = tuple(int(t.weight.shape[0]) for t in tables)
    key = (sizes, str(x_cat.device))
    offs = _TABLE_OFFSETS.get(key)
    if offs is None:
        offs = _TABLE_OFFSETS[key] = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]), dtype=torch.long).to(x_cat.device)
    return (x_cat[:, :nf].long() + offs).reshape(-1)


def atom_encoder(enc, idx, extra):
    """AtomEncoder.forward (models/score_model.py:30-41): sum of the categorical embeddings (+ Linear over [sum, extra features]).  All
    tables are looked up with ONE gather over a concatenated table (`idx` from atom_encoder_index: one edge grouping for its
    fixed-order backward instead of one per feature: 16 for the ligand)."""
    tables = enc.atom_embedding_list
    nf = len(tables)
    big = torch.cat([t.weight for t in tables], 0) if nf > 1 else tables[0].weight
    emb = take(big, idx)
    if nf > 1:
        emb = emb.view(idx.shape[0] // nf, nf, -1).sum(1)
    if enc.additional_features_dim > 0:
        emb = _linear(torch.cat([emb, extra], dim=1), enc.additional_features_embedder)
    return emb


_TABLE_OFFSETS = {}
_BN_MAPS = {}


def _bn_maps(irreps: str, device):
    """Constant index tensors of an irreps layout: column -> channel, channel-averaging matrix [D, F] (1/dim entries), and the
    columns of the 0e (scalar, even) fields, which are the only ones that are centred and biased."""
    key = (irreps, dev_key(device))
    m = _BN_MAPS.get(key)
    if m is None:
        col2chan, cols0e, ch = [], [], 0
        for mul, l, p in parse_irreps(irreps):
            d = 2 * l + 1
            for u in range(mul):
                if l == 0 and p == 1:
                    cols0e.append(len(col2chan))
                col2chan += [ch] * d
                ch += 1
        D, Fc = len(col2chan), ch
        c2c = torch.tensor(col2chan)
        A = torch.zeros(D, Fc)
        A[torch.arange(D), c2c] = 1.0 / torch.bincount(c2c, minlength=Fc).float()[c2c]
        lo = cols0e[0] if cols0e else 0
        assert cols0e == list(range(lo, lo + len(cols0e))), "the 0e columns of a layout are contiguous"
        expand = torch.zeros(Fc, D)
        expand[c2c, torch.arange(D)] = 1.0          # [F, D] 0/1: per-channel factor -> per-column factor as a matrix product
        fields, col, i0 = [], 0, 0
        for mul, l, p in parse_irreps(irreps):
            for u in range(mul):
                fields.append([col, 2 * l + 1, i0 if (l == 0 and p == 1) else -1])
                i0 += 1 if (l == 0 and p == 1) else 0
                col += 2 * l + 1
        m = {"col2chan": c2c.to(device), "avg": A.to(device), "expand": expand.to(device), "lo0e": lo, "n0e": len(cols0e), "D": D,
             "fields": torch.tensor(fields, dtype=torch.int32).to(device)}
        _BN_MAPS[key] = m
    return m


def irreps_batch_norm(bn, x, eps=1e-5, momentum=0.1, residual=None, exclude=None):
    """e3nn.nn.BatchNorm (0.5.0: affine, normalization='component', reduce='mean').  Training: per-channel batch mean of the 0e
    fields and batch mean of the squared (centred) components of every field, running averages updated with `momentum` -- one HIP
    launch forward, one backward (train_ops.IrrepsBatchNormFn; as torch ops ~14 + ~25 launches per call, and the step is host-bound
    at the reference's batch size); eval: running statistics, torch ops.  `residual` [N, <= D] is added to the leading columns of the
    result (the layer's  out + pad(node_attr), models/tensor_layers.py:211-213).  `exclude`: two row ranges left out of the batch
    statistics (the filler graph of a capacity-padded step, `prepare_batch(pad=...)`); their rows get no input gradient."""
    m = _bn_maps(bn.irreps, x.device)
    if bn.training and x.shape[0] > 0:
        return IrrepsBatchNormFn.apply(x, m["D"], bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, m["fields"], momentum, eps,
                                       exclude)
    out = irreps_batch_norm_torch(bn, x[:, :m["D"]], eps, momentum)
    return out if residual is None else out + F.pad(residual, (0, out.shape[1] - residual.shape[1]))


def irreps_batch_norm_torch(bn, x, eps=1e-5, momentum=0.1):
    """The same BatchNorm as torch ops: the eval-mode path (running statistics), and the formulation the HIP kernels are tested
    against in training mode (tests/test_gpu_train_step.py)."""
    m = _bn_maps(bn.irreps, x.device)
    lo, n0e, D = m["lo0e"], m["n0e"], m["D"]
    has0e = n0e > 0
    if has0e:   # the 0e fields are one contiguous column range: slices + pads, no column gathers
        mean = x[:, lo:lo + n0e].mean(dim=0) if bn.training else bn.running_mean
        x = x - F.pad(mean, (lo, D - lo - n0e))
    var = (x * x).mean(dim=0) @ m["avg"] if bn.training else bn.running_var
    if bn.training:
        with torch.no_grad():
            if has0e:
                bn.running_mean.mul_(1 - momentum).add_(momentum * mean.detach())
            bn.running_var.mul_(1 - momentum).add_(momentum * var.detach())
    # per-channel factor spread over the channel's columns by a 0/1 matrix product (exact: one non-zero term per column) -- an
    # index_select here has an atomic index_add as its backward (at::native::indexFuncLargeIndex<ReduceAdd> in the step's profile)
    out = x * (((var + eps).pow(-0.5) * bn.weight) @ m["expand"])
    if has0e:
        out = out + F.pad(bn.bias, (lo, D - lo - n0e))
    return out


# ----------------------------------------------------------------------------- layers
def conv_layer(layer, node_attr, edge_index, edge_attr, vec4, in_level, out_level, hub=None, group_sizes=None, bn_exclude=None):
    """TensorProductConvLayer.forward (models/tensor_layers.py:195-217) with FasterTensorProduct on the HIP op."""
    n, din = node_attr.shape
    dout = LEVEL_DIMS[out_level]
    if edge_index.shape[1] == 0:
        out = torch.zeros(n, dout, dtype=node_attr.dtype, device=node_attr.device)
    else:
        src, dst = edge_index[0], edge_index[1]
        xrow = gather_pad(node_attr, dst)           # node_attr[edge_dst] as the kernels' zero-padded 80-float rows
        sm = stream_map(in_level, out_level)
        fcs = [layer.fc] if layer.edge_groups == 1 else list(layer.fc)
        # all edge groups of the layer in ONE launch of the HIP op (forward and backward); the first Linear of every group's FCBlock
        # runs for all groups in one launch on slices of one buffer (train_ops.FcFirstStageFn), ReLU and Dropout fused (the groups of
        # a layer share the dropout rate)
        sizes = list(group_sizes) if group_sizes is not None else [edge_attr.shape[0]]
        live = [(fc, ne) for fc, ne in zip(fcs, sizes) if ne > 0]
        call = 0
        if hub is None:
            raise RuntimeError("conv_layer needs the step's StreamHub (train_forward._stream_hub): the model's tile streams are packed once per step")
        hub.fc_calls = call = getattr(hub, "fc_calls", 0) + 1
        hid = fc_first_stage(edge_attr, [ne for _, ne in live], [fc for fc, _ in live], seed=getattr(hub, "drop_seed", None), call=call)
        msg = TensorProductHubFn.apply(xrow, vec4, hid, hub.big, hub, in_level, out_level, tuple(ne for _, ne in live),
                                       tuple(hub.block(fc) for fc, _ in live))
        # the 80-float message rows go through the segmented mean and into the BatchNorm kernel as they are (it reads the layout's
        # `dout` columns): no slice copy of the [E, 80] tensor
        return irreps_batch_norm(layer.batch_norm, scatter_mean(msg, src, n), residual=node_attr, exclude=bn_exclude)
    return out + F.pad(node_attr, (0, dout - din))


def center_tensor_product(x, vec, w):
    """o3.FullyConnectedTensorProduct(74-irreps, '1x0e+1x1o', '2x1o+2x1e') with per-edge weights (final_conv.tp,
    models/tensor_layers.py:185): instruction-major weights [0e*1o->1o | 1o*0e->1o | 1o*1o->1e | 1e*0e->1e | 1e*1o->1o | 0o*1o->1e]."""
    if train_ops.FUSED_HEADS and x.is_cuda and x.shape[0] > 0:
        return train_ops.CenterTpFn.apply(x, vec, w)
    E = x.shape[0]
    v = SQ3 * F.normalize(vec, dim=-1)
    x0e, x1o, x1e, x0o = x[:, :32], x[:, 32:50].reshape(E, 6, 3), x[:, 50:68].reshape(E, 6, 3), x[:, 68:74]
    wa, wb = w[:, 0:64].reshape(E, 32, 2), w[:, 64:76].reshape(E, 6, 2)
    wc, wd = w[:, 76:88].reshape(E, 6, 2), w[:, 88:100].reshape(E, 6, 2)
    we, wf = w[:, 100:112].reshape(E, 6, 2), w[:, 112:124].reshape(E, 6, 2)
    vb = v[:, None, :].expand(E, 6, 3)
    s2 = 1.0 / math.sqrt(2.0)
    # per-edge contractions over u as broadcast products + sums (an einsum with a batch index runs as a batched library GEMM: a
    # handful of Tensile launches per head for [E, 6..32] operands)
    con = lambda wgt, y: (wgt.unsqueeze(-1) * y.unsqueeze(2)).sum(1)            # [E,u,w] x [E,u,k] -> [E,w,k]
    out1o = (con(wa, x0e.unsqueeze(-1) * v.unsqueeze(1)) + con(wb, x1o)
             + s2 * con(we, torch.linalg.cross(x1e, vb, dim=-1))) / math.sqrt(44.0)
    out1e = (s2 * con(wc, torch.linalg.cross(x1o, vb, dim=-1)) + con(wd, x1e)
             + con(wf, x0o.unsqueeze(-1) * v.unsqueeze(1))) / math.sqrt(18.0)
    return torch.cat([out1o.reshape(E, 6), out1e.reshape(E, 6)], dim=1)


def bond_tensor_product(x, edge_vec, bond_vec, w):
    """final_tp_tor (o3.FullTensorProduct('1x0e+1x1o', '2e')) followed by tor_bond_conv.tp (FCTP with two live paths):
    1o x T1 -> 32x0e (weights [0:192]) and 1e x T1 -> 32x0o ([192:384]); output order [0o | 0e].  T1 is the 1o block of the
    full product: (3/sqrt2)(b b^T - I/3)(sqrt3 v) for unit bond direction b and unit edge direction v (score_model.py:431-441)."""
    if train_ops.FUSED_HEADS and x.is_cuda and x.shape[0] > 0:
        return train_ops.BondTpFn.apply(x, edge_vec, bond_vec, w)
    E = x.shape[0]
    v = SQ3 * F.normalize(edge_vec, dim=-1)
    b = F.normalize(bond_vec, dim=-1)
    t1 = (3.0 / math.sqrt(2.0)) * (b * (b * v).sum(-1, keepdim=True) - v / 3.0)
    x1o, x1e = x[:, 32:50].reshape(E, 6, 3), x[:, 50:68].reshape(E, 6, 3)
    c = 1.0 / math.sqrt(6.0) / SQ3
    out0e = c * (w[:, :192].reshape(E, 6, 32) * (x1o * t1[:, None, :]).sum(-1).unsqueeze(-1)).sum(1)
    out0o = c * (w[:, 192:].reshape(E, 6, 32) * (x1e * t1[:, None, :]).sum(-1).unsqueeze(-1)).sum(1)
    return torch.cat([out0o, out0e], dim=1)


# ----------------------------------------------------------------------------- the model
_HUBS = weakref.WeakKeyDictionary()


TWO_STREAM_EMBEDDING = os.environ.get("CBD_TRAIN_TWO_STREAMS", "1") != "0"
_EMBED_STREAMS = {}


def _embed_side_stream(dev):
    """One side stream per (device, launching stream) for the ligand embedding chain / torsion head of the training forward (see
    forward()).  HIP multiplexes a process's streams onto a few hardware queues (four by default), and two streams that share a queue
    run one after the other: a side stream created after many others (bench.py's earlier legs create dozens) landed on the launching
    stream's queue in the full bench run and the overlap was gone (19.6 instead of 18.6 ms per step; alone: 18.6).  So four candidates
    are created back to back (they land on different queues) and each is PROBED once: with a ~1 ms spin kernel running on the launching
    stream, an event recorded on an idle candidate completes at once unless the candidate sits behind that kernel (the same test
    csrc/engine.hip::pick_setup_stream makes for the set-up streams of the sampler)."""
    cur = torch.cuda.current_stream(dev)
    k = (dev_key(dev), int(cur.cuda_stream))
    st = _EMBED_STREAMS.get(k)
    if st is None:
        import time
        cands = [torch.cuda.Stream(device=dev) for _ in range(4)]
        torch.cuda.synchronize(dev)
        st = cands[0]
        spin = getattr(torch.cuda, "_sleep", None)        # (private torch helper; without it the first candidate is taken unprobed)
        for c in cands if spin is not None else []:
            spin(2_000_000)                               # ~1 ms of GPU spin on the launching stream
            ev = torch.cuda.Event()
            ev.record(c)
            t0 = time.perf_counter()
            while not ev.query() and time.perf_counter() - t0 < 3e-4:
                pass
            free = ev.query()
            torch.cuda.synchronize(dev)
            if free:
                st = c
                break
        _EMBED_STREAMS[k] = st
    return st


def _stream_hub(model, dev) -> StreamHub:
    """the model's StreamHub (every FCBlock that feeds a FasterTensorProduct, with the irreps levels of its layer), built once"""
    hub = _HUBS.get(model)          # kept beside the model, not on it: a hub holds non-leaf tensors, which copy.deepcopy(model) refuses
    if hub is None or hub.src.device != canonical_device(dev):
        blocks = []
        for layers in (model.rec_emb_layers, model.lig_emb_layers):
            for l, layer in enumerate(layers):
                blocks.append((layer.fc, min(l, 3), min(l + 1, 3)))
        for layer in model.conv_layers:
            blocks += [(fc, 3, 3) for fc in ([layer.fc] if layer.edge_groups == 1 else list(layer.fc))]
        hub = _HUBS[model] = StreamHub(blocks, dev)
    return hub


class _Prepared:
    """input-only tensors of one step (see `_prepare`)"""

    def tensors(self):
        out = []
        for v in self.__dict__.values():
            if torch.is_tensor(v):
                out.append(v)
            elif isinstance(v, Csr):
                out += [v.index, v.perm, v.rowptr]
        return out


def _pad_edges(ei, bucket, a_lo, a_n, b_lo, b_n, distinct=False):
    """[2, E] -> [2, ceil(E / bucket) * bucket] with padding edges (a_lo + i % a_n, b_lo + i % b_n) -- nodes of the filler graph (a_lo, b_lo:
    device scalars).  `distinct`: both ends index the same node set; the second is shifted so that no edge is a self loop."""
    n = int(ei.shape[1])
    cap = -(-max(n, 1) // bucket) * bucket
    if cap == n:
        return ei
    i = torch.arange(cap - n, device=ei.device)
    a = i % a_n
    b = (a + 1 + (i // a_n) % (b_n - 1)) % b_n if distinct else i % b_n
    return torch.cat([ei, torch.stack([a_lo + a, b_lo + b])], 1)


_DROPOUT_ACTIVE = weakref.WeakKeyDictionary()


def _dropout_active(model) -> bool:
    """does any FCBlock of the tensor-product layers drop units right now?  (cached per model and training flag)"""
    hit = _DROPOUT_ACTIVE.get(model)
    if hit is None or hit[0] != model.training:
        layers = list(model.rec_emb_layers) + list(model.lig_emb_layers) + list(model.conv_layers)
        on = any(isinstance(m, torch.nn.Dropout) and m.p > 0 and m.training for layer in layers for m in layer.modules())
        hit = _DROPOUT_ACTIVE[model] = (model.training, on)
    return hit[1]


def _prepare(model, data, host, dev, csr_cache=None, pad=None) -> _Prepared:
    """Everything of the forward pass that depends on the batch alone, not on the weights (called under the side stream).
    `pad`: see prepare_batch (the last graph of the batch is the filler; `pad` is completed with the ranges the forward pass needs)."""
    g = _Prepared()
    lig, rec = data["ligand"], data["receptor"]
    B = data.num_graphs
    ct = data.complex_t
    lig_batch, rec_batch = lig.batch, rec.batch
    g.tr_sigma, _, _ = model.t_to_sigma(ct["tr"], ct["rot"], ct["tor"])
    if _dropout_active(model):
        # seed of this step's dropout masks in the fused FCBlock stage (csrc/train_fc.hip), drawn from torch's CPU generator (so that
        # torch.manual_seed reproduces a step) and kept in device memory (so that a hipGraph replay reads a fresh one from its inputs)
        g.drop_seed = torch.randint(0, 2 ** 62, (1,), dtype=torch.long).to(dev, non_blocking=True)
    lig_pos, rec_pos = lig.pos.float(), rec.pos.float()
    g.lig_pos = lig_pos
    t_host = host["t"] if host else {k: v.detach().cpu() for k, v in ct.items()}
    n_rot = host["n_rot"] if host else torch.bincount(lig_batch[data["ligand", "ligand"].edge_index[0][lig.edge_mask.bool()]], minlength=B).tolist()
    g.n_rot = n_rot
    _, rot_sigma_h, tor_sigma_h = model.t_to_sigma(t_host["tr"], t_host["rot"], t_host["tor"])
    g.so3_norm = so3.score_norm(rot_sigma_h).unsqueeze(1).to(dev, non_blocking=True)
    tor_sigma_edge = np.repeat(tor_sigma_h.numpy(), n_rot)
    g.torus_norm = torch.sqrt(torch.tensor(torus.score_norm(tor_sigma_edge)).float()).to(dev, non_blocking=True) if sum(n_rot) else None
    bond_ei = data["ligand", "ligand"].edge_index.long()
    edge_mask = lig.edge_mask.bool()
    # the three radius graphs of the step: masks first, ONE read-back of the three edge counts, then compaction without read-backs
    cutoff = (g.tr_sigma * 3 + 20).unsqueeze(1)
    bonds = bond_pos = None
    if sum(n_rot):
        cols = getattr(data, "rot_bond_cols", None)
        bonds = bond_ei.index_select(1, cols) if cols is not None else bond_ei[:, edge_mask]
        bond_pos = (lig_pos[bonds[0]] + lig_pos[bonds[1]]) / 2
    lig_ptr, rec_ptr = getattr(data, "lig_ptr", None), getattr(data, "rec_ptr", None)
    if lig_ptr is not None:         # HIP searches: count, one read-back for all three graphs, fill (train_ops.RadiusQuery)
        qs = [RadiusQuery(lig_pos, lig_pos, model.lig_max_radius, lig_ptr, lig_batch, 32 + 1, drop_self=True),
              RadiusQuery(rec_pos, lig_pos, 1.0, rec_ptr, lig_batch, 10000, cutoff=cutoff)]
        if bonds is not None:
            qs.append(RadiusQuery(lig_pos, bond_pos, model.lig_max_radius, lig_ptr, lig_batch[bonds[0]], 32))
        found = radius_queries(qs)
        ei, lr = found[0], found[1]
        t_edges = found[2] if bonds is not None else None
        if pad is not None:
            # padding edges between nodes of the filler graph (the last graph), spread over its atoms / residues / rotatable bonds (the
            # last columns of `bonds`)
            pad["edges_real"] = {"ll": int(ei.shape[1]), "lr": int(lr.shape[1]), "t": int(t_edges.shape[1]) if t_edges is not None else 0}
            nfl, nfr, nfb = int(host["nl"][-1]), int(host["nr"][-1]), int(n_rot[-1])
            l0, r0 = lig_ptr[-2], rec_ptr[-2]
            ei = _pad_edges(ei, pad["buckets"]["ll"], l0, nfl, l0, nfl, distinct=True)
            lr = _pad_edges(lr, pad["buckets"]["lr"], l0, nfl, r0, nfr)
            if t_edges is not None:
                b0 = torch.full_like(l0, int(bonds.shape[1]) - nfb)
                t_edges = _pad_edges(t_edges, pad["buckets"]["t"], b0, nfb, l0, nfl)
    else:                           # a batch collated elsewhere: dense masks, one read-back of their three counts
        if pad is not None:
            raise RuntimeError("capacity padding needs the collation of train_forward.collate (node offsets of the graphs)")
        m_ll = radius_graph_mask(lig_pos, model.lig_max_radius, lig_batch)
        m_lr = radius_mask(rec_pos / cutoff[rec_batch], lig_pos / cutoff[lig_batch], 1, rec_batch, lig_batch, max_num_neighbors=10000)
        m_t = radius_mask(lig_pos, bond_pos, model.lig_max_radius, lig_batch, lig_batch[bonds[0]]) if bonds is not None else None
        n_edges = torch.stack([m.sum() for m in (m_ll, m_lr, m_t) if m is not None]).tolist()
        ei, lr = mask_edges(m_ll, n_edges[0]), mask_edges(m_lr, n_edges[1])
        t_edges = mask_edges(m_t, n_edges[2]) if m_t is not None else None
    radius_edges = torch.stack([ei[1], ei[0]], dim=0)          # [neighbour; centre]
    g.lr = lr
    r_ei = g.r_ei = data["receptor", "receptor"].edge_index.long()
    nL, nR = lig_pos.shape[0], rec_pos.shape[0]
    g.nL = nL

    # receptor graph (score_model.py:524-546)
    # edge vectors, unit vectors and distance expansions: one launch per edge set (train_ops.edge_geometry)
    _, g.r_vec4, g.r_smear = edge_geometry(rec_pos, rec_pos, r_ei[0], r_ei[1], model.rec_distance_expansion)
    g.rec_cat = atom_encoder_index(model.rec_node_embedding, rec.x)
    g.rec_batch_src = rec_batch[r_ei[0]]
    # ligand graph (score_model.py:492-522)
    l_ei = g.l_ei = torch.cat([bond_ei, radius_edges], 1)
    g.l_attr0 = torch.cat([data["ligand", "ligand"].edge_attr.float(),
                           torch.zeros(radius_edges.shape[1], model.in_lig_edge_features, device=dev)], 0)
    _, g.l_vec4, g.l_smear = edge_geometry(lig_pos, lig_pos, l_ei[0], l_ei[1], model.lig_distance_expansion)
    g.lig_cat = atom_encoder_index(model.lig_node_embedding, lig.x)
    # cross graph (score_model.py:564-587), joint graph (score_model.py:354-362)
    _, lr_vec4, g.c_smear = edge_geometry(lig_pos, rec_pos, lr[0], lr[1], model.cross_distance_expansion)
    lr_j = torch.stack([lr[0], lr[1] + nL], 0)
    g.edge_index = torch.cat([l_ei, lr_j, r_ei + nL, torch.stack([lr_j[1], lr_j[0]], 0)], 1)
    g.vec4 = torch.cat([g.l_vec4, lr_vec4, g.r_vec4, -lr_vec4], 0)
    g.s1 = l_ei.shape[1]
    g.s2 = g.s1 + lr_j.shape[1]
    g.s3 = g.s2 + r_ei.shape[1]
    g.ei2 = g.edge_index[:, :g.s2].contiguous()          # the last interaction layer updates the ligand side only
    g.vec4_2 = g.vec4[:g.s2]
    # centre geometry (score_model.py:635-648)
    counts = (torch.tensor(host["nl"]).to(dev, non_blocking=True) if host and "nl" in host else torch.bincount(lig_batch, minlength=B)).unsqueeze(1)
    center = ScatterSumFn.apply(lig_pos, csr_of(lig_batch, B, csr_cache)) / counts    # fixed-order sum (an index_add_ would be atomic)
    c_raw, _, g.center_smear = edge_geometry(center, lig_pos, lig_batch, None, model.center_distance_expansion, raw=True, unit=False)
    g.c_raw = c_raw
    g.c_vec2 = c_raw[:, :3]
    # torsion graph (score_model.py:650-664)
    g.bonds, g.t_ei = bonds, None
    if t_edges is not None:
        t_ei = g.t_ei = t_edges
        t_raw, _, g.t_smear = edge_geometry(bond_pos, lig_pos, t_ei[0], t_ei[1], model.lig_distance_expansion, raw=True, unit=False)
        g.t_raw = t_raw
        g.t_vec = t_raw[:, :3]
        g.bond_vec_e = (lig_pos[bonds[1]] - lig_pos[bonds[0]])[t_ei[0]]
    # edge groupings of every index tensor the step gathers / scatters through (cached per tensor: the later csr_of calls hit)
    nJ = nL + nR
    n_tab = lambda enc: sum(t.weight.shape[0] for t in enc.atom_embedding_list)
    warm = [(g.rec_cat, n_tab(model.rec_node_embedding)), (g.lig_cat, n_tab(model.lig_node_embedding)), (r_ei[0], nR), (r_ei[1], nR), (rec_batch, B), (g.rec_batch_src, B), (lig_batch, B), (l_ei[0], nL), (l_ei[1], nL), (lr[0], nL),
            (g.edge_index[0], nJ), (g.edge_index[1], nJ), (g.ei2[0], nJ), (g.ei2[1], nJ)]
    if g.t_ei is not None:
        warm += [(g.bonds[0], nL), (g.bonds[1], nL), (g.t_ei[1], nL), (g.t_ei[0], int(g.bonds.shape[1]))]
    if pad is not None:
        nLr = nL - int(host["nl"][-1])
        nRr = nR - int(host["nr"][-1])
        nb = int(g.bonds.shape[1]) if g.bonds is not None else 0
        pad.update(B_real=B - 1, T_real=int(sum(n_rot[:-1])),
                   ex_lig=((nLr, nL), (0, 0)), ex_rec=((nRr, nR), (0, 0)), ex_joint=((nLr, nL), (nL + nRr, nL + nR)),
                   ex_graph=((B - 1, B), (0, 0)), ex_bond=((nb - int(n_rot[-1]), nb), (0, 0)))
    cache = csr_cache if csr_cache is not None else {}
    csr_build_many(warm, cache)                 # all of them with ONE radix sort
    for k, (idx, n) in enumerate(warm):
        setattr(g, f"_csr{k}", csr_of(idx, n, cache))
    return g


class PreparedBatch:
    """A batch after `prepare_batch`: collated tensors, the input-only tensors of `_prepare`, the edge groupings, the side-stream
    tensors to keep alive, and the event the compute stream has to wait for."""

    def __init__(self, batch, g, csr, keep, event, n_graphs, pad=None):
        self.batch, self.g, self.csr, self.keep, self.event, self.num_graphs, self.pad = batch, g, csr, keep, event, n_graphs, pad


FILLER_NAME = "__filler__"
PAD_BUCKETS = {"ll": 1024, "lr": 8192, "t": 512}      # coarse: few distinct shapes = few graphs (the filler edges cost ~5 % of the step)


FILLER_LIG, FILLER_REC = 32, 31      # coprime: padding edge i joins atom i % 32 and residue i % 31 -- 992 distinct pairs, degrees stay small


def filler_complex(like: HeteroData) -> HeteroData:
    """The filler graph of a capacity-padded step: a chain of 32 ligand atoms (its inner bonds rotatable) and 31 residues, with the
    feature widths and dtypes of `like`.  It is appended to the batch as one more graph; the padding edges of the three radius graphs
    live inside it, SPREAD over its nodes (thousands of copies of one edge would make one node's segmented sums -- one wave per row --
    the long tail of every layer: 375 us instead of 45 per call in the first profile), so it is a connected component of its own:
    nothing flows between it and the real graphs, its rows are excluded from every BatchNorm statistic and its predictions from the
    loss."""
    f = HeteroData()
    lig, rec = like["ligand"], like["receptor"]
    rng = np.random.default_rng(7)
    nl, nr = FILLER_LIG, min(FILLER_REC, int(rec.x.shape[0]))
    f["ligand"].x = lig.x[:1].repeat(nl, 1).clone()
    # generic (non-colinear, non-planar) coordinates: a symmetric filler would make some equivariant outputs EXACTLY zero, and the heads
    # divide by their norms -- a 0/0 in the filler's (discarded) row would still send NaN into the shared weights' gradients
    step = rng.normal(size=(nl, 3))
    step = 1.5 * step / np.linalg.norm(step, axis=1, keepdims=True)
    f["ligand"].pos = torch.tensor(np.cumsum(step, axis=0), dtype=lig.pos.dtype)
    a = np.arange(nl - 1)
    ei = np.stack([np.stack([a, a + 1], 1).reshape(-1), np.stack([a + 1, a], 1).reshape(-1)])          # (0,1),(1,0),(1,2),(2,1),...
    f["ligand", "ligand"].edge_index = torch.tensor(ei, dtype=like["ligand", "ligand"].edge_index.dtype)
    ea = like["ligand", "ligand"].edge_attr
    f["ligand", "ligand"].edge_attr = torch.zeros(ei.shape[1], ea.shape[1], dtype=ea.dtype)
    f["ligand", "ligand"].edge_attr[:, 0] = 1
    mask = np.zeros(ei.shape[1], dtype=bool)
    mask[2 * np.arange(1, nl - 2)] = True                   # inner bonds, one direction each: 29 rotatable bonds
    f["ligand"].edge_mask = torch.from_numpy(mask)
    f["ligand"].mask_rotate = np.stack([np.arange(nl) > k for k in range(1, nl - 2)])
    f["receptor"].x = rec.x[:1].repeat(nr, 1).clone()
    rstep = rng.normal(size=(nr, 3))
    rstep = 3.8 * rstep / np.linalg.norm(rstep, axis=1, keepdims=True)
    f["receptor"].pos = torch.tensor(np.cumsum(rstep, axis=0) + np.array([0.0, 8.0, 0.0]), dtype=rec.pos.dtype)
    r = np.arange(nr)
    pairs = np.concatenate([np.stack([r, (r + 1) % nr]), np.stack([r, (r + 2) % nr])], axis=1)
    f["receptor", "receptor"].edge_index = torch.tensor(pairs, dtype=like["receptor", "receptor"].edge_index.dtype)
    f.complex_t = {k: torch.full((1,), 0.5) for k in ("tr", "rot", "tor")}
    f.name = FILLER_NAME
    return f


_FILLERS = {}


def _filler_for(d0: HeteroData) -> HeteroData:
    key = (tuple(d0["ligand"].x.shape[1:]), tuple(d0["receptor"].x.shape[1:]), d0["ligand"].x.dtype, d0["receptor"].x.dtype,
           tuple(d0["ligand", "ligand"].edge_attr.shape[1:]))
    f = _FILLERS.get(key)
    if f is None:
        f = _FILLERS[key] = filler_complex(d0)
    return f


def prepare_batch(model, data, dev, pad=None) -> PreparedBatch:
    """Collation + everything of the forward pass that depends on the batch alone (see `forward`).  Thread-safe with respect to a
    training step in flight on another host thread: it enqueues on the side stream only, fills its own caches, and does not touch the
    model's parameters.  Random numbers: ONE int64 -- the step's dropout seed -- is drawn from torch's global CPU generator per call (when
    the model has dropout), so that a step is reproducible under torch.manual_seed (tests/test_gpu_train_step.py); with the look-ahead
    thread of train_epoch (off by default) that draw happens on the worker thread, i.e. its position relative to the caller's own draws
    from the global generator is not fixed -- seed per step, or keep look_ahead off, where that matters.
    `pad` (True or a dict of bucket sizes {"ll", "lr", "t"}; `data` must be a list of graphs): capacity padding for the hipGraph-captured
    step (train_graph.py) -- a filler graph is appended and the three radius graphs of the step (ligand-ligand, ligand-receptor,
    torsion) are padded with edges INSIDE the filler up to the next multiple of their bucket, so that every tensor of the step has a
    shape that depends on the batch composition and the buckets only, not on the noise."""
    dev = torch.device(dev)
    if dev.type != "cuda":
        raise RuntimeError("the training forward runs on the MI355X only (HIP tensor-product kernels, no CPU fallback)")
    torch.cuda.set_device(dev)
    keep, csr = [], {}
    side = _copy_stream(dev)
    info = None
    if pad:
        if isinstance(data, Batch) or not isinstance(data, (list, tuple)):
            raise TypeError("capacity padding takes the list of graphs (the filler graph is collated with them)")
        buckets = dict(PAD_BUCKETS, **(pad if isinstance(pad, dict) else {}))
        data = list(data) + [_filler_for(data[0])]
        info = {"buckets": buckets}
    if isinstance(data, Batch):       # collated elsewhere: its tensors may still be in flight on the compute stream
        batch = data.to(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
    else:
        batch = collate(data, dev, keep=keep)
    with torch.cuda.stream(side):
        g = _prepare(model, batch, getattr(batch, "host", None), dev, csr, pad=info)
        event = torch.cuda.Event()
        event.record(side)
    keep.extend(g.tensors())
    return PreparedBatch(batch, g, csr, keep, event, batch.num_graphs, pad=info)


def forward(model, data):
    """(tr_pred [B,3], rot_pred [B,3], tor_pred [sum R], None) like the reference forward (score_model.py:333-449).  `data`: a list of
    noised graphs (what the reference's DataListLoader yields), their collation, or a `PreparedBatch`.  For a capacity-padded batch
    (`prepare_batch(pad=...)`) the filler graph's rows are excluded from the BatchNorm statistics and from the returned predictions."""
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("the training forward runs on the MI355X only (HIP tensor-product kernels, no CPU fallback)")
    if getattr(model, "asyncronous_noise_schedule", False):
        raise NotImplementedError("fine-tuning a model with an asyncronous noise schedule is outside the MI355X training path")
    # ---- everything that depends on the step's INPUTS only -- collation, the three radius graphs (their edge counts are read back to
    #      the host), host table look-ups, the joint edge list, edge vectors / distance expansions / centre geometry, and the edge
    #      groupings (stable sorts) of every index tensor the step gathers or scatters through -- is `prepare_batch`: it runs on the SIDE
    #      stream (the read-backs wait for that stream, not for the previous step's backward pass; the ~350 small launches overlap with
    #      it instead of lengthening the compute stream) and may be done AHEAD of the step, on another host thread, while the previous
    #      step's backward pass is being enqueued (training.train_epoch does).
    prep = data if isinstance(data, PreparedBatch) else prepare_batch(model, data, dev)
    use_csr_cache(prep.csr)     # edge groupings are per step (the graphs change with the poses)
    if prep.event is not None:  # (None: the static twin of a hipGraph capture, train_graph.py -- its tensors are already in place)
        _rotate_keep(dev)
        torch.cuda.current_stream(dev).wait_event(prep.event)
        _keep_until_main_passes(prep.keep, dev)
    data, g = prep.batch, prep.g
    pad = prep.pad or {}
    ex_lig, ex_rec, ex_joint, ex_graph, ex_bond = (pad.get(k) for k in ("ex_lig", "ex_rec", "ex_joint", "ex_graph", "ex_bond"))
    ns = model.ns
    lig, rec = data["ligand"], data["receptor"]
    B = data.num_graphs
    ct = data.complex_t
    lig_batch, rec_batch = lig.batch, rec.batch
    tr_sigma, lig_pos, n_rot, r_ei, l_ei, lr, edge_index = g.tr_sigma, g.lig_pos, g.n_rot, g.r_ei, g.l_ei, g.lr, g.edge_index
    s1, s2, s3, nL = g.s1, g.s2, g.s3, g.nL

    hub = _stream_hub(model, dev)
    hub.pack()
    # (the time embeddings first: the ligand chain needs them, and its side stream forks from HERE -- behind the packed streams, the batch
    #  tensors and these two small ops, in front of the receptor chain's launches)
    graph_sigma_emb = model.timestep_emb_func(ct["tr"])
    node_sigma_emb = take(graph_sigma_emb, lig_batch)
    fork_ev = None
    if TWO_STREAM_EMBEDDING and dev.type == "cuda" and not torch.cuda.is_current_stream_capturing():     # (a captured step stays on one stream)
        fork_ev = torch.cuda.Event()
        fork_ev.record(torch.cuda.current_stream(dev))
    hub.drop_seed, hub.fc_calls = getattr(g, "drop_seed", None), 0       # the step's dropout stream (train_ops.fc_first_stage)
    M = lambda seq, x, call: _mlp(seq, x, seed=hub.drop_seed, call=call)      # embeddings / heads: Linear (+ ReLU + Dropout) on the HIP kernels

    # ---- receptor embedding (score_model.py:297-326), recomputed with gradients every step
    rec_edge_attr = M(model.rec_edge_embedding, g.r_smear, 100)
    rec_node = atom_encoder(model.rec_node_embedding, g.rec_cat, rec.x[:, 1:].float())
    for l, layer in enumerate(model.rec_emb_layers):
        ea = edge_cat(rec_edge_attr, rec_node, r_ei[0], r_ei[1])
        rec_node = conv_layer(layer, rec_node, r_ei, ea, g.r_vec4, min(l, 3), min(l + 1, 3), hub, bn_exclude=ex_rec)
    rec_sigma_emb = M(model.rec_sigma_embedding, graph_sigma_emb, 110)
    rec_node = torch.cat([rec_node[:, :ns] + take(rec_sigma_emb, rec_batch), rec_node[:, ns:]], dim=1)
    rec_edge_attr = rec_edge_attr + take(rec_sigma_emb, g.rec_batch_src)

    # ---- ligand graph + embedding (score_model.py:492-522, 282-295)
    # The ligand chain is independent of the receptor chain above until the joint graph, and its launches are tiny (a few dozen 32-edge
    # waves on a 256-CU chip: pure latency, ~25 small kernels + 3 x 4 tensor-product kernels forward and backward).  It runs on a SIDE
    # stream, forked here from everything queued so far and joined before the joint graph: the GPU overlaps it with the receptor chain
    # (queued first, above), and autograd replays the same split in the backward pass (a node's backward runs on its forward's stream).
    # Same kernels on the same inputs -> bitwise the single-stream step.  (The receptor chain's launches are queued BEFORE the fork, so
    # the fork point lies behind them in stream order: the side stream must not wait for them -> it forks from an event recorded at
    # the top of this function, behind the time embeddings the chain reads.)
    side = _embed_side_stream(dev) if fork_ev is not None else None
    hub.side_stream = side          # _HubFn.backward waits for it: the side stream's tensor-product calls write hub.grads (train_ops._HubFn)
    cur = torch.cuda.current_stream(dev) if side is not None else None
    if side is not None:
        side.wait_event(fork_ev)
    with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
        l_attr = torch.cat([g.l_attr0, take(node_sigma_emb, l_ei[0]), g.l_smear], 1)
        lig_node = atom_encoder(model.lig_node_embedding, g.lig_cat, node_sigma_emb)
        lig_edge_attr = M(model.lig_edge_embedding, l_attr, 120)
        for l, layer in enumerate(model.lig_emb_layers):
            ea = edge_cat(lig_edge_attr, lig_node, l_ei[0], l_ei[1])
            lig_node = conv_layer(layer, lig_node, l_ei, ea, g.l_vec4, min(l, 3), min(l + 1, 3), hub, bn_exclude=ex_lig)
    if side is not None:
        cur.wait_stream(side)

    # ---- cross graph (score_model.py:345-352, 564-587)
    lr_edge_attr = M(model.cross_edge_embedding, torch.cat([take(node_sigma_emb, lr[0]), g.c_smear], 1), 130)

    # ---- joint graph, interaction layers (score_model.py:354-376)
    node = torch.cat([lig_node, rec_node], 0)
    edge_attr = torch.cat([lig_edge_attr, lr_edge_attr, rec_edge_attr, lr_edge_attr], 0)
    nconv = len(model.conv_layers)
    for l, layer in enumerate(model.conv_layers):
        if l < nconv - 1:
            ea = edge_cat(edge_attr, node, edge_index[0], edge_index[1])
            node = conv_layer(layer, node, edge_index, ea, g.vec4, 3, 3, hub, group_sizes=[s1, s2 - s1, s3 - s2, ea.shape[0] - s3],
                              bn_exclude=ex_joint)
        else:
            ea = edge_cat(edge_attr[:s2], node, g.ei2[0], g.ei2[1])
            node = conv_layer(layer, node, g.ei2, ea, g.vec4_2, 3, 3, hub, group_sizes=[s1, s2 - s1], bn_exclude=ex_joint)
    lig_node = node[:nL]
    fork_heads = None
    if side is not None:      # the torsion head (below) runs on the side stream next to the centre convolution: both read lig_node only
        fork_heads = torch.cuda.Event()
        fork_heads.record(cur)

    # ---- centre convolution -> translation / rotation scores (score_model.py:393-420, 635-648)
    c_attr = torch.cat([g.center_smear, node_sigma_emb], 1)
    c_attr = torch.cat([M(model.center_edge_embedding, c_attr, 140), lig_node[:, :ns]], -1)
    gp = scatter_mean(center_tensor_product(lig_node, g.c_vec2, M(model.final_conv.fc, c_attr, 150)), lig_batch, B)
    gp = irreps_batch_norm(model.final_conv.batch_norm, gp, exclude=ex_graph)
    so3_norm = g.so3_norm
    if pad:     # the heads see the real graphs only: the filler's (zero) row would be a 0 / 0 in the normalisations below
        nb_ = pad["B_real"]
        gp, graph_sigma_emb, tr_sigma, so3_norm = gp[:nb_], graph_sigma_emb[:nb_], tr_sigma[:nb_], so3_norm[:nb_]
    tr_pred = gp[:, :3] + gp[:, 6:9]
    rot_pred = gp[:, 3:6] + gp[:, 9:]
    tr_norm = torch.linalg.vector_norm(tr_pred, dim=1).unsqueeze(1)
    tr_pred = tr_pred / tr_norm * M(model.tr_final_layer, torch.cat([tr_norm, graph_sigma_emb], dim=1), 160)
    rot_norm = torch.linalg.vector_norm(rot_pred, dim=1).unsqueeze(1)
    rot_pred = rot_pred / rot_norm * M(model.rot_final_layer, torch.cat([rot_norm, graph_sigma_emb], dim=1), 170)
    tr_pred = tr_pred / tr_sigma.unsqueeze(1)
    rot_pred = rot_pred * so3_norm

    if model.no_torsion or sum(n_rot) == 0:
        return tr_pred, rot_pred, torch.empty(0, device=dev), None

    # ---- torsion head (score_model.py:431-448, 650-664); on the side stream, forked behind the joint layers (see the ligand chain above)
    if fork_heads is not None:
        side.wait_event(fork_heads)
    with (torch.cuda.stream(side) if fork_heads is not None else contextlib.nullcontext()):
        t_ei, bonds = g.t_ei, g.bonds
        t_attr = M(model.final_edge_embedding, g.t_smear, 180)
        bond_attr = take(lig_node, bonds[0]) + take(lig_node, bonds[1])
        t_attr = torch.cat([t_attr, take(lig_node[:, :ns], t_ei[1]), take(bond_attr[:, :ns], t_ei[0])], -1)
        msg = bond_tensor_product(take(lig_node, t_ei[1]), g.t_vec, g.bond_vec_e, M(model.tor_bond_conv.fc, t_attr, 190))
        tor = scatter_mean(msg, t_ei[0], bonds.shape[1])
        tor = irreps_batch_norm(model.tor_bond_conv.batch_norm, tor, exclude=ex_bond)
        torus_norm = g.torus_norm
        if pad:
            tor, torus_norm = tor[:pad["T_real"]], torus_norm[:pad["T_real"]]
        tor_pred = M(model.tor_final_layer, tor, 200).squeeze(1)
        tor_pred = tor_pred * torus_norm   # sqrt(torus.score_norm(sigma_tor of the bond's graph)), score_model.py:443-447
    if fork_heads is not None:
        cur.wait_stream(side)
    return tr_pred, rot_pred, tor_pred, None
