"""Differentiable forward pass of `TensorProductScoreModel` for the fine-tuning step (reference
models/score_model.py:282-449 under `model.train()`, called from utils/training.py:198 `model(data)`).

Heterogeneous batches (different complexes, one diffusion time each), training-mode e3nn BatchNorm (batch statistics, running
averages updated), Dropout where the reference has it.  Every FasterTensorProduct layer -- > 99 % of the FLOPs -- runs on the
hand-written gfx950 kernels of csrc/tp_train.hip through `train_ops.tensor_product`; the surrounding small ops (embeddings,
radius graphs, scatter-mean, BatchNorm, the two e3nn heads) are autograd-visible torch ops on the same device.  The inference
path (`model.eval()` + sampling) does not come through here: it runs on the fused engine (engine.py).

Works in eval mode too (running statistics, no dropout), which the tests use to cross-check this path against the engine.
"""
from __future__ import annotations

import math
import weakref
from typing import List

import numpy as np
import torch
import torch.nn.functional as F

from . import so3, torus
from .hetero import Batch, HeteroData
from .score_model import parse_irreps
from .train_ops import (LEVEL_DIMS, NODE_STRIDE, IrrepsBatchNormFn, StreamHub, TensorProductHubFn, csr_of, first_linear, gather_rows, scatter_mean as _scatter_mean_op,
                        scatter_sum, stream_map, tensor_product)

SQ3 = math.sqrt(3.0)


# ----------------------------------------------------------------------------- batching
def collate(data_list: List[HeteroData], device) -> Batch:
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

    def dv(parts, dim=0, static=False):
        """per-graph host-to-device copies, concatenated on the device (a host-side cat of the 2 MB/complex language-model
        features costs more than the whole GPU step on a many-core host).  static=True: tensors that belong to the COMPLEX, not to the
        noised sample (receptor features / trace / graph, ligand atom types / bonds) -- the buffer hands out shallow copies that share
        them, so their device copies are cached across steps (_dev_cached) instead of being uploaded again for every sample."""
        with torch.cuda.stream(side):
            up = [(_dev_cached(p, device) if static else p.to(device, non_blocking=True)) for p in parts]
            out = torch.cat(up, dim) if len(up) > 1 else (up[0].clone() if static else up[0])
        keep_alive.append(out)
        return out

    def dv_edges(parts, offsets):
        """edge_index tensors: cached raw copies, per-graph node offsets added on the device"""
        with torch.cuda.stream(side):
            up = [_dev_cached(p, device) + o for p, o in zip(parts, offsets)]
            out = torch.cat(up, 1) if len(up) > 1 else up[0]
        keep_alive.append(out)
        return out

    b["ligand"].x = dv([d["ligand"].x for d in data_list], static=True)
    b["ligand"].pos = dv([d["ligand"].pos for d in data_list])
    b["ligand"].edge_mask = dv([d["ligand"].edge_mask for d in data_list], static=True)
    b["ligand"].batch = dv([torch.repeat_interleave(torch.arange(len(nl)), torch.tensor(nl))])
    b["ligand", "ligand"].edge_index = dv_edges([d["ligand", "ligand"].edge_index for d in data_list], lo)
    b["ligand", "ligand"].edge_attr = dv([d["ligand", "ligand"].edge_attr for d in data_list], static=True)
    b["receptor"].x = dv([d["receptor"].x for d in data_list], static=True)
    b["receptor"].pos = dv([d["receptor"].pos for d in data_list], static=True)
    b["receptor"].batch = dv([torch.repeat_interleave(torch.arange(len(nr)), torch.tensor(nr))])
    b["receptor", "receptor"].edge_index = dv_edges([d["receptor", "receptor"].edge_index for d in data_list], ro)
    b.complex_t = {k: dv([torch.cat([torch.as_tensor(d.complex_t[k], dtype=torch.float32).reshape(-1) for d in data_list])])
                   for k in ("tr", "rot", "tor")}
    # host copies of what the forward pass would otherwise read back from the device (each read-back is a pipeline bubble)
    b.host = {"t": {k: torch.cat([torch.as_tensor(d.complex_t[k], dtype=torch.float32).reshape(-1) for d in data_list]) for k in ("tr", "rot", "tor")},
              "n_rot": [int(d["ligand"].edge_mask.sum()) for d in data_list], "nl": nl}
    torch.cuda.current_stream(device).wait_stream(side)
    for t in keep_alive:
        t.record_stream(torch.cuda.current_stream(device))
    return b


_COPY_STREAMS = {}
_DEV_CACHE = {}          # (host storage pointer, shape, dtype, version, device) -> (host tensor kept alive, device copy)
_DEV_CACHE_BYTES = [0]
_DEV_CACHE_LIMIT = 4 << 30


def _dev_cached(t: torch.Tensor, device):
    """Device copy of a host tensor that does not change between training steps, uploaded once.  The key holds the tensor's storage
    address, shape, dtype and version counter; the entry keeps the host tensor alive, so the address cannot be recycled while the
    entry exists.  Device tensors pass through.  Bounded (4 GB): when full the cache is dropped and refills."""
    if t.is_cuda:
        return t
    key = (t.data_ptr(), tuple(t.shape), t.dtype, t._version, str(device))
    hit = _DEV_CACHE.get(key)
    if hit is not None:
        return hit[1]
    if _DEV_CACHE_BYTES[0] > _DEV_CACHE_LIMIT:
        _DEV_CACHE.clear()
        _DEV_CACHE_BYTES[0] = 0
    d = t.to(device, non_blocking=True)
    _DEV_CACHE[key] = (t, d)
    _DEV_CACHE_BYTES[0] += d.numel() * d.element_size()
    return d


def _copy_stream(device):
    key = str(device)
    if key not in _COPY_STREAMS:
        _COPY_STREAMS[key] = torch.cuda.Stream(device=device)
    return _COPY_STREAMS[key]


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
    d.record_stream(main)
    return d


# ----------------------------------------------------------------------------- graph ops (torch_cluster / torch_scatter semantics)
def radius(x, y, r, batch_x, batch_y, max_num_neighbors=32):
    """For every y the x's of the same graph with |x - y|^2 < r^2, first `max_num_neighbors` in index order
    (torch_cluster.radius; row 0 = index into y, row 1 = index into x)."""
    d2 = torch.zeros(y.shape[0], x.shape[0], dtype=x.dtype, device=x.device)
    for k in range(x.shape[1]):
        diff = x[None, :, k] - y[:, None, k]
        d2 = d2 + diff * diff
    ok = (d2 < r * r) & (batch_y[:, None] == batch_x[None, :])
    rank = torch.cumsum(ok.to(torch.int32), dim=1)
    ok = ok & (rank <= max_num_neighbors)
    row, col = torch.nonzero(ok, as_tuple=True)
    return torch.stack([row, col], dim=0)


def radius_graph(x, r, batch, max_num_neighbors=32):
    ei = radius(x, x, r, batch, batch, max_num_neighbors + 1)
    keep = ei[0] != ei[1]
    return torch.stack([ei[1][keep], ei[0][keep]], dim=0)   # [neighbour; centre]


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


def atom_encoder(enc, x_cat, extra):
    """AtomEncoder.forward (models/score_model.py:30-41): sum of the categorical embeddings (+ Linear over [sum, extra features]).  All
    tables are looked up with ONE gather over a concatenated table (one edge grouping for its fixed-order backward instead of one per
    feature: 16 for the ligand)."""
    tables = enc.atom_embedding_list
    nf = len(tables)
    if nf == 1:
        emb = take(tables[0].weight, x_cat[:, 0].long())
    else:
        key = (id(enc), str(x_cat.device))
        offs = _TABLE_OFFSETS.get(key)
        if offs is None:
            sizes = [t.weight.shape[0] for t in tables]
            offs = _TABLE_OFFSETS[key] = torch.tensor(np.concatenate([[0], np.cumsum(sizes)[:-1]]), dtype=torch.long).to(x_cat.device)
        big = torch.cat([t.weight for t in tables], 0)
        idx = (x_cat[:, :nf].long() + offs).reshape(-1)
        emb = take(big, idx).view(x_cat.shape[0], nf, -1).sum(1)
    if enc.additional_features_dim > 0:
        emb = enc.additional_features_embedder(torch.cat([emb, extra], dim=1))
    return emb


_TABLE_OFFSETS = {}
_BN_MAPS = {}


def _bn_maps(irreps: str, device):
    """Constant index tensors of an irreps layout: column -> channel, channel-averaging matrix [D, F] (1/dim entries), and the
    columns of the 0e (scalar, even) fields, which are the only ones that are centred and biased."""
    key = (irreps, str(device))
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


def irreps_batch_norm(bn, x, eps=1e-5, momentum=0.1, residual=None):
    """e3nn.nn.BatchNorm (0.5.0: affine, normalization='component', reduce='mean').  Training: per-channel batch mean of the 0e
    fields and batch mean of the squared (centred) components of every field, running averages updated with `momentum` -- one HIP
    launch forward, one backward (train_ops.IrrepsBatchNormFn; as torch ops ~14 + ~25 launches per call, and the step is host-bound
    at the reference's batch size); eval: running statistics, torch ops.  `residual` [N, <= D] is added to the leading columns of the
    result (the layer's  out + pad(node_attr), models/tensor_layers.py:211-213)."""
    m = _bn_maps(bn.irreps, x.device)
    if bn.training and x.shape[0] > 0:
        return IrrepsBatchNormFn.apply(x, m["D"], bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, m["fields"], momentum, eps)
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
def _fc_hidden(fc, x):
    # Linear -> ReLU -> Dropout; the last Linear lives inside the HIP op, the first one's weight gradient on cbd_outer_accum
    return fc[2](fc[1](first_linear(x, fc[0])))


def conv_layer(layer, node_attr, edge_index, edge_attr_groups, vec4, in_level, out_level, hub=None):
    """TensorProductConvLayer.forward (models/tensor_layers.py:195-217) with FasterTensorProduct on the HIP op."""
    n, din = node_attr.shape
    dout = LEVEL_DIMS[out_level]
    if edge_index.shape[1] == 0:
        out = torch.zeros(n, dout, dtype=node_attr.dtype, device=node_attr.device)
    else:
        src, dst = edge_index[0], edge_index[1]
        xpad = F.pad(node_attr, (0, NODE_STRIDE - din))
        sm = stream_map(in_level, out_level)
        groups = edge_attr_groups if isinstance(edge_attr_groups, (list, tuple)) else [edge_attr_groups]
        fcs = [layer.fc] if layer.edge_groups == 1 else list(layer.fc)
        # all edge groups of the layer in ONE launch of the HIP op (forward and backward); the first Linear of every group's
        # FCBlock (+ ReLU + Dropout) stays a torch op
        live = [(fc, ea) for fc, ea in zip(fcs, groups) if ea.shape[0] > 0]
        hid = torch.cat([_fc_hidden(fc, ea) for fc, ea in live], dim=0) if len(live) > 1 else _fc_hidden(*live[0])
        if hub is not None:     # the model's streams packed once per step (train_ops.StreamHub)
            msg = TensorProductHubFn.apply(take(xpad, dst), vec4, hid, hub.big, hub, in_level, out_level, tuple(ea.shape[0] for _, ea in live),
                                           tuple(hub.block(fc) for fc, _ in live))
        else:
            msg = tensor_product(take(xpad, dst), vec4, hid, [sm.stream(fc) for fc, _ in live], in_level, out_level,
                                 [ea.shape[0] for _, ea in live])
        # the 80-float message rows go through the segmented mean and into the BatchNorm kernel as they are (it reads the layout's
        # `dout` columns): no slice copy of the [E, 80] tensor
        return irreps_batch_norm(layer.batch_norm, scatter_mean(msg, src, n), residual=node_attr)
    return out + F.pad(node_attr, (0, dout - din))


def center_tensor_product(x, vec, w):
    """o3.FullyConnectedTensorProduct(74-irreps, '1x0e+1x1o', '2x1o+2x1e') with per-edge weights (final_conv.tp,
    models/tensor_layers.py:185): instruction-major weights [0e*1o->1o | 1o*0e->1o | 1o*1o->1e | 1e*0e->1e | 1e*1o->1o | 0o*1o->1e]."""
    E = x.shape[0]
    v = SQ3 * F.normalize(vec, dim=-1)
    x0e, x1o, x1e, x0o = x[:, :32], x[:, 32:50].reshape(E, 6, 3), x[:, 50:68].reshape(E, 6, 3), x[:, 68:74]
    wa, wb = w[:, 0:64].reshape(E, 32, 2), w[:, 64:76].reshape(E, 6, 2)
    wc, wd = w[:, 76:88].reshape(E, 6, 2), w[:, 88:100].reshape(E, 6, 2)
    we, wf = w[:, 100:112].reshape(E, 6, 2), w[:, 112:124].reshape(E, 6, 2)
    vb = v[:, None, :].expand(E, 6, 3)
    s2 = 1.0 / math.sqrt(2.0)
    out1o = (torch.einsum("euw,eu,ek->ewk", wa, x0e, v) + torch.einsum("euw,euk->ewk", wb, x1o)
             + s2 * torch.einsum("euw,euk->ewk", we, torch.linalg.cross(x1e, vb, dim=-1))) / math.sqrt(44.0)
    out1e = (s2 * torch.einsum("euw,euk->ewk", wc, torch.linalg.cross(x1o, vb, dim=-1)) + torch.einsum("euw,euk->ewk", wd, x1e)
             + torch.einsum("euw,eu,ek->ewk", wf, x0o, v)) / math.sqrt(18.0)
    return torch.cat([out1o.reshape(E, 6), out1e.reshape(E, 6)], dim=1)


def bond_tensor_product(x, edge_vec, bond_vec, w):
    """final_tp_tor (o3.FullTensorProduct('1x0e+1x1o', '2e')) followed by tor_bond_conv.tp (FCTP with two live paths):
    1o x T1 -> 32x0e (weights [0:192]) and 1e x T1 -> 32x0o ([192:384]); output order [0o | 0e].  T1 is the 1o block of the
    full product: (3/sqrt2)(b b^T - I/3)(sqrt3 v) for unit bond direction b and unit edge direction v (score_model.py:431-441)."""
    E = x.shape[0]
    v = SQ3 * F.normalize(edge_vec, dim=-1)
    b = F.normalize(bond_vec, dim=-1)
    t1 = (3.0 / math.sqrt(2.0)) * (b * (b * v).sum(-1, keepdim=True) - v / 3.0)
    x1o, x1e = x[:, 32:50].reshape(E, 6, 3), x[:, 50:68].reshape(E, 6, 3)
    c = 1.0 / math.sqrt(6.0) / SQ3
    out0e = c * torch.einsum("euw,eu->ew", w[:, :192].reshape(E, 6, 32), (x1o * t1[:, None, :]).sum(-1))
    out0o = c * torch.einsum("euw,eu->ew", w[:, 192:].reshape(E, 6, 32), (x1e * t1[:, None, :]).sum(-1))
    return torch.cat([out0o, out0e], dim=1)


# ----------------------------------------------------------------------------- the model
_HUBS = weakref.WeakKeyDictionary()


def _stream_hub(model, dev) -> StreamHub:
    """the model's StreamHub (every FCBlock that feeds a FasterTensorProduct, with the irreps levels of its layer), built once"""
    hub = _HUBS.get(model)          # kept beside the model, not on it: a hub holds non-leaf tensors, which copy.deepcopy(model) refuses
    if hub is None or hub.src.device != dev:
        blocks = []
        for layers in (model.rec_emb_layers, model.lig_emb_layers):
            for l, layer in enumerate(layers):
                blocks.append((layer.fc, min(l, 3), min(l + 1, 3)))
        for layer in model.conv_layers:
            blocks += [(fc, 3, 3) for fc in ([layer.fc] if layer.edge_groups == 1 else list(layer.fc))]
        hub = _HUBS[model] = StreamHub(blocks, dev)
    return hub


def forward(model, data):
    """(tr_pred [B,3], rot_pred [B,3], tor_pred [sum R], None) like the reference forward (score_model.py:333-449)."""
    dev = next(model.parameters()).device
    if dev.type != "cuda":
        raise RuntimeError("the training forward runs on the MI355X only (HIP tensor-product kernels, no CPU fallback)")
    if getattr(model, "asyncronous_noise_schedule", False):
        raise NotImplementedError("fine-tuning a model with an asyncronous noise schedule is outside the MI355X training path")
    from .train_ops import clear_csr_cache
    clear_csr_cache()          # edge groupings are per step (the graphs change with the poses)
    data = collate(data, dev)
    ns = model.ns
    lig, rec = data["ligand"], data["receptor"]
    B = data.num_graphs
    ct = data.complex_t
    lig_batch, rec_batch = lig.batch, rec.batch

    # ---- everything that needs a device->host read-back (edge counts of the three radius graphs, boolean masks) or a host table
    #      look-up depends on the step's INPUTS only, so it runs on the side stream: the read-backs wait for that stream, not for the
    #      previous step's backward pass on the compute stream, and the host enqueues the rest of the step without a single stall
    host = getattr(data, "host", None)
    main, side = torch.cuda.current_stream(dev), _copy_stream(dev)
    if host is None:                # a batch collated elsewhere: its tensors may still be in flight on the compute stream
        side.wait_stream(main)
    with torch.cuda.stream(side):
        tr_sigma, rot_sigma, tor_sigma = model.t_to_sigma(ct["tr"], ct["rot"], ct["tor"])
        lig_pos, rec_pos = lig.pos.float(), rec.pos.float()
        t_host = host["t"] if host else {k: v.detach().cpu() for k, v in ct.items()}
        n_rot = host["n_rot"] if host else torch.bincount(lig_batch[data["ligand", "ligand"].edge_index[0][lig.edge_mask.bool()]], minlength=B).tolist()
        _, rot_sigma_h, tor_sigma_h = model.t_to_sigma(t_host["tr"], t_host["rot"], t_host["tor"])
        so3_norm = so3.score_norm(rot_sigma_h).unsqueeze(1).to(dev, non_blocking=True)
        tor_sigma_edge = np.repeat(tor_sigma_h.numpy(), n_rot)
        torus_norm = torch.sqrt(torch.tensor(torus.score_norm(tor_sigma_edge)).float()).to(dev, non_blocking=True) if sum(n_rot) else None
        bond_ei = data["ligand", "ligand"].edge_index.long()
        edge_mask = lig.edge_mask.bool()
        radius_edges = radius_graph(lig_pos, model.lig_max_radius, lig_batch)
        cutoff = (tr_sigma * 3 + 20).unsqueeze(1)
        lr = radius(rec_pos / cutoff[rec_batch], lig_pos / cutoff[lig_batch], 1, rec_batch, lig_batch, max_num_neighbors=10000)
        bonds = bond_pos = t_ei = None
        if sum(n_rot):
            bonds = bond_ei[:, edge_mask]
            bond_pos = (lig_pos[bonds[0]] + lig_pos[bonds[1]]) / 2
            t_ei = radius(lig_pos, bond_pos, model.lig_max_radius, lig_batch, lig_batch[bonds[0]])
        # atoms per graph: known on the host (collate); torch.bincount would read its output size back from the device
        counts = (torch.tensor(host["nl"]).to(dev, non_blocking=True) if host and "nl" in host else torch.bincount(lig_batch, minlength=B)).unsqueeze(1)
        r_ei = data["receptor", "receptor"].edge_index.long()
    main.wait_stream(side)
    for t in (tr_sigma, rot_sigma, tor_sigma, lig_pos, rec_pos, so3_norm, torus_norm, bond_ei, edge_mask, radius_edges, cutoff, lr, bonds,
              bond_pos, t_ei, counts, r_ei):
        if t is not None:
            t.record_stream(main)

    hub = _stream_hub(model, dev)
    hub.pack()

    # ---- receptor embedding (score_model.py:297-326), recomputed with gradients every step
    r_vec = rec_pos[r_ei[1]] - rec_pos[r_ei[0]]
    rec_edge_attr = model.rec_edge_embedding(gaussian_smearing(model.rec_distance_expansion, r_vec.norm(dim=-1)))
    r_vec4 = unit4(r_vec)
    rec_node = atom_encoder(model.rec_node_embedding, rec.x[:, :1], rec.x[:, 1:].float())
    for l, layer in enumerate(model.rec_emb_layers):
        ea = torch.cat([rec_edge_attr, take(rec_node[:, :ns], r_ei[0]), take(rec_node[:, :ns], r_ei[1])], -1)
        rec_node = conv_layer(layer, rec_node, r_ei, ea, r_vec4, min(l, 3), min(l + 1, 3), hub)
    graph_sigma_emb = model.timestep_emb_func(ct["tr"])
    rec_sigma_emb = model.rec_sigma_embedding(graph_sigma_emb)
    rec_node = torch.cat([rec_node[:, :ns] + take(rec_sigma_emb, rec_batch), rec_node[:, ns:]], dim=1)
    rec_edge_attr = rec_edge_attr + take(rec_sigma_emb, rec_batch[r_ei[0]])

    # ---- ligand graph + embedding (score_model.py:492-522, 282-295)
    node_sigma_emb = take(graph_sigma_emb, lig_batch)
    l_ei = torch.cat([bond_ei, radius_edges], 1)
    l_attr = torch.cat([data["ligand", "ligand"].edge_attr.float(),
                        torch.zeros(radius_edges.shape[1], model.in_lig_edge_features, device=dev)], 0)
    l_vec = lig_pos[l_ei[1]] - lig_pos[l_ei[0]]
    l_attr = torch.cat([l_attr, take(node_sigma_emb, l_ei[0]), gaussian_smearing(model.lig_distance_expansion, l_vec.norm(dim=-1))], 1)
    l_vec4 = unit4(l_vec)
    lig_node = atom_encoder(model.lig_node_embedding, lig.x, node_sigma_emb)
    lig_edge_attr = model.lig_edge_embedding(l_attr)
    for l, layer in enumerate(model.lig_emb_layers):
        ea = torch.cat([lig_edge_attr, take(lig_node[:, :ns], l_ei[0]), take(lig_node[:, :ns], l_ei[1])], -1)
        lig_node = conv_layer(layer, lig_node, l_ei, ea, l_vec4, min(l, 3), min(l + 1, 3), hub)

    # ---- cross graph (score_model.py:345-352, 564-587)
    c_vec = rec_pos[lr[1]] - lig_pos[lr[0]]
    lr_attr = torch.cat([take(node_sigma_emb, lr[0]), gaussian_smearing(model.cross_distance_expansion, c_vec.norm(dim=-1))], 1)
    lr_edge_attr = model.cross_edge_embedding(lr_attr)
    lr_vec4 = unit4(c_vec)

    # ---- joint graph, interaction layers (score_model.py:354-376)
    nL = lig_node.shape[0]
    node = torch.cat([lig_node, rec_node], 0)
    lr_j = torch.stack([lr[0], lr[1] + nL], 0)
    edge_index = torch.cat([l_ei, lr_j, r_ei + nL, torch.stack([lr_j[1], lr_j[0]], 0)], 1)
    edge_attr = torch.cat([lig_edge_attr, lr_edge_attr, rec_edge_attr, lr_edge_attr], 0)
    vec4 = torch.cat([l_vec4, lr_vec4, r_vec4, -lr_vec4], 0)
    s1 = l_ei.shape[1]
    s2 = s1 + lr_j.shape[1]
    s3 = s2 + r_ei.shape[1]
    nconv = len(model.conv_layers)
    for l, layer in enumerate(model.conv_layers):
        if l < nconv - 1:
            ea = torch.cat([edge_attr, take(node[:, :ns], edge_index[0]), take(node[:, :ns], edge_index[1])], -1)
            node = conv_layer(layer, node, edge_index, [ea[:s1], ea[s1:s2], ea[s2:s3], ea[s3:]], vec4, 3, 3, hub)
        else:
            ea = torch.cat([edge_attr[:s2], take(node[:, :ns], edge_index[0, :s2]), take(node[:, :ns], edge_index[1, :s2])], -1)
            node = conv_layer(layer, node, edge_index[:, :s2], [ea[:s1], ea[s1:s2]], vec4[:s2], 3, 3, hub)
    lig_node = node[:nL]

    # ---- centre convolution -> translation / rotation scores (score_model.py:393-420, 635-648)
    center = scatter_sum(lig_pos, lig_batch, B) / counts
    c_vec2 = lig_pos - center[lig_batch]
    c_attr = torch.cat([gaussian_smearing(model.center_distance_expansion, c_vec2.norm(dim=-1)), node_sigma_emb], 1)
    c_attr = torch.cat([model.center_edge_embedding(c_attr), lig_node[:, :ns]], -1)
    gp = scatter_mean(center_tensor_product(lig_node, c_vec2, model.final_conv.fc(c_attr)), lig_batch, B)
    gp = irreps_batch_norm(model.final_conv.batch_norm, gp)
    tr_pred = gp[:, :3] + gp[:, 6:9]
    rot_pred = gp[:, 3:6] + gp[:, 9:]
    tr_norm = torch.linalg.vector_norm(tr_pred, dim=1).unsqueeze(1)
    tr_pred = tr_pred / tr_norm * model.tr_final_layer(torch.cat([tr_norm, graph_sigma_emb], dim=1))
    rot_norm = torch.linalg.vector_norm(rot_pred, dim=1).unsqueeze(1)
    rot_pred = rot_pred / rot_norm * model.rot_final_layer(torch.cat([rot_norm, graph_sigma_emb], dim=1))
    tr_pred = tr_pred / tr_sigma.unsqueeze(1)
    rot_pred = rot_pred * so3_norm

    if model.no_torsion or sum(n_rot) == 0:
        return tr_pred, rot_pred, torch.empty(0, device=dev), None

    # ---- torsion head (score_model.py:431-448, 650-664)
    t_vec = lig_pos[t_ei[1]] - bond_pos[t_ei[0]]
    t_attr = model.final_edge_embedding(gaussian_smearing(model.lig_distance_expansion, t_vec.norm(dim=-1)))
    bond_attr = take(lig_node, bonds[0]) + take(lig_node, bonds[1])
    t_attr = torch.cat([t_attr, take(lig_node[:, :ns], t_ei[1]), take(bond_attr[:, :ns], t_ei[0])], -1)
    bond_vec = lig_pos[bonds[1]] - lig_pos[bonds[0]]
    msg = bond_tensor_product(take(lig_node, t_ei[1]), t_vec, bond_vec[t_ei[0]], model.tor_bond_conv.fc(t_attr))
    tor = scatter_mean(msg, t_ei[0], bonds.shape[1])
    tor = irreps_batch_norm(model.tor_bond_conv.batch_norm, tor)
    tor_pred = model.tor_final_layer(tor).squeeze(1)
    tor_pred = tor_pred * torus_norm   # sqrt(torus.score_norm(sigma_tor of the bond's graph)), score_model.py:443-447
    return tr_pred, rot_pred, tor_pred, None
