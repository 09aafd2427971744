"""Autograd op for the fine-tuning step: `FasterTensorProduct` fed by the last Linear of its FCBlock, forward and backward
on hand-written gfx950 kernels (csrc/tp_train.hip through the C ABI `cbd_tp_forward` / `cbd_tp_backward`).

Replaces, inside `TensorProductConvLayer.forward` (reference models/tensor_layers.py:195-206),
    tp(node_attr[edge_dst], edge_sh, fc[-1](h))        with h = Dropout(ReLU(fc[0](edge_attr)))
and its autograd graph.  The per-edge [E, weight_numel] tensor is never built in the forward pass; backward writes its
gradient once (packed-row order) and finishes the Linear's backward with two rocBLAS GEMMs.

There is no PyTorch fallback: without the HIP library (or on CPU tensors) the op raises.
"""
from __future__ import annotations

import ctypes as C
import os
from functools import lru_cache

import numpy as np
import torch

from .hostcfg import dev_key

from .engine import load_library, _check

NODE_STRIDE = 80      # csrc/common.h
KDIM = 96
TILE_W_FLOATS = 48 * 64
LEVEL_DIMS = [32, 50, 68, 74]


def _bind(lib):   # symbols are bound (and checked against include/cbdock.h) by engine.load_library
    return lib


class StreamMap:
    """How one FCBlock's parameters (reference layout) land in the MFMA tile stream of level (IN, OUT):
       stream[i] = scale[i] * flat[src[i]],  flat = [fc.0.weight (96x96) | fc.0.bias | fc.3.weight (W x 96) | fc.3.bias]
    obtained by probing the library's own packer (`cbd_pack_conv_stream`, host only), plus the positions of the logical
    second-Linear matrix W2p[(tile-3)*32 + row][hidden unit] and bias b2p inside the stream."""

    def __init__(self, in_level: int, out_level: int):
        lib = _bind(load_library())
        self.in_level, self.out_level = in_level, out_level
        n = int(lib.cbd_conv_stream_floats(in_level, out_level))
        self.wp = int(lib.cbd_tp_packed_width(in_level, out_level))
        self.ntiles = self.wp // 32 + 3
        from .score_model import faster_tp_weight_numel, get_irrep_seq
        seq = get_irrep_seq(32, 6, False, True)
        self.weight_numel = W = faster_tp_weight_numel(seq[in_level], seq[out_level])
        sizes = [KDIM * KDIM, KDIM, W * KDIM, W]
        offs = np.concatenate([[0], np.cumsum(sizes)])

        def pack(parts):
            out = np.zeros(n, dtype=np.float32)
            arrs = [np.ascontiguousarray(p, dtype=np.float32) for p in parts]
            _check(lib.cbd_pack_conv_stream(in_level, out_level, *[a.ctypes.data_as(C.c_void_p) for a in arrs],
                                            out.ctypes.data_as(C.c_void_p)))
            return out

        scale = pack([np.ones(s, dtype=np.float32) for s in sizes])
        idx = pack([np.arange(offs[k] + 1, offs[k + 1] + 1, dtype=np.float32) for k in range(4)])
        live = scale != 0
        src = np.zeros(n, dtype=np.int64)
        src[live] = np.rint(idx[live].astype(np.float64) / scale[live].astype(np.float64)).astype(np.int64) - 1
        assert src.min() >= 0 and src.max() < offs[-1]
        self.n, self.scale_np, self.src_np = n, scale, src
        # every parameter sits at exactly ONE live stream position, so the backward of the stream gather is itself a gather through
        # the inverse map (autograd's own backward of index_select is an atomic index_add: at::native::indexFuncLargeIndex<ReduceAdd>)
        pos = np.flatnonzero(live)
        assert np.array_equal(np.sort(src[pos]), np.arange(offs[-1])), "the packer maps every FCBlock parameter to one stream slot"
        inv = np.empty(int(offs[-1]), dtype=np.int64)
        inv[src[pos]] = pos
        self.inv_np = inv
        # physical position of (tile T, row r, k-step s, lane half hf) (csrc/engine.hip::pack_rows_f32) and the hidden unit the
        # second Linear's k-step addresses (C/D register layout of the first GEMM)
        T, r, s, hf = np.meshgrid(np.arange(3, self.ntiles), np.arange(32), np.arange(48), np.arange(2), indexing="ij")
        phys = T * TILE_W_FLOATS + ((s >> 2) * 64 + hf * 32 + r) * 4 + (s & 3)
        hidden = 32 * (s // 16) + ((s % 16) & 3) + 8 * ((s % 16) >> 2) + 4 * hf
        w2p = np.zeros((self.wp, KDIM), dtype=np.int64)
        w2p[((T - 3) * 32 + r).ravel(), hidden.ravel()] = phys.ravel()
        self.w2p_np = w2p
        self.b2p_np = (self.ntiles + 1) * TILE_W_FLOATS + 96 + np.arange(self.wp)
        self._dev = {}

    def on(self, device):
        d = self._dev.get(dev_key(device))
        if d is None:
            d = {"src": torch.from_numpy(self.src_np).to(device), "scale": torch.from_numpy(self.scale_np).to(device),
                 "inv": torch.from_numpy(self.inv_np).to(device),
                 "w2p": torch.from_numpy(self.w2p_np.ravel()).to(device), "b2p": torch.from_numpy(self.b2p_np).to(device)}
            self._dev[dev_key(device)] = d
        return d

    def stream(self, fc) -> torch.Tensor:
        """Differentiable tile stream of an FCBlock `nn.Sequential(Linear, ReLU, Dropout, Linear)`."""
        w1, w2 = fc[0], fc[3]
        flat = torch.cat([w1.weight.reshape(-1), w1.bias, w2.weight.reshape(-1), w2.bias])
        d = self.on(flat.device)
        return _StreamFn.apply(flat, d["src"], d["scale"], d["inv"])


class _StreamFn(torch.autograd.Function):
    """stream = scale * flat[src]; backward = (g * scale)[inv]: a gather both ways (bitwise deterministic, no atomics)."""

    @staticmethod
    def forward(ctx, flat, src, scale, inv):
        ctx.save_for_backward(scale, inv)
        return flat.index_select(0, src) * scale

    @staticmethod
    def backward(ctx, g):
        scale, inv = ctx.saved_tensors
        return (g * scale).index_select(0, inv), None, None, None


@lru_cache(maxsize=None)
def stream_map(in_level: int, out_level: int) -> StreamMap:
    return StreamMap(in_level, out_level)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


class KernelTimer:
    """HIP-event timing of the two training kernels (measurement only; enabled by tools/train_bench.py).  Events are recorded on
    the stream the kernels are launched on and read back once, in `summary()`."""

    def __init__(self):
        self.enabled = False
        self.records = []      # (kind, in_level, out_level, E, start, stop)

    def wrap(self, kind, in_level, out_level, E, launch):
        if not self.enabled:
            return launch()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        self.records.append((kind, in_level, out_level, E, a, b))

    def summary(self):
        """{kind: (total ms, launches, algorithmic FLOPs)}: forward = the second Linear + CG contraction (2*96*W + 2*sum fan*m*dim
        per edge); backward = the same work re-computed plus as much again for g_w / g_mid (counted once: the MFMA part)."""
        torch.cuda.synchronize()
        out = {}
        for kind, i, o, E, a, b in self.records:
            sm = stream_map(i, o)
            w = sm.weight_numel
            fl = E * (2.0 * KDIM * w + 2.0 * _cg_flops(i, o))
            ms, n, f = out.get(kind, (0.0, 0, 0.0))
            out[kind] = (ms + a.elapsed_time(b), n + 1, f + fl)
        self.records = []
        return out


def _cg_flops(in_level, out_level):
    n1o, n1e, n0o = (6 if in_level >= 1 else 0), (6 if in_level >= 2 else 0), (6 if in_level >= 3 else 0)
    f0e, f1o = 32 + n1o, 32 + n1o + n1e
    f1e = n1o + n1e + n0o if out_level >= 2 else 0
    f0o = n1e + n0o if out_level >= 3 else 0
    return f0e * 32 + (f1o + f1e) * 6 * 3 + f0o * 6


TIMER = KernelTimer()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_handle():
    """HIP stream torch is currently enqueueing on, as a raw handle.  torch.cuda.current_stream() builds a Stream object through
    several Python layers (40 us per call, ~70 calls per training step); the C accessor the compiler back-ends use costs < 1 us."""
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class StreamHub:
    """The tile streams of ALL FCBlocks of the model, packed once per training step (the weights change every step), and the way back
    for their gradients.  Per block and step this replaces cat + gather + scale forward, the W2 gather / zero-fill / two scatters of
    the tensor-product backward and the stream gather's own backward (~12 launches x 28 blocks) by seven launches for the whole
    model -- the step is launch-bound at the reference's batch sizes.

    Only the second Linear's parameters enter the training kernels through the stream (the first Linear is a torch op with its own
    backward, FirstLinearFn), so the hub gathers from fc[3].weight / fc[3].bias; the first Linear's slots of the stream stay zero."""

    def __init__(self, blocks, device):
        """blocks: [(fc, in_level, out_level)] in a fixed order; fc = nn.Sequential(Linear, ReLU, Dropout, Linear)"""
        self.blocks = [(fc, stream_map(i, o)) for fc, i, o in blocks]
        self.index = {id(fc): b for b, (fc, _) in enumerate(self.blocks)}
        src, scale, w2p, b2p, p2g, pscale = [], [], [], [], [], []
        self.stream_off, self.flat_off, self.g_off, self.w_off = [], [], [], []
        so = fo = go = wo = 0
        for fc, sm in self.blocks:
            W = sm.weight_numel
            lo2 = KDIM * KDIM + KDIM                          # [w1 | b1 | w2 | b2]: the second Linear starts here
            live2 = (sm.scale_np != 0) & (sm.src_np >= lo2)
            src.append(np.where(live2, sm.src_np - lo2 + fo, 0))
            scale.append(np.where(live2, sm.scale_np, 0.0).astype(np.float32))
            w2p.append(sm.w2p_np.ravel() + so)
            self.stream_off.append(so)
            self.flat_off.append(fo)
            self.g_off.append(go)
            self.w_off.append(wo)
            wo += sm.wp * KDIM
            so += sm.n
            fo += W * KDIM + W
            go += sm.wp * KDIM
        self.gb_off = []
        for fc, sm in self.blocks:
            self.gb_off.append(go)
            go += sm.wp
        self.n_stream, self.n_flat, self.n_grad = so, fo, go
        # parameter element -> its slot in the gradient buffer [dW2p of every block | db2p of every block] and the packer's scale
        for b, (fc, sm) in enumerate(self.blocks):
            W = sm.weight_numel
            lo2 = KDIM * KDIM + KDIM
            pos = sm.inv_np[lo2:]                             # stream slot of every second-Linear parameter
            slot = np.full(sm.n, -1, dtype=np.int64)
            # gradient buffer: [dW2p (wp x 96) of every block | db2p (wp) of every block].  (db2p as a 97th column of the dW2p GEMM,
            # g_w^T [h | 1], was measured: the N = 97 GEMM costs 1.9 ms more per step than N = 96, the column sum it saves 1.3 ms.)
            slot[sm.w2p_np.ravel()] = self.g_off[b] + np.arange(sm.wp * KDIM)
            slot[sm.b2p_np] = self.gb_off[b] + np.arange(sm.wp)
            assert (slot[pos] >= 0).all(), "every second-Linear parameter sits in the W2p matrix or the bias table of its stream"
            p2g.append(slot[pos])
            pscale.append(sm.scale_np[pos])
        # transposed tile streams for the g_h pass (cbd_tp_backward_gh): slot -> position in w2p_all (the logical [wp, 96] matrices back
        # to back), or `n_w2p` = the appended zero for the padding tile behind every block
        tidx, self.t_off, to = [], [], 0
        n_w2p = wo
        f = np.arange(48)
        lane = np.arange(64)
        kb, st = f // 16, f % 16
        for b, (fc, sm) in enumerate(self.blocks):
            ntw = sm.wp // 32
            T = np.arange(ntw)
            w = T[:, None, None] * 32 + ((st & 3) + 8 * (st >> 2))[None, :, None] + 4 * (lane >> 5)[None, None, :]      # [ntw, 48, 64]
            k = (kb * 32)[None, :, None] + (lane & 31)[None, None, :]
            t_src = self.w_off[b] + w * KDIM + k
            t_slot = T[:, None, None] * TILE_W_FLOATS + (((f >> 2)[None, :, None] * 64 + lane[None, None, :]) * 4 + (f & 3)[None, :, None])
            t_map = np.full((ntw + 1) * TILE_W_FLOATS, n_w2p, dtype=np.int64)
            t_map[t_slot.ravel()] = t_src.ravel()
            tidx.append(t_map)
            self.t_off.append(to)
            to += t_map.size
        t = lambda a, dt: torch.from_numpy(np.concatenate(a).astype(dt)).to(device)
        self.t_idx = t(tidx, np.int64)
        self.big_t = None
        self.src, self.scale = t(src, np.int64), t(scale, np.float32)
        self.w2p_idx = t(w2p, np.int64)
        self.p2g, self.pscale = t(p2g, np.int64), t(pscale, np.float32)
        self.big = self.w2p_all = self.grads = None
        self.used = set()

    def pack(self):
        """-> the packed streams of this step (autograd-connected to every fc[3].weight / fc[3].bias)"""
        flat = torch.cat([p.reshape(-1) for fc, _ in self.blocks for p in (fc[3].weight, fc[3].bias)])
        self.big = _HubFn.apply(flat, self)
        with torch.no_grad():
            self.w2p_all = self.big.index_select(0, self.w2p_idx)          # the logical W2p matrices [wp, 96], back to back
            self.big_t = torch.cat([self.w2p_all, self.w2p_all.new_zeros(1)]).index_select(0, self.t_idx)     # transposed tile streams
            self.grads = torch.zeros(self.n_grad, device=flat.device, dtype=torch.float32)
        self.used = set()
        return self.big

    def block(self, fc) -> int:
        return self.index[id(fc)]

    def stream_ptr(self, b):
        return self.big.data_ptr() + 4 * self.stream_off[b]

    def stream_t_ptr(self, b):
        return self.big_t.data_ptr() + 4 * self.t_off[b]

    def w2p(self, b):
        sm = self.blocks[b][1]
        return self.w2p_all[self.w_off[b]:self.w_off[b] + sm.wp * KDIM].view(sm.wp, KDIM)

    def grad_views(self, b):
        """slots of block b in the gradient buffer: dW2p [wp, 96], db2p [wp]"""
        sm = self.blocks[b][1]
        if b in self.used:
            raise RuntimeError("an FCBlock feeds one tensor-product call per step")
        self.used.add(b)
        return (self.grads[self.g_off[b]:self.g_off[b] + sm.wp * KDIM].view(sm.wp, KDIM), self.grads[self.gb_off[b]:self.gb_off[b] + sm.wp])


class _HubFn(torch.autograd.Function):
    """big = scale * flat[src]; the gradient does not arrive through `big` (the tensor-product calls write dW2p / db2p of their blocks
    into hub.grads and return nothing for it) -- backward maps that buffer to the parameters: a gather and a scale."""

    @staticmethod
    def forward(ctx, flat, hub):
        ctx.hub = hub
        ctx.set_materialize_grads(False)
        return flat.index_select(0, hub.src) * hub.scale

    @staticmethod
    def backward(ctx, g):
        hub = ctx.hub
        # The tensor-product calls return NO gradient for `big` (they write dW2p / db2p into hub.grads), so autograd orders this node
        # behind them on the HOST but inserts no stream synchronisation for it.  The ligand chain's calls run on a side stream
        # (train_forward.forward): wait for it here, or this gather could read hub.grads before their kernels have written it.
        side = getattr(hub, "side_stream", None)
        if side is not None:
            torch.cuda.current_stream(hub.grads.device).wait_stream(side)
        return hub.grads.index_select(0, hub.p2g) * hub.pscale, None


_DW_SCRATCH = {}


def _dw_scratch(n_floats, device, stream=None):
    """One persistent buffer per (device, STREAM) for the partial blocks of cbd_tp_backward_dw (up to 113 MB per edge group, 22 groups
    per step, sizes changing with the edge counts: as fresh allocations they churn the caching allocator -- the fine-tuning leg of
    bench.py, which runs after the other legs have filled the cache, went from 36 to 48 ms per step).  The groups a stream runs use it
    one after the other; the backward passes of the two embedding chains run on two streams (train_forward.forward) and must not share."""
    key = (dev_key(device), int((stream if stream is not None else _stream_handle()).value or 0))
    buf = _DW_SCRATCH.get(key)
    if buf is None or buf.numel() < n_floats:
        buf = _DW_SCRATCH[key] = torch.empty(max(n_floats, 1 << 22), device=device, dtype=torch.float32)
    return buf[:n_floats]


DW_MAX_CHUNKS = int(os.environ.get("CBD_DW_MAX_CHUNKS", "96"))

class TensorProductHubFn(torch.autograd.Function):
    """TensorProductFn with the weight streams (and the way back for their gradients) in a StreamHub: `big` is an input only so that
    autograd runs the hub's backward after every tensor-product backward."""

    @staticmethod
    def forward(ctx, xrow, vec4, h, big, hub, in_level, out_level, group_edges, blocks):
        lib = _bind(load_library())
        xrow, vec4, h = xrow.contiguous().float(), vec4.contiguous().float(), h.contiguous().float()
        E = xrow.shape[0]
        assert xrow.shape == (E, NODE_STRIDE) and vec4.shape == (E, 4) and h.shape == (E, KDIM)
        assert len(group_edges) == len(blocks) and sum(group_edges) == E
        n = len(blocks)
        ge = (C.c_int64 * n)(*[int(x) for x in group_edges])
        ws = (C.c_void_p * n)(*[hub.stream_ptr(b) for b in blocks])
        msg = torch.empty(E, NODE_STRIDE, device=xrow.device, dtype=torch.float32)
        TIMER.wrap("fwd", in_level, out_level, E, lambda: _check(lib.cbd_tp_forward(
            in_level, out_level, n, ge, _ptr(xrow), _ptr(vec4), _ptr(h), ws, _ptr(msg), _stream_handle())))
        ctx.save_for_backward(xrow, vec4, h, big)
        ctx.meta = (hub, in_level, out_level, list(group_edges), list(blocks))
        return msg

    @staticmethod
    def backward(ctx, gmsg):
        xrow, vec4, h, big = ctx.saved_tensors
        hub, in_level, out_level, group_edges, blocks = ctx.meta
        lib = _bind(load_library())
        sm = stream_map(in_level, out_level)
        E, n = xrow.shape[0], len(blocks)
        gmsg = gmsg.contiguous().float()
        gx = torch.empty_like(xrow)
        # g_w is never stored: the g_h pass and the dW2p pass each re-form its tiles (the library-GEMM forms on a stored g_w that these
        # kernels replaced live in experiments/train_ops_reference.py, the equivalence reference of tests/test_gpu_train_op.py)
        ge = (C.c_int64 * n)(*[int(x) for x in group_edges])
        ws = (C.c_void_p * n)(*[hub.stream_ptr(b) for b in blocks])
        gh = torch.empty_like(h) if ctx.needs_input_grad[2] else None
        TIMER.wrap("bwd", in_level, out_level, E, lambda: _check(lib.cbd_tp_backward(
            in_level, out_level, n, ge, _ptr(xrow), _ptr(vec4), _ptr(h), ws, _ptr(gmsg), _ptr(gx), None, _stream_handle())))
        if gh is not None:
            # g_h on the matrix cores from re-formed g_w tiles (cbd_tp_backward_gh): one launch for all groups
            wt = (C.c_void_p * n)(*[hub.stream_t_ptr(b) for b in blocks])
            TIMER.wrap("gh", in_level, out_level, E, lambda: _check(lib.cbd_tp_backward_gh(
                in_level, out_level, n, ge, _ptr(xrow), _ptr(vec4), wt, _ptr(gmsg), _ptr(gh), _stream_handle())))
        # dW2p / db2p with the edges as the MFMA k dimension, ALL groups of the layer in one launch (cbd_tp_backward_dw_groups), the
        # per-chunk partial blocks added in a fixed order by one more (cbd_partial_reduce)
        wp = sm.wp
        live = [(int(ne), b) for ne, b in zip(group_edges, blocks) if ne]
        ng = len(live)
        # chunks per group: enough workgroups to fill the chip (14 tile groups x chunks), few enough that the reduction of the
        # partial blocks (167 k floats each at 74 -> 74) stays small next to the pass itself
        chunks = [max(1, min(DW_MAX_CHUNKS, ((ne + 31) // 32) // 4)) for ne, _ in live]
        width = wp * KDIM + wp
        part = _dw_scratch(sum(chunks) * width, xrow.device)
        assert sum(ne for ne, _ in live) == E
        ge_l = (C.c_int64 * ng)(*[ne for ne, _ in live])
        nc = (C.c_int32 * ng)(*chunks)
        TIMER.wrap("dw", in_level, out_level, E, lambda: _check(lib.cbd_tp_backward_dw_groups(
            in_level, out_level, ng, ge_l, nc, _ptr(xrow), _ptr(vec4), _ptr(h), _ptr(gmsg), _ptr(part), _stream_handle())))
        views = [hub.grad_views(b) for _, b in live]
        oa = (C.c_void_p * ng)(*[v[0].data_ptr() for v in views])
        ob = (C.c_void_p * ng)(*[v[1].data_ptr() for v in views])
        _check(lib.cbd_partial_reduce(ng, nc, width, wp * KDIM, _ptr(part), oa, ob, _stream_handle()))
        return (gx if ctx.needs_input_grad[0] else None), None, gh, None, None, None, None, None, None


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) for any nn.Linear of the step outside the FCBlocks' first stage, on cbd_linear_forward / _backward (fp32 MFMA, no
    library GEMM; csrc/train_fc.hip).  act 0: identity; act 1: dropout_p(relu(.)) with the hash mask of the step (`seed`, `call`)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, p, seed, call):
        lib = _bind(load_library())
        x2 = x.reshape(-1, x.shape[-1]).contiguous().float()
        E, K = x2.shape
        N = int(weight.shape[0])
        w = weight.contiguous().float()
        y = torch.empty(E, N, device=x.device, dtype=torch.float32)
        _check(lib.cbd_linear_forward(E, K, N, _ptr(x2), K, _ptr(w), None if bias is None else _ptr(bias.contiguous().float()), int(act), float(p),
                                      None if seed is None else _ptr(seed), int(call), _ptr(y), _stream_handle()))
        ctx.save_for_backward(x2, w, y if act else None)
        ctx.meta = (int(act), float(p), bias is not None, tuple(x.shape[:-1]))
        return y.reshape(tuple(x.shape[:-1]) + (N,))

    @staticmethod
    def backward(ctx, gy):
        x2, w, y = ctx.saved_tensors
        act, p, has_bias, lead = ctx.meta
        lib = _bind(load_library())
        E, K = x2.shape
        N = int(w.shape[0])
        if E == 0:
            return (torch.zeros(lead + (K,), device=gy.device), torch.zeros_like(w), torch.zeros(N, device=gy.device) if has_bias else None,
                    None, None, None, None)
        gy2 = gy.reshape(E, N).contiguous().float()
        gpre = torch.empty_like(gy2) if act else None
        gx = torch.empty(E, K, device=gy.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        n_chunks = int(lib.cbd_linear_backward_chunks(E))
        width = N * K + N
        partial = torch.empty(n_chunks, width, device=gy.device, dtype=torch.float32)
        _check(lib.cbd_linear_backward(E, K, N, _ptr(gy2), None if y is None else _ptr(y), _ptr(x2), K, _ptr(w), act, p,
                                       None if gpre is None else _ptr(gpre), None if gx is None else _ptr(gx), _ptr(partial), _stream_handle()))
        out = torch.empty(width, device=gy.device, dtype=torch.float32)
        oa, ob = (C.c_void_p * 1)(out.data_ptr()), (C.c_void_p * 1)(out.data_ptr() + 4 * N * K)
        _check(lib.cbd_partial_reduce(1, (C.c_int32 * 1)(n_chunks), width, N * K, _ptr(partial), oa, ob, _stream_handle()))
        return (None if gx is None else gx.reshape(lead + (K,)), out[:N * K].view(N, K), out[N * K:] if has_bias else None, None, None, None, None)


def linear(x, lin, act=0, p=0.0, seed=None, call=0):
    """lin(x) (act 0) or Dropout_p(ReLU(lin(x))) (act 1) for an nn.Linear on the HIP kernels; rows = all leading dimensions of x.
    `seed`: device int64 scalar of the step's dropout stream (needed when p > 0)."""
    if not x.is_cuda:
        raise RuntimeError("train_ops.linear runs on the MI355X only (HIP kernels, no CPU fallback)")
    if act and p > 0 and seed is None:
        raise RuntimeError("train_ops.linear: dropout needs the step's seed (train_forward sets it per batch)")
    return LinearFn.apply(x, lin.weight, lin.bias, int(act), float(p), seed if (act and p > 0) else None, int(call))


def mlp(seq, x, seed=None, call=0):
    """nn.Sequential of Linear / ReLU / Dropout / Tanh modules (the embeddings and heads of the score model, reference
    models/score_model.py:186-243) with every Linear -- and a ReLU / Dropout pair behind it, in either order -- on the fused kernels.
    `call`: base of the dropout stream indices of this module (each fused Linear takes the next one)."""
    mods = list(seq)
    i, k = 0, 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Linear):
            nxt = mods[i + 1:i + 3]
            kinds = tuple(type(q) for q in nxt)
            if kinds in ((torch.nn.ReLU, torch.nn.Dropout), (torch.nn.Dropout, torch.nn.ReLU)):
                drop = nxt[0] if isinstance(nxt[0], torch.nn.Dropout) else nxt[1]
                x = linear(x, m, act=1, p=float(drop.p) if drop.training else 0.0, seed=seed, call=call + k)
                i += 3
            elif kinds[:1] == (torch.nn.ReLU,):
                x = linear(x, m, act=1, p=0.0)
                i += 2
            else:
                x = linear(x, m)
                i += 1
            k += 1
        else:
            x = m(x)
            i += 1
    return x


class FcFirstStageFn(torch.autograd.Function):
    """hid = Dropout_p(ReLU(Linear_g(x)))  for every edge group g of a layer over ONE [E, 96] tensor of edge rows -- the first stage of the
    FCBlocks (reference models/layers.py:8-15) as one launch forward (cbd_fc1_forward) and three backward (cbd_fc1_backward: mask + input
    gradient; cbd_outer_accum_groups + cbd_partial_reduce: weight / bias gradients of all groups).  Round 3: one library GEMM per group, a
    clamp, torch's dropout; backward a masked scale, a threshold and GEMM + outer_accum + sum per group (csrc/train_fc.hip)."""

    @staticmethod
    def forward(ctx, x, seed, call, p, sizes, *wb):
        lib = _bind(load_library())
        x = x.contiguous().float()
        E, n = x.shape[0], len(sizes)
        assert x.shape[1] == KDIM and sum(sizes) == E and len(wb) == 2 * n
        ws = [w.contiguous().float() for w in wb[0::2]]
        bs = [b.contiguous().float() for b in wb[1::2]]
        hid = torch.empty(E, KDIM, device=x.device, dtype=torch.float32)
        ge = (C.c_int64 * n)(*[int(v) for v in sizes])
        wp_, bp_ = (C.c_void_p * n)(*[w.data_ptr() for w in ws]), (C.c_void_p * n)(*[b.data_ptr() for b in bs])
        _check(lib.cbd_fc1_forward(n, ge, _ptr(x), wp_, bp_, float(p), None if seed is None else _ptr(seed), int(call), _ptr(hid), _stream_handle()))
        ctx.save_for_backward(x, hid, *ws)
        ctx.meta = (tuple(int(v) for v in sizes), float(p))
        return hid

    @staticmethod
    def backward(ctx, ghid):
        x, hid, *ws = ctx.saved_tensors
        sizes, p = ctx.meta
        lib = _bind(load_library())
        n, E = len(sizes), x.shape[0]
        ghid = ghid.contiguous().float()
        gpre = torch.empty_like(x)
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        ge = (C.c_int64 * n)(*sizes)
        wp_ = (C.c_void_p * n)(*[w.data_ptr() for w in ws])
        _check(lib.cbd_fc1_backward(n, ge, _ptr(ghid), _ptr(hid), wp_, p, _ptr(gpre), None if gx is None else _ptr(gx), _stream_handle()))
        pf = int(lib.cbd_outer_accum_part_floats())
        parts = [max(1, min(1024, (ne + 63) // 64)) for ne in sizes]
        partial = torch.empty(sum(parts), pf, device=x.device, dtype=torch.float32)
        npart = (C.c_int32 * n)(*parts)
        _check(lib.cbd_outer_accum_groups(n, ge, npart, _ptr(gpre), _ptr(x), _ptr(partial), _stream_handle()))
        gwb = torch.empty(n, pf, device=x.device, dtype=torch.float32)
        oa = (C.c_void_p * n)(*[gwb[k].data_ptr() for k in range(n)])
        ob = (C.c_void_p * n)(*[gwb[k].data_ptr() + 4 * KDIM * KDIM for k in range(n)])
        _check(lib.cbd_partial_reduce(n, npart, pf, KDIM * KDIM, _ptr(partial), oa, ob, _stream_handle()))
        grads = []
        for k in range(n):
            grads += [gwb[k, :KDIM * KDIM].view(KDIM, KDIM), gwb[k, KDIM * KDIM:]]
        return (gx, None, None, None, None, *grads)


def fc_first_stage(x, sizes, fcs, seed=None, call=0):
    """Dropout(ReLU(fc[0](x))) of every edge group's FCBlock `fc` = nn.Sequential(Linear, ReLU, Dropout, Linear) over the rows of x; the
    groups share the dropout rate (they are built with the layer's).  `seed`: device int64 scalar of the step's dropout stream."""
    if not x.is_cuda:
        raise RuntimeError("fc_first_stage runs on the MI355X only (HIP kernels, no CPU fallback)")
    assert sum(sizes) == x.shape[0] and all(n > 0 for n in sizes) and len(sizes) == len(fcs)
    drop = fcs[0][2]
    p = float(drop.p) if drop.training else 0.0
    if p > 0 and seed is None:
        raise RuntimeError("fc_first_stage: dropout needs the step's seed (train_forward sets it per batch)")
    wb = [q for fc in fcs for q in (fc[0].weight, fc[0].bias)]
    return FcFirstStageFn.apply(x, seed if p > 0 else None, int(call), p, tuple(int(n) for n in sizes), *wb)


# ----------------------------------------------------------------------------- deterministic scatter / gather
# The reference's training graph scatters with atomics twice per layer: torch_scatter.scatter in TensorProductConvLayer.forward
# (models/tensor_layers.py:206) and autograd's index_add for every `node_attr[edge_index]` gather.  Atomic float adds make the step
# differ from run to run in the last bits.  Here every scatter is a segmented sum over edges grouped by target row (`cbd_segment_sum`,
# fixed order), so a training step is bitwise repeatable; the grouping (stable argsort + row pointers) is cached per index tensor.
@lru_cache(maxsize=4096)
def _csr_scratch_bytes(n: int, n_rows: int) -> int:
    need = C.c_size_t(0)
    _check(_bind(load_library()).cbd_csr_build(n, n_rows, None, None, None, None, 0, C.byref(need), None))
    return int(need.value)


class Csr:
    """Edges grouped by target row: perm = stable argsort of `index`, rowptr = first sorted position of every row -- one `cbd_csr_build`
    call (radix sort over the bits the row count needs + binary searches, all enqueued on the current stream).  torch.argsort(stable=True)
    synchronises the stream and torch.bincount reads its output size back: 33 + 34 pipeline flushes per training step in the profiles of
    round 3, which is what kept the host from running ahead of the GPU."""

    def __init__(self, index: torch.Tensor, n_rows: int):
        if not index.is_cuda:
            raise RuntimeError("edge grouping runs on the MI355X only (HIP kernels, no CPU fallback)")
        index = index.long().contiguous()
        lib = _bind(load_library())
        self.n_rows = int(n_rows)
        self.index = index
        n = int(index.shape[0])
        # sized for the next power of two (cacheable) AND for n itself: rocPRIM's temporary-storage size is not promised to be monotonic
        # in n across its size-dependent algorithm switch
        need = max(_csr_scratch_bytes(1 << max(n - 1, 0).bit_length(), self.n_rows), _csr_scratch_bytes(n, self.n_rows))
        scratch = torch.empty(need, dtype=torch.uint8, device=index.device)
        self.perm = torch.empty(n, dtype=torch.long, device=index.device)
        self.rowptr = torch.empty(self.n_rows + 1, dtype=torch.long, device=index.device)
        _check(lib.cbd_csr_build(n, self.n_rows, _ptr(index), _ptr(self.perm), _ptr(self.rowptr), _ptr(scratch), need, None,
                                 _stream_handle()))
        self._counts = None

    @classmethod
    def from_parts(cls, index, n_rows, perm, rowptr):
        c = cls.__new__(cls)
        c.n_rows, c.index, c.perm, c.rowptr, c._counts = int(n_rows), index, perm, rowptr, None
        return c

    @property
    def counts(self):
        if self._counts is None:
            self._counts = self.rowptr[1:] - self.rowptr[:-1]
        return self._counts


def csr_build_many(items, cache):
    """Groupings of several (index, n_rows) pairs with ONE radix sort (cbd_csr_build_batched; 7 launches instead of 6 per tensor),
    entered into `cache` under the keys csr_of() looks up.  Pairs already in the cache are skipped."""
    todo, seen = [], set()
    for index, n_rows in items:
        key = _csr_key(index, n_rows)
        if key not in cache and key not in seen:
            seen.add(key)
            todo.append((key, index, index.long().contiguous(), int(n_rows)))
    if not todo:
        return
    lib = _bind(load_library())
    dev = todo[0][2].device
    for lo in range(0, len(todo), 32):
        chunk = todo[lo:lo + 32]
        ns = len(chunk)
        ptrs = (C.c_void_p * ns)(*[t[2].data_ptr() for t in chunk])
        seg_n = (C.c_int64 * ns)(*[int(t[2].shape[0]) for t in chunk])
        seg_rows = (C.c_int64 * ns)(*[t[3] for t in chunk])
        n_tot, r_tot = sum(seg_n), sum(seg_rows) + ns
        need = C.c_size_t(0)
        _check(lib.cbd_csr_build_batched(ns, ptrs, seg_n, seg_rows, None, None, None, 0, C.byref(need), None))
        scratch = torch.empty(need.value, dtype=torch.uint8, device=dev)
        perm = torch.empty(n_tot, dtype=torch.long, device=dev)
        rowptr = torch.empty(r_tot, dtype=torch.long, device=dev)
        _check(lib.cbd_csr_build_batched(ns, ptrs, seg_n, seg_rows, _ptr(perm), _ptr(rowptr), _ptr(scratch), need.value, None, _stream_handle()))
        po = ro = 0
        for key, orig, idx, n_rows in chunk:
            n = int(idx.shape[0])
            c = Csr.from_parts(idx, n_rows, perm[po:po + n], rowptr[ro:ro + n_rows + 1])
            c.keep = orig
            cache[key] = c
            po += n
            ro += n_rows + 1


def _csr_key(index, n_rows):
    return (index.data_ptr(), int(index.shape[0]), int(index.stride(0)), str(index.dtype), int(n_rows), str(index.device), index._version)


_CSR_CACHE = {}          # groupings of the step being enqueued (train_forward.forward installs the prepared batch's own dict)


def csr_of(index: torch.Tensor, n_rows: int, cache=None) -> Csr:
    """`cache`: the dict to look up / fill instead of the current step's (a batch prepared ahead of its step fills its own)"""
    cache = _CSR_CACHE if cache is None else cache
    key = _csr_key(index, n_rows)
    c = cache.get(key)
    if c is None:
        if len(cache) > 256:
            cache.clear()
        c = Csr(index, n_rows)
        c.keep = index             # keeps the storage alive: the data_ptr in the key cannot be recycled while the entry exists
        cache[key] = c
    return c


def clear_csr_cache():
    use_csr_cache({})


def use_csr_cache(cache: dict):
    global _CSR_CACHE
    _CSR_CACHE = cache


def _segment_sum(vals: torch.Tensor, csr: Csr, mean: bool = False) -> torch.Tensor:
    if not vals.is_cuda:
        raise RuntimeError("segment_sum runs on the MI355X only (HIP kernel, no CPU fallback)")
    lib = _bind(load_library())
    v = vals.contiguous().float()
    v2 = v.reshape(v.shape[0], -1)
    out = torch.empty(csr.n_rows, v2.shape[1], device=v.device, dtype=torch.float32)
    if v2.shape[1] == 0 or csr.n_rows == 0:
        return out.reshape((csr.n_rows,) + tuple(v.shape[1:]))
    fn = lib.cbd_segment_mean if mean else lib.cbd_segment_sum
    _check(fn(csr.n_rows, v2.shape[1], _ptr(v2), _ptr(csr.perm), _ptr(csr.rowptr), _ptr(out), _stream_handle()))
    return out.reshape((csr.n_rows,) + tuple(v.shape[1:]))


class ScatterSumFn(torch.autograd.Function):
    """out[n] = sum_{e: index[e] = n} src[e]  (fixed order); backward = gather."""

    @staticmethod
    def forward(ctx, src, csr):
        ctx.csr = csr
        return _segment_sum(src, csr)

    @staticmethod
    def backward(ctx, g):
        return g.index_select(0, ctx.csr.index), None


class ScatterMeanFn(torch.autograd.Function):
    """torch_scatter.scatter(src, index, reduce='mean'): out[n] = sum_{e: index[e] = n} src[e] / max(count[n], 1), fixed order, the
    division inside the kernel; backward = gather of g / count."""

    @staticmethod
    def forward(ctx, src, csr):
        ctx.csr = csr
        return _segment_sum(src, csr, mean=True)

    @staticmethod
    def backward(ctx, g):
        csr = ctx.csr
        g = g.contiguous().float()
        width = int(np.prod(g.shape[1:]))
        if width % 4 == 0:      # one launch: gather + division by the clamped count
            out = torch.empty((csr.index.shape[0],) + tuple(g.shape[1:]), device=g.device, dtype=torch.float32)
            _check(_bind(load_library()).cbd_segment_mean_backward(int(csr.index.shape[0]), width, _ptr(g), _ptr(csr.index), _ptr(csr.rowptr),
                                                                   _ptr(out), _stream_handle()))
            return out, None
        if getattr(csr, "_inv_counts", None) is None:
            csr._inv_counts = 1.0 / csr.counts.clamp(min=1).to(torch.float32)
        scale = csr._inv_counts.reshape((-1,) + (1,) * (g.dim() - 1))
        return (g * scale).index_select(0, csr.index), None


class IrrepsBatchNormFn(torch.autograd.Function):
    """Train-mode e3nn BatchNorm (+ the layer's residual) on cbd_irreps_bn_forward / _backward: one launch each way.  `x` may be wider
    than the irreps layout (`dim` columns are read, e.g. the 80-float message rows): no slice copy before, no padding op behind."""

    @staticmethod
    def forward(ctx, x, dim, weight, bias, res, running_mean, running_var, fields, momentum, eps, exclude=None):
        """`exclude`: ((lo0, hi0), (lo1, hi1)) row ranges left out of the statistics (filler rows of a capacity-padded step) or None"""
        lib = _bind(load_library())
        ex = None
        if exclude is not None:
            (a0, b0), (a1, b1) = exclude
            ex = (C.c_int64 * 4)(int(a0), int(b0), int(a1), int(b1))
        ctx.ex = ex
        x = x.contiguous().float()
        n, ldx = x.shape
        nf = int(fields.shape[0])
        out = torch.empty(n, dim, device=x.device, dtype=torch.float32)
        stats = torch.empty(2, nf, device=x.device, dtype=torch.float32)
        r = None if res is None else res.contiguous().float()
        _check(lib.cbd_irreps_bn_forward(n, dim, ldx, nf, _ptr(fields), _ptr(x), None if r is None else _ptr(r),
                                         0 if r is None else int(r.shape[1]), _ptr(weight), _ptr(bias) if bias.numel() else None,
                                         _ptr(running_mean) if running_mean.numel() else None, _ptr(running_var), float(momentum),
                                         float(eps), _ptr(out), _ptr(stats[0]), _ptr(stats[1]), ex, _stream_handle()))
        ctx.save_for_backward(x, weight, stats, fields)
        ctx.res_dim = None if r is None else int(r.shape[1])
        ctx.n_bias = int(bias.numel())
        ctx.dim = dim
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _bind(load_library())
        x, weight, stats, fields = ctx.saved_tensors
        g = g.contiguous().float()
        n, ldx = x.shape
        nf = int(fields.shape[0])
        gx = torch.empty_like(x)
        gw = torch.empty(nf, device=x.device, dtype=torch.float32)
        gb = torch.empty(ctx.n_bias, device=x.device, dtype=torch.float32)
        _check(lib.cbd_irreps_bn_backward(n, ctx.dim, ldx, nf, _ptr(fields), _ptr(g), _ptr(x), _ptr(weight), _ptr(stats[0]), _ptr(stats[1]),
                                          _ptr(gx), _ptr(gw), _ptr(gb) if ctx.n_bias else None, ctx.ex, _stream_handle()))
        gres = None if ctx.res_dim is None else g[:, :ctx.res_dim]
        return gx, None, gw, gb, gres, None, None, None, None, None, None


class GatherFn(torch.autograd.Function):
    """x[index]; backward = segmented sum of the incoming gradient rows per source row (fixed order) instead of an atomic index_add."""

    @staticmethod
    def forward(ctx, x, csr):
        ctx.csr = csr
        return x.index_select(0, csr.index)

    @staticmethod
    def backward(ctx, g):
        return _segment_sum(g, ctx.csr), None


class EdgeCatFn(torch.autograd.Function):
    """[edge_attr | node[src][:32] | node[dst][:32]] (the FCBlock input of a layer) on cbd_edge_cat; backward on
    cbd_edge_cat_backward: both node gathers' gradients as ONE fixed-order pass into a zero-padded [N, D] tensor."""

    @staticmethod
    def forward(ctx, edge_attr, node, csr_src, csr_dst):
        lib = _bind(load_library())
        edge_attr, node = edge_attr.contiguous().float(), node.contiguous().float()
        E = edge_attr.shape[0]
        assert edge_attr.shape[1] == 32 and node.shape[1] >= 32 and csr_src.index.shape[0] == E and csr_dst.index.shape[0] == E
        out = torch.empty(E, KDIM, device=node.device, dtype=torch.float32)
        _check(lib.cbd_edge_cat(E, _ptr(edge_attr), _ptr(node), int(node.shape[1]), _ptr(csr_src.index), _ptr(csr_dst.index), _ptr(out),
                                _stream_handle()))
        ctx.csrs = (csr_src, csr_dst)
        ctx.node_shape = tuple(node.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _bind(load_library())
        cs, cd = ctx.csrs
        g = g.contiguous().float()
        N, D = ctx.node_shape
        g_node = None
        if ctx.needs_input_grad[1]:
            g_node = torch.empty(N, D, device=g.device, dtype=torch.float32)
            _check(lib.cbd_edge_cat_backward(N, D, _ptr(g), _ptr(cs.perm), _ptr(cs.rowptr), _ptr(cd.perm), _ptr(cd.rowptr), _ptr(g_node),
                                             _stream_handle()))
        return (g[:, :32] if ctx.needs_input_grad[0] else None), g_node, None, None


def edge_cat(edge_attr, node, src, dst):
    if not node.is_cuda:
        raise RuntimeError("edge_cat runs on the MI355X only (HIP kernels, no CPU fallback)")
    n = node.shape[0]
    return EdgeCatFn.apply(edge_attr, node, csr_of(src, n), csr_of(dst, n))


class GatherPadFn(torch.autograd.Function):
    """node[index] widened to the kernels' 80-float rows (zero padded) on cbd_gather_pad; backward = cbd_segment_sum_ld."""

    @staticmethod
    def forward(ctx, node, csr):
        lib = _bind(load_library())
        node = node.contiguous().float()
        E = int(csr.index.shape[0])
        out = torch.empty(E, NODE_STRIDE, device=node.device, dtype=torch.float32)
        _check(lib.cbd_gather_pad(E, int(node.shape[1]), NODE_STRIDE, _ptr(node), _ptr(csr.index), _ptr(out), _stream_handle()))
        ctx.csr = csr
        ctx.node_shape = tuple(node.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _bind(load_library())
        csr = ctx.csr
        g = g.contiguous().float()
        N, D = ctx.node_shape
        out = torch.empty(N, D, device=g.device, dtype=torch.float32)
        if N:
            _check(lib.cbd_segment_sum_ld(N, D, int(g.shape[1]), _ptr(g), _ptr(csr.perm), _ptr(csr.rowptr), _ptr(out), _stream_handle()))
        return out, None


def gather_pad(node, index):
    if not node.is_cuda:
        raise RuntimeError("gather_pad runs on the MI355X only (HIP kernels, no CPU fallback)")
    return GatherPadFn.apply(node, csr_of(index, node.shape[0]))


def scatter_mean(src, index, dim_size):
    if src.shape[0] == 0:
        return src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    return ScatterMeanFn.apply(src, csr_of(index, dim_size))


def scatter_sum(src, index, dim_size):
    if src.shape[0] == 0:
        return src.new_zeros((dim_size,) + tuple(src.shape[1:]))
    return ScatterSumFn.apply(src, csr_of(index, dim_size))


def gather_rows(x, index):
    if index.shape[0] == 0:
        return x.index_select(0, index)
    return GatherFn.apply(x, csr_of(index, x.shape[0]))


# ----------------------------------------------------------------------------- radius graphs
class RadiusQuery:
    """One batched radius search (cbd_radius_count now, cbd_radius_fill after the edge counts of ALL searches of the step have been read
    back together).  x, y: [n, 3] fp32; xptr [B + 1] node offsets of the graphs in x; ybatch [len(y)] graph of every query."""

    def __init__(self, x, y, r, xptr, ybatch, cap, drop_self=False, cutoff=None):
        if not x.is_cuda:
            raise RuntimeError("the radius search runs on the MI355X only (HIP kernels, no CPU fallback)")
        self.x, self.y = x.contiguous().float(), y.contiguous().float()
        self.cut = None if cutoff is None else cutoff.contiguous().float().reshape(-1)
        self.r2, self.cap, self.drop = float(np.float32(r) * np.float32(r)), int(cap), int(bool(drop_self))
        self.xptr, self.ybatch = xptr.contiguous(), ybatch.contiguous()
        assert self.xptr.dtype == torch.long and self.ybatch.dtype == torch.long
        self.ny = int(self.y.shape[0])
        self.counts = torch.empty(self.ny, dtype=torch.long, device=x.device)
        _check(_bind(load_library()).cbd_radius_count(self.ny, _ptr(self.x), _ptr(self.y), None if self.cut is None else _ptr(self.cut), self.r2,
                                                      _ptr(self.xptr), _ptr(self.ybatch), self.cap, self.drop, _ptr(self.counts), _stream_handle()))
        incl = torch.cumsum(self.counts, 0)
        self.offsets = incl - self.counts
        self.total = incl[-1] if self.ny else torch.zeros((), dtype=torch.long, device=x.device)

    def edges(self, n_edges: int) -> torch.Tensor:
        """[2, n_edges]: row 0 = query index, row 1 = point index (torch_cluster.radius order)"""
        out = torch.empty(2, int(n_edges), dtype=torch.long, device=self.x.device)
        if n_edges:
            _check(_bind(load_library()).cbd_radius_fill(self.ny, _ptr(self.x), _ptr(self.y), None if self.cut is None else _ptr(self.cut), self.r2,
                                                         _ptr(self.xptr), _ptr(self.ybatch), self.cap, self.drop, _ptr(self.offsets), _ptr(out[0]),
                                                         _ptr(out[1]), _stream_handle()))
        return out


def radius_queries(queries):
    """edge lists of several RadiusQuery objects with ONE device->host read-back (their edge counts)"""
    totals = torch.stack([q.total for q in queries]).tolist()
    return [q.edges(n) for q, n in zip(queries, totals)]


def edge_geometry(pos_a, pos_b, idx_a, idx_b, expansion=None, raw=False, unit=True):
    """(raw4 | None, unit4 | None, smear | None) of the edges (idx_a[e] -> idx_b[e]): vec = pos_b[idx_b] - pos_a[idx_a] (an index may be
    None: identity), one launch (cbd_edge_geometry).  `expansion`: a GaussianSmearing module (offset buffer, coeff)."""
    if not pos_a.is_cuda:
        raise RuntimeError("edge_geometry runs on the MI355X only (HIP kernel, no CPU fallback)")
    pos_a, pos_b = pos_a.contiguous().float(), pos_b.contiguous().float()
    E = int((idx_a if idx_a is not None else idx_b if idx_b is not None else pos_b).shape[0])
    dev = pos_a.device
    raw4 = torch.empty(E, 4, device=dev, dtype=torch.float32) if raw else None
    unit4 = torch.empty(E, 4, device=dev, dtype=torch.float32) if unit else None
    K, mu, coeff, smear = 0, None, 0.0, None
    if expansion is not None:
        mu = expansion.offset.contiguous().float()
        K, coeff = int(mu.shape[0]), float(expansion.coeff)
        smear = torch.empty(E, K, device=dev, dtype=torch.float32)
    p = lambda t: None if t is None else _ptr(t.contiguous())
    _check(_bind(load_library()).cbd_edge_geometry(E, _ptr(pos_a), _ptr(pos_b), p(idx_a), p(idx_b), K, p(mu), coeff, p(raw4), p(unit4), p(smear),
                                                   _stream_handle()))
    return raw4, unit4, smear


FUSED_LOSS = True       # False: the loss as torch ops (training.loss_from_targets; kept for the equivalence test)


class ScoreLossFn(torch.autograd.Function):
    """The denoising score-matching loss (apply_mean form) and its gradient with respect to the predictions on cbd_score_loss: one
    launch forward (which also writes the gradients), one small multiply per prediction backward.  Returns the reference's 11 values
    as one [11] tensor (reference utils/training.py:17-126)."""

    @staticmethod
    def forward(ctx, tr_pred, rot_pred, tor_pred, tr_score, tr_sigma, rot_score, rot_norm, tor_score, tor_norm2, weights, has_tor):
        lib = _bind(load_library())
        f = lambda t: None if t is None else t.contiguous().float()
        tr_pred, rot_pred, tor_pred, tr_score, tr_sigma, rot_score, rot_norm, tor_score, tor_norm2 = (
            f(t) for t in (tr_pred, rot_pred, tor_pred, tr_score, tr_sigma, rot_score, rot_norm, tor_score, tor_norm2))
        B = tr_pred.shape[0]
        T = int(tor_pred.numel()) if (has_tor and tor_pred is not None) else 0
        out = torch.empty(11, device=tr_pred.device, dtype=torch.float32)
        g_tr, g_rot = torch.empty_like(tr_pred), torch.empty_like(rot_pred)
        g_tor = torch.empty(T, device=tr_pred.device, dtype=torch.float32)
        p = lambda t: None if (t is None or t.numel() == 0) else _ptr(t)
        _check(lib.cbd_score_loss(B, T, int(bool(has_tor)), _ptr(tr_pred), _ptr(tr_score), _ptr(tr_sigma), _ptr(rot_pred), _ptr(rot_score),
                                  _ptr(rot_norm), p(tor_pred), p(tor_score), p(tor_norm2), float(weights[0]), float(weights[1]),
                                  float(weights[2]), _ptr(out), _ptr(g_tr), _ptr(g_rot), p(g_tor), _stream_handle()))
        ctx.save_for_backward(g_tr, g_rot, g_tor)
        ctx.tor_shape = None if tor_pred is None else tor_pred.shape
        return out

    @staticmethod
    def backward(ctx, gout):
        g_tr, g_rot, g_tor = ctx.saved_tensors
        k = gout[0]
        return (g_tr * k if ctx.needs_input_grad[0] else None, g_rot * k if ctx.needs_input_grad[1] else None,
                (g_tor * k).reshape(ctx.tor_shape) if (ctx.needs_input_grad[2] and ctx.tor_shape is not None) else None,
                None, None, None, None, None, None, None, None)


FUSED_HEADS = True      # False: the heads' tensor products as torch ops (train_forward.center_tensor_product / bond_tensor_product)


class CenterTpFn(torch.autograd.Function):
    """final_conv.tp with per-edge weights (reference models/score_model.py:245-255) on cbd_center_tp_forward / _backward."""

    @staticmethod
    def forward(ctx, x, vec, w):
        lib = _bind(load_library())
        x, vec, w = x.contiguous().float(), vec.contiguous().float(), w.contiguous().float()
        n = x.shape[0]
        out = torch.empty(n, 12, device=x.device, dtype=torch.float32)
        _check(lib.cbd_center_tp_forward(n, _ptr(x), x.shape[1], _ptr(vec), _ptr(w), _ptr(out), _stream_handle()))
        ctx.save_for_backward(x, vec, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, vec, w = ctx.saved_tensors
        lib = _bind(load_library())
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        _check(lib.cbd_center_tp_backward(x.shape[0], _ptr(x), x.shape[1], _ptr(vec), _ptr(w), _ptr(g.contiguous().float()), _ptr(gx), _ptr(gw),
                                          _stream_handle()))
        return gx, None, gw


class BondTpFn(torch.autograd.Function):
    """final_tp_tor + tor_bond_conv.tp (reference models/score_model.py:257-274, 431-441) on cbd_bond_tp_forward / _backward."""

    @staticmethod
    def forward(ctx, x, edge_vec, bond_vec, w):
        lib = _bind(load_library())
        x, edge_vec, bond_vec, w = (t.contiguous().float() for t in (x, edge_vec, bond_vec, w))
        n = x.shape[0]
        out = torch.empty(n, 64, device=x.device, dtype=torch.float32)
        _check(lib.cbd_bond_tp_forward(n, _ptr(x), x.shape[1], _ptr(edge_vec), _ptr(bond_vec), _ptr(w), _ptr(out), _stream_handle()))
        ctx.save_for_backward(x, edge_vec, bond_vec, w)
        return out

    @staticmethod
    def backward(ctx, g):
        x, edge_vec, bond_vec, w = ctx.saved_tensors
        lib = _bind(load_library())
        gx, gw = torch.empty_like(x), torch.empty_like(w)
        _check(lib.cbd_bond_tp_backward(x.shape[0], _ptr(x), x.shape[1], _ptr(edge_vec), _ptr(bond_vec), _ptr(w), _ptr(g.contiguous().float()),
                                        _ptr(gx), _ptr(gw), _stream_handle()))
        return gx, None, None, gw
