"""Library-GEMM forms of the fine-tuning ops that the hand-written kernels replaced (rounds 2-4) -- the EQUIVALENCE REFERENCES of
tests/test_gpu_train_op.py / test_gpu_train_graph.py and nothing else.  Not imported by the package: the product's training path
(confidence_bootstrapping_amd/train_forward.py) runs on the HIP kernels only.

  TensorProductFn / tensor_product   the per-FCBlock form of the tensor-product op: forward and g_x on cbd_tp_forward / cbd_tp_backward with a
                                     STORED g_w [E, wp]; g_h and dW2p as library GEMMs on it (torch.mm).  The product form is
                                     train_ops.TensorProductHubFn (g_w never stored, cbd_tp_backward_gh / _dw_groups).
  weight_grad_split_k                dW = g_w^T h as a batched split-K library GEMM (round 3).
  FirstLinearFn / GroupedFirstLinearFn / first_stage_reference
                                     the FCBlocks' first stage as library GEMMs + torch ReLU / Dropout (round 3); product form:
                                     train_ops.FcFirstStageFn.
  linear_reference                   nn.Linear (+ ReLU + Dropout) through torch; product form: train_ops.LinearFn."""
import ctypes as C

import torch

from confidence_bootstrapping_amd.engine import load_library
from confidence_bootstrapping_amd.train_ops import (KDIM, NODE_STRIDE, TIMER, _bind, _check, _ptr, _stream_handle, stream_map)


class TensorProductFn(torch.autograd.Function):
    """msg[E, 80] = FasterTensorProduct(xrow[E, 80], [1, sqrt3 vec[E, :3]], W2_g h + b2_g): the edges are the concatenation of
    `group_edges[g]` edges per group g, group g using (W2_g, b2_g) inside `streams[g]`; all groups run in one launch."""

    @staticmethod
    def forward(ctx, xrow, vec4, h, in_level, out_level, group_edges, *streams):
        if not xrow.is_cuda:
            raise RuntimeError("TensorProductFn runs on the MI355X HIP kernels only (no CPU fallback)")
        lib = _bind(load_library())
        xrow, vec4, h = xrow.contiguous().float(), vec4.contiguous().float(), h.contiguous().float()
        streams = [st.contiguous().float() for st in streams]
        E = xrow.shape[0]
        assert xrow.shape == (E, NODE_STRIDE) and vec4.shape == (E, 4) and h.shape == (E, KDIM)
        assert len(group_edges) == len(streams) and sum(group_edges) == E
        n = len(streams)
        ge = (C.c_int64 * n)(*[int(x) for x in group_edges])
        ws = (C.c_void_p * n)(*[st.data_ptr() for st in streams])
        msg = torch.empty(E, NODE_STRIDE, device=xrow.device, dtype=torch.float32)
        TIMER.wrap("fwd", in_level, out_level, E, lambda: _check(lib.cbd_tp_forward(
            in_level, out_level, n, ge, _ptr(xrow), _ptr(vec4), _ptr(h), ws, _ptr(msg), _stream_handle())))
        ctx.save_for_backward(xrow, vec4, h, *streams)
        ctx.meta = (in_level, out_level, list(group_edges))
        return msg

    @staticmethod
    def backward(ctx, gmsg):
        xrow, vec4, h, *streams = ctx.saved_tensors
        in_level, out_level, group_edges = ctx.meta
        lib = _bind(load_library())
        sm = stream_map(in_level, out_level)
        d = sm.on(xrow.device)
        E, n = xrow.shape[0], len(streams)
        gmsg = gmsg.contiguous().float()
        gx = torch.empty_like(xrow)
        gw = torch.empty(E, sm.wp, device=xrow.device, dtype=torch.float32)
        ge = (C.c_int64 * n)(*[int(x) for x in group_edges])
        ws = (C.c_void_p * n)(*[st.data_ptr() for st in streams])
        TIMER.wrap("bwd", in_level, out_level, E, lambda: _check(lib.cbd_tp_backward(
            in_level, out_level, n, ge, _ptr(xrow), _ptr(vec4), _ptr(h), ws, _ptr(gmsg), _ptr(gx), _ptr(gw), _stream_handle())))
        gh = torch.empty_like(h) if ctx.needs_input_grad[2] else None
        gstreams, lo = [], 0
        for g, (ne, stream) in enumerate(zip(group_edges, streams)):
            hi = lo + ne
            gwg = gw[lo:hi]
            if gh is not None and ne:
                torch.mm(gwg, stream[d["w2p"]].view(sm.wp, KDIM), out=gh[lo:hi])
            gs = None
            if ctx.needs_input_grad[6 + g]:
                gs = torch.zeros_like(stream)
                if ne:
                    gs[d["w2p"]] = (gwg.t() @ h[lo:hi]).reshape(-1)
                    gs[d["b2p"]] = gwg.sum(0)
            gstreams.append(gs)
            lo = hi
        return (gx if ctx.needs_input_grad[0] else None), None, gh, None, None, None, *gstreams


def tensor_product(xrow, vec4, h, streams, in_level, out_level, group_edges=None):
    """`streams`: one stream tensor or a list (one per edge group, with `group_edges` = edges per group)."""
    if torch.is_tensor(streams):
        streams, group_edges = [streams], [xrow.shape[0]]
    return TensorProductFn.apply(xrow, vec4, h, in_level, out_level, tuple(int(x) for x in group_edges), *streams)



def weight_grad_split_k(gw, h, dw):
    """dw = gw^T h  ([Wp, E] x [E, 96]) for E up to 10^5..10^6 edge rows.  The library runs this shape WITHOUT split-K: 19 x 3 macro-tiles
    = 57 workgroups on 256 CUs, 56-69 TFLOP/s.  Cut into 16 edge chunks as one batched GEMM (912 workgroups) plus a fixed-order sum of the
    partial products it reaches 112-117 TFLOP/s (tools/micro-benchmarks of round 3: E = 50 000: 0.275 -> 0.155 ms, E = 74 000: 0.378 ->
    0.231 ms; with only 3-4 chunks it is SLOWER than the plain call)."""
    E = gw.shape[0]
    S = 16 if E >= 16384 else 8 if E >= 4096 else 1
    if S == 1:
        torch.mm(gw.t(), h, out=dw)
        return
    chunk = E // S
    main = chunk * S
    part = torch.bmm(gw[:main].view(S, chunk, -1).transpose(1, 2), h[:main].view(S, chunk, -1))
    torch.sum(part, 0, out=dw)
    if main < E:
        dw.addmm_(gw[main:].t(), h[main:])



class FirstLinearFn(torch.autograd.Function):
    """y = x W^T + b for the FCBlock's first Linear (96 -> 96) with the weight / bias gradient on `cbd_outer_accum`: the reduction
    over 10^5..10^6 edges into a 96 x 96 matrix that library GEMMs run at ~10 TFLOP/s (csrc/tp_train.hip::outer_accum_kernel)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        gx = g @ weight if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            lib = _bind(load_library())
            E = x.shape[0]
            g32, x32 = g.contiguous().float(), x.contiguous().float()
            n_parts = max(1, min(1024, (E + 63) // 64))
            pf = int(lib.cbd_outer_accum_part_floats())
            parts = torch.empty(n_parts, pf, device=x.device, dtype=torch.float32)
            _check(lib.cbd_outer_accum(E, _ptr(g32), _ptr(x32), n_parts, _ptr(parts), _stream_handle()))
            tot = parts.sum(0)
            gw, gb = tot[:KDIM * KDIM].view(KDIM, KDIM), tot[KDIM * KDIM:]
        return gx, gw, gb


class GroupedFirstLinearFn(torch.autograd.Function):
    """The first Linear of every edge group's FCBlock over ONE [E, 96] tensor of edge rows: group g owns the contiguous rows
    [lo_g, hi_g) and its own (W_g, b_g).  Results and input gradients are written into slices of one buffer -- slicing the input per
    group in autograd instead costs a zero-filled [E, 96] tensor, a copy and an accumulation per group in the backward pass
    (SliceBackward0: 116 launches per step in the profile of round 3)."""

    @staticmethod
    def forward(ctx, x, sizes, *wb):
        x = x.contiguous().float()
        out = torch.empty(x.shape[0], wb[0].shape[0], device=x.device, dtype=torch.float32)
        lo = 0
        for g, ne in enumerate(sizes):
            torch.addmm(wb[2 * g + 1], x[lo:lo + ne], wb[2 * g].t(), out=out[lo:lo + ne])
            lo += ne
        ctx.save_for_backward(x, *wb[0::2])
        ctx.sizes = sizes
        return out

    @staticmethod
    def backward(ctx, g):
        x, *ws = ctx.saved_tensors
        lib = _bind(load_library())
        g = g.contiguous().float()
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        pf = int(lib.cbd_outer_accum_part_floats())
        grads, lo = [], 0
        for k, ne in enumerate(ctx.sizes):
            gg, xg = g[lo:lo + ne], x[lo:lo + ne]
            if gx is not None:
                torch.mm(gg, ws[k], out=gx[lo:lo + ne])
            n_parts = max(1, min(1024, (ne + 63) // 64))
            parts = torch.empty(n_parts, pf, device=x.device, dtype=torch.float32)
            _check(lib.cbd_outer_accum(ne, _ptr(gg), _ptr(xg), n_parts, _ptr(parts), _stream_handle()))
            tot = parts.sum(0)
            grads += [tot[:KDIM * KDIM].view(KDIM, KDIM), tot[KDIM * KDIM:]]
            lo += ne
        return (gx, None, *grads)


def grouped_first_linear(x, sizes, linears):
    """[linear_g(x[lo_g:hi_g])] concatenated, for nn.Linear(96, 96) modules and group sizes that add up to x.shape[0] (all > 0)."""
    if not x.is_cuda:
        raise RuntimeError("grouped_first_linear runs on the MI355X only (HIP weight-gradient kernel, no CPU fallback)")
    assert sum(sizes) == x.shape[0] and all(n > 0 for n in sizes) and len(sizes) == len(linears)
    wb = [p for lin in linears for p in (lin.weight, lin.bias)]
    return GroupedFirstLinearFn.apply(x, tuple(int(n) for n in sizes), *wb)


def first_linear(x, linear):
    """`linear(x)` for an nn.Linear(96, 96) on [E, 96] edge rows (E >= 1), HIP weight-gradient reduction."""
    if not x.is_cuda:
        raise RuntimeError("first_linear runs on the MI355X only (HIP weight-gradient kernel, no CPU fallback)")
    if x.shape[0] == 0:
        return x.new_zeros(0, linear.weight.shape[0]) + 0 * linear.bias
    return FirstLinearFn.apply(x, linear.weight, linear.bias)


def first_stage_reference(x, sizes, fcs):
    """Dropout(ReLU(fc[0](x))) per edge group: library GEMM per group + torch ReLU / Dropout (the round-3 form of train_ops.fc_first_stage)"""
    drop = fcs[0][2]
    pre = grouped_first_linear(x, sizes, [fc[0] for fc in fcs])
    return torch.nn.functional.dropout(torch.relu(pre), p=drop.p, training=drop.training)


def linear_reference(x, lin, act=0, p=0.0):
    y = torch.nn.functional.linear(x, lin.weight, lin.bias)
    return torch.nn.functional.dropout(torch.relu(y), p=p, training=p > 0) if act else y
