"""CPU restatement of the torch_cluster / torch_scatter ops the reference's hot path calls.

TEST INFRASTRUCTURE ONLY (oracle) -- see oracle/e3nn_ref.py header for the import rule.

torch-cluster==1.6.0 and torch-scatter==2.0.9 (environment.yml:177,179) are un-vendored and absent
here: PARITY UNPINNED at this boundary.  Published semantics restated:
  radius(x, y, r, batch_x, batch_y, max_num_neighbors)  -> [2, E]; row0 = index into y, row1 = index into x,
      for every y the x's of the same example with |x - y|^2 < r^2 (GPU back-end: strict '<', x scanned in
      index order, stop after max_num_neighbors).                     call sites score_model.py:568-573,655
  radius_graph(x, r, batch, loop=False, max_num_neighbors=32, flow='source_to_target')
      = radius(x, x, r, batch, batch, cap [+1 if not loop]) with self pairs dropped and rows flipped to
      [neighbour; centre].                                            call site  score_model.py:502
  scatter(src, index, dim=0, dim_size, reduce='mean') : sum / max(count, 1).   tensor_layers.py:206
"""
from __future__ import annotations

import torch


def radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32):
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    # squared distance accumulated coordinate by coordinate in fp32, like the CUDA kernel's scalar loop
    d2 = torch.zeros(y.shape[0], x.shape[0], dtype=x.dtype)
    for k in range(x.shape[1]):
        diff = x[None, :, k] - y[:, None, k]
        d2 = d2 + diff * diff
    ok = (d2 < r * r) & (batch_y[:, None] == batch_x[None, :])
    # cap: keep the first max_num_neighbors x's in index order for each y
    rank = torch.cumsum(ok.to(torch.long), dim=1)
    ok = ok & (rank <= max_num_neighbors)
    row, col = torch.nonzero(ok, as_tuple=True)  # row-major => sorted by y then x
    return torch.stack([row, col], dim=0)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32):
    ei = radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    row, col = ei[0], ei[1]  # row = centre (y idx), col = neighbour (x idx)
    if not loop:
        keep = row != col
        row, col = row[keep], col[keep]
    # flow='source_to_target': edge_index = [neighbour; centre]
    return torch.stack([col, row], dim=0)


def scatter(src, index, dim=0, dim_size=None, reduce="mean"):
    assert dim == 0
    n = int(dim_size) if dim_size is not None else int(index.max()) + 1
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype)
    out.index_add_(0, index, src)
    if reduce in ("sum", "add"):
        return out
    assert reduce == "mean"
    cnt = torch.zeros(n, dtype=src.dtype)
    cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype))
    cnt = cnt.clamp(min=1)
    return out / cnt.reshape((n,) + (1,) * (src.dim() - 1))


def scatter_mean(src, index, dim=0, dim_size=None):
    return scatter(src, index, dim=dim, dim_size=dim_size, reduce="mean")
