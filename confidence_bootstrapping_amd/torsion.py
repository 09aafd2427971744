"""Rotatable-bond masks of a ligand graph (reference utils/torsion.py:15-45, `get_transformation_mask`).

Same call, same result, no networkx / torch_geometric: `pyg_data['ligand', 'ligand'].edge_index` lists every bond twice in a row
(a -> b, then b -> a); bond k is rotatable when deleting it disconnects the graph and the SMALLEST connected component of what is
left has more than one atom; the mask row of the bond marks that component and sits on the direction whose SOURCE atom is outside
it (`edge_mask[2k + 1]` when `edges[2k, 0]` is inside, else `edge_mask[2k]`).  The reference has no bond-order test, and it takes
the smallest component of the WHOLE graph -- for a ligand of several fragments that can be a fragment the bond is not part of;
both are reproduced (tests/golden/g17_torsion_masks.npz comes from running the reference's function).

Ties between components of equal size go to the one found first by networkx's `connected_components`, i.e. the component whose
lowest-numbered atom is smallest (`sorted` is stable).
"""
from __future__ import annotations

import numpy as np


def _components(n, nbr, skip):
    """component label per node (labels in order of each component's lowest-numbered node), without the undirected bond `skip`"""
    comp = np.full(n, -1, dtype=np.int64)
    c = 0
    a0, b0 = skip
    for s in range(n):
        if comp[s] >= 0:
            continue
        comp[s] = c
        stack = [s]
        while stack:
            u = stack.pop()
            for v in nbr[u]:
                if comp[v] < 0 and not ((u == a0 and v == b0) or (u == b0 and v == a0)):
                    comp[v] = c
                    stack.append(v)
        c += 1
    return comp, c


def get_transformation_mask(pyg_data):
    """-> (mask_edges [E] bool, mask_rotate [R, N] bool) for the ligand of `pyg_data` (its ligand store and lig_bond edges are all the
    graph holds when the reference calls this, process_mols.py:650)."""
    ei = pyg_data["ligand", "ligand"].edge_index
    edges = (ei.cpu().numpy() if hasattr(ei, "cpu") else np.asarray(ei)).T.astype(np.int64)
    st = pyg_data["ligand"]
    n = int(st.x.shape[0]) if "x" in st else int(st.pos.shape[0])
    nbr = [set() for _ in range(n)]
    directed = set()
    for a, b in edges:
        directed.add((int(a), int(b)))
        if a != b:
            nbr[a].add(int(b))
            nbr[b].add(int(a))
    nbr = [sorted(s) for s in nbr]
    E = edges.shape[0]
    to_rotate = []
    for i in range(0, E, 2):
        assert edges[i, 0] == edges[i + 1, 1]
        a, b = int(edges[i, 0]), int(edges[i, 1])
        comp, nc = _components(n, nbr, (a, b))
        if nc > 1:
            sizes = np.bincount(comp, minlength=nc)
            small = int(np.argmin(sizes))                   # first minimum = lowest-numbered component among equals
            if sizes[small] > 1:
                side = np.nonzero(comp == small)[0]
                if comp[a] == small:
                    to_rotate += [None, side]
                else:
                    to_rotate += [side, None]
                continue
        to_rotate += [None, None]
    mask_edges = np.asarray([r is not None for r in to_rotate], dtype=bool)
    mask_rotate = np.zeros((int(mask_edges.sum()), n), dtype=bool)
    idx = 0
    for i in range(min(E, len(directed))):        # the reference bounds this loop by the DiGraph's edge count (duplicates collapse)
        if mask_edges[i]:
            mask_rotate[idx][to_rotate[i]] = True
            idx += 1
    return mask_edges, mask_rotate
