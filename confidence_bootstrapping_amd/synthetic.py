"""Deterministic synthetic protein-ligand complexes in the reference's graph schema.

DockGen / PDBBind data and rdkit are unavailable (no network), so benchmarks and tests run on seeded
synthetic complexes whose tensor layout is exactly what `datasets/process_mols.py:448-489,567-589` and
`datasets/moad.py:202-212` produce:

  ['ligand'].x [Nl,16] int64          categorical atom features within lig_feature_dims
  ['ligand'].pos [Nl,3] f32           heavy-atom conformer
  ['ligand'].edge_mask [2*bonds] bool rotatable-bond mask (one direction per rotatable bond)
  ['ligand'].mask_rotate [R,Nl] bool  atoms moved by each torsion (side containing edge[1])
  ['ligand','ligand'].edge_index [2, 2*bonds] (each bond twice, consecutive), .edge_attr [.,4] one-hot
  ['receptor'].x [Nr, 1+1280] f32     col 0 residue type, cols 1.. language-model embedding
  ['receptor'].pos [Nr,3] f32         C-alpha trace centred on its centroid
  ['receptor','receptor'].edge_index [2, 24*Nr]  kNN graph, row0 = neighbour, row1 = centre
and, with `add_atoms()` (the all-atom schema the confidence model reads, process_mols.py:490-526):
  ['atom'].x [Na,4] f32               categorical: residue type, atomic number idx, atom_type_2, atom_type_3
  ['atom'].pos [Na,3] f32             heavy atoms, stored residue by residue
  ['atom','atom_contact','atom'].edge_index [2, 8*Na]   kNN-8 graph, row0 = neighbour, row1 = centre
  ['atom','atom_rec_contact','receptor'].edge_index [2, Na]  (atom, its residue)
  ['receptor'].side_chain_vecs [Nr,4,3] (carried through crop_beyond, not read by the model)
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.spatial import cKDTree

from .hetero import HeteroData

LIG_FEATURE_DIMS = [119, 4, 12, 12, 8, 10, 6, 6, 2, 8, 2, 2, 2, 2, 2, 2]  # datasets/process_mols.py:95-112
REC_RESIDUE_TYPES = 38                                                     # datasets/process_mols.py:121-123
LM_DIM = 1280

# named workloads of BASELINE.json / SURVEY.md section 8
WORKLOADS = {
    "tiny":   dict(Nl=12, Nr=40, R=2, knn=8),
    "c2_dockgen_median": dict(Nl=28, Nr=384, R=6, knn=24),
    "c4_large_pocket":   dict(Nl=64, Nr=1024, R=16, knn=24),
}


def _receptor_trace(rng, n, globular=False):
    """Compact self-avoiding C-alpha walk: 3.8 A steps inside a sphere.  Default (the test complexes and their goldens): a loose
    coil, sphere sized for Rg ~ 2.2 n^0.38 A (~ 320 A^3 per residue).  globular=True: the sphere of a folded protein, 135 A^3 per
    residue (384 residues -> radius 23 A, ~ 27 C-alphas within 10 A), which gives the cross-edge counts of SURVEY.md 8:
    ~ 125-150 residues within 20 A of a pocket atom."""
    rg = 2.2 * n ** 0.38
    rs = (n * 135.0 * 3.0 / (4.0 * np.pi)) ** (1.0 / 3.0) if globular else 1.15 * rg / np.sqrt(0.6)
    pts = np.zeros((n, 3))
    i, stuck = 1, 0
    while i < n:
        ok = False
        for _ in range(60):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            p = pts[i - 1] + 3.8 * d
            if np.linalg.norm(p) > rs:
                continue
            if i > 1 and np.min(np.linalg.norm(pts[:i - 1] - p, axis=1)) < (3.7 if globular else 3.6):
                continue
            pts[i], ok = p, True
            break
        if ok:
            i, stuck = i + 1, 0
        else:  # back-track a few residues
            stuck += 1
            i = max(1, i - min(8, 1 + stuck))
            if stuck > 200:
                rs *= 1.05
                stuck = 0
    return pts - pts.mean(0)


def _ring(rng):
    a = np.arange(6) * np.pi / 3
    return np.stack([1.4 * np.cos(a), 1.4 * np.sin(a), np.zeros(6)], 1)


def _rand_rot(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _ligand(rng, n):
    """Rings and chain atoms joined by single bonds.  Returns pos [n,3], bonds [(i,j)], bond_type."""
    pos, bonds, btype, deg = [], [], [], []

    def add_fragment(frag, internal, anchor):
        base = len(pos)
        for p in frag:
            pos.append(p)
            deg.append(0)
        for (a, b, t) in internal:
            bonds.append((base + a, base + b))
            btype.append(t)
            deg[base + a] += 1
            deg[base + b] += 1
        if anchor is not None:
            bonds.append((anchor, base))
            btype.append(0)
            deg[anchor] += 1
            deg[base] += 1

    while len(pos) < n:
        left = n - len(pos)
        use_ring = left >= 6 and rng.random() < 0.45
        if use_ring:
            frag = _ring(rng) @ _rand_rot(rng).T
            internal = [(k, (k + 1) % 6, 3) for k in range(6)]
        else:
            frag = np.zeros((1, 3))
            internal = []
        if not pos:
            add_fragment(list(frag), internal, None)
            continue
        P = np.asarray(pos)
        cands = [k for k in range(len(pos)) if deg[k] < 3]
        placed = False
        for _ in range(200):
            a = int(rng.choice(cands))
            out = P[a] - P.mean(0)
            d = out / (np.linalg.norm(out) + 1e-9) + 0.9 * rng.normal(size=3)
            d /= np.linalg.norm(d)
            f = frag - frag[0]
            if use_ring:  # point the ring away from the anchor
                f = f @ _rand_rot(rng).T
                c = f.mean(0)
                if np.dot(c, d) < 0.5 * np.linalg.norm(c):
                    continue
            f = f + P[a] + 1.5 * d
            dist = np.linalg.norm(P[None, :, :] - f[:, None, :], axis=-1)
            dist[0, a] = 10.0
            if dist.min() < 2.1:
                continue
            add_fragment(list(f), internal, a)
            placed = True
            break
        if not placed:
            raise RuntimeError("ligand growth failed")
    return np.asarray(pos[:n]), bonds, btype


def _components_without(n, bonds, skip):
    adj = [[] for _ in range(n)]
    for k, (a, b) in enumerate(bonds):
        if k == skip:
            continue
        adj[a].append(b)
        adj[b].append(a)
    comp = -np.ones(n, dtype=int)
    c = 0
    for s in range(n):
        if comp[s] >= 0:
            continue
        stack = [s]
        comp[s] = c
        while stack:
            u = stack.pop()
            for v in adj[u]:
                if comp[v] < 0:
                    comp[v] = c
                    stack.append(v)
        c += 1
    return comp, c


def make_complex(Nl=28, Nr=384, R=6, knn=24, seed=1234, name=None, globular=False, pocket_depth=0.85) -> HeteroData:
    """One synthetic complex.  Raises if the requested number of rotatable bonds cannot be realised.
    globular / pocket_depth: receptor at folded-protein density with the crystal pose at `pocket_depth` x the distance of the
    outermost residue from the centroid (the benchmark geometry, see _receptor_trace); the defaults are the test geometry."""
    for attempt in range(200):
        rng = np.random.default_rng(seed + 7919 * attempt)
        try:
            lpos, bonds, btype = _ligand(rng, Nl)
        except RuntimeError:
            continue
        # rotatable candidates: bridges with >= 2 atoms on both sides (utils/torsion.py:15-45 semantics)
        cand = []
        for k in range(len(bonds)):
            comp, nc = _components_without(Nl, bonds, k)
            if nc == 2:
                sizes = np.bincount(comp)
                if sizes.min() >= 2:
                    cand.append((k, comp))
        if len(cand) < R:
            continue
        pick = sorted(rng.choice(len(cand), size=R, replace=False).tolist()) if R > 0 else []
        break
    else:
        raise RuntimeError(f"could not build a ligand with Nl={Nl}, R={R}")

    nb = len(bonds)
    edge_index = np.zeros((2, 2 * nb), dtype=np.int64)
    edge_attr = np.zeros((2 * nb, 4), dtype=np.float32)
    for k, (a, b) in enumerate(bonds):
        edge_index[:, 2 * k] = (a, b)
        edge_index[:, 2 * k + 1] = (b, a)
        edge_attr[2 * k, btype[k]] = 1
        edge_attr[2 * k + 1, btype[k]] = 1
    edge_mask = np.zeros(2 * nb, dtype=bool)
    mask_rotate = np.zeros((R, Nl), dtype=bool)
    rows = []
    for ci in pick:
        k, comp = cand[ci]
        a, b = bonds[k]
        sizes = np.bincount(comp)
        small = int(np.argmin(sizes))          # smaller side rotates (ties -> component 0)
        side = comp == small
        # the masked direction is the one whose SECOND endpoint lies in the rotated side
        if side[a]:
            rows.append((2 * k + 1, side))     # direction (b, a)
        else:
            rows.append((2 * k, side))         # direction (a, b)
    rows.sort(key=lambda r: r[0])
    for i, (e, side) in enumerate(rows):
        edge_mask[e] = True
        mask_rotate[i] = side

    rpos = _receptor_trace(rng, Nr, globular)
    tree = cKDTree(rpos)
    _, nbr = tree.query(rpos, k=min(knn, Nr - 1) + 1)
    nbr = nbr[:, 1:]
    centre = np.repeat(np.arange(Nr), nbr.shape[1])
    rec_edge_index = np.stack([nbr.reshape(-1), centre]).astype(np.int64)  # [neighbour; centre]
    rec_x = np.zeros((Nr, 1 + LM_DIM), dtype=np.float32)
    rec_x[:, 0] = rng.integers(0, 20, size=Nr)
    rec_x[:, 1:] = rng.normal(0, 0.5, size=(Nr, LM_DIM))

    lig_x = np.stack([rng.integers(0, d, size=Nl) for d in LIG_FEATURE_DIMS], 1).astype(np.int64)
    # put the crystal pose on the protein surface
    surf = rpos[np.argmax(np.linalg.norm(rpos, axis=1))]
    lpos = lpos - lpos.mean(0) + surf * pocket_depth

    d = HeteroData()
    d["ligand"].x = torch.from_numpy(lig_x)
    d["ligand"].pos = torch.from_numpy(lpos.astype(np.float32))
    d["ligand"].edge_mask = torch.from_numpy(edge_mask)
    d["ligand"].mask_rotate = mask_rotate
    d["ligand", "ligand"].edge_index = torch.from_numpy(edge_index)
    d["ligand", "ligand"].edge_attr = torch.from_numpy(edge_attr)
    d["receptor"].x = torch.from_numpy(rec_x)
    d["receptor"].pos = torch.from_numpy(rpos.astype(np.float32))
    d["receptor", "receptor"].edge_index = torch.from_numpy(rec_edge_index)
    d.original_center = torch.zeros(1, 3)
    d.name = name or f"synthetic_Nl{Nl}_Nr{Nr}_R{R}_s{seed}"
    return d


REC_ATOM_FEATURE_DIMS = [38, 119, 23, 38]   # datasets/process_mols.py:114-119


def add_atoms(d: HeteroData, seed=1234, atom_knn=8, mean_side_atoms=4.0) -> HeteroData:
    """Adds the all-atom receptor stores to a complex made by make_complex (own RNG stream, so the coarse-grained
    part is unchanged).  Every residue gets backbone N, CA, C, O plus 0..10 side-chain atoms placed around its C-alpha
    with >= 1.2 A separation; atoms are stored residue by residue like the reference's featuriser."""
    rng = np.random.default_rng([seed, 77])
    rpos = d["receptor"].pos.numpy().astype(np.float64)
    Nr = len(rpos)
    res_type = d["receptor"].x[:, 0].numpy().astype(np.int64)
    pos, feat, res_of = [], [], []
    for r in range(Nr):
        n_side = int(min(10, rng.poisson(mean_side_atoms)))
        mine = [rpos[r]]                                        # CA sits on the trace point
        for k in range(3 + n_side):
            for _ in range(100):
                dirn = rng.normal(size=3)
                dirn /= np.linalg.norm(dirn)
                base = mine[0] if k < 3 else mine[int(rng.integers(0, len(mine)))]
                p = base + 1.5 * dirn
                if np.min(np.linalg.norm(np.asarray(mine) - p, axis=1)) >= 1.2:
                    mine.append(p)
                    break
        order = [1, 0] + list(range(2, len(mine)))              # N, CA, C, O, side chain...
        for j, k in enumerate(order):
            pos.append(mine[k])
            if j < 4:
                f = [res_type[r], (6, 5, 5, 7)[j], (8, 1, 0, 13)[j], (17, 1, 0, 26)[j]]
            else:
                f = [res_type[r], int(rng.choice([5, 5, 5, 6, 7, 15])), int(rng.integers(0, 23)), int(rng.integers(0, 38))]
            feat.append(f)
            res_of.append(r)
    pos = np.asarray(pos)
    Na = len(pos)
    tree = cKDTree(pos)
    _, nbr = tree.query(pos, k=atom_knn + 1)
    nbr = nbr[:, 1:]
    centre = np.repeat(np.arange(Na), atom_knn)
    d["atom"].x = torch.from_numpy(np.asarray(feat, dtype=np.float32))
    d["atom"].pos = torch.from_numpy(pos.astype(np.float32))
    d["atom", "atom_contact", "atom"].edge_index = torch.from_numpy(np.stack([nbr.reshape(-1), centre]).astype(np.int64))
    d["atom", "atom_rec_contact", "receptor"].edge_index = torch.from_numpy(
        np.stack([np.arange(Na), np.asarray(res_of)]).astype(np.int64))
    d["receptor"].side_chain_vecs = torch.from_numpy(rng.normal(size=(Nr, 4, 3)).astype(np.float32))
    return d


def make_workload(workload: str, seed=1234, all_atoms=False, **geometry) -> HeteroData:
    d = make_complex(seed=seed, name=workload, **WORKLOADS[workload], **geometry)
    return add_atoms(d, seed=seed) if all_atoms else d


# ---- the benchmark workload of BASELINE.json configs[1] (bench.py, tests/test_gpu_configs.py) ----------------------------------
BENCH_GEOMETRY = dict(globular=True, pocket_depth=0.7)   # folded-protein density, pocket at 0.7 of the surface radius
TR_HEAD_SCALE = 0.02


def scale_tr_head(model, scale=TR_HEAD_SCALE):
    """Random-init weights know nothing about the pocket; scaling the last layer of the translation head keeps its random drift
    below 1 A over a trajectory, so that the poses stay on the path the pre-drawn noise prescribes (ideal_path_noise)."""
    with torch.no_grad():
        model.tr_final_layer[3].weight.mul_(scale)
        model.tr_final_layer[3].bias.mul_(scale)
    return model


def ideal_path_noise(pos0, pocket, sched, margs):
    """Pre-drawn translation noise that carries the centroid of pose b along pocket + sigma_tr(t_i) eps_b -- the probability-flow path
    of the reverse process for a point-mass data distribution, i.e. what a trained score model produces:
    z[i, b] = (sigma(t_{i+1}) - sigma(t_i)) eps_b / (g_i sqrt(dt_i)), eps_b fixed by the initial pose, sigma(t_S) = sigma_min."""
    S = len(sched)
    lo, hi = margs.tr_sigma_min, margs.tr_sigma_max
    sig = np.array([lo ** (1 - t) * hi ** t for t in list(sched) + [0.0]])
    eps = (pos0.mean(1) - pocket) / sig[0]                                   # [B, 3]
    z = torch.zeros(S, pos0.shape[0], 3)
    for i in range(S):
        dt = sched[i] - sched[i + 1] if i < S - 1 else sched[i]
        g = sig[i] * np.sqrt(2 * np.log(hi / lo))
        z[i] = (sig[i + 1] - sig[i]) / (g * np.sqrt(dt)) * eps
    return z


def ideal_path_inputs(cplx, margs, sched, samples, seed):
    """Initial poses (the reference's randomize_position, re-centred on the pocket) and the noise of one complex of the benchmark
    workload: (pos0 [B,Nl,3], z_tr [S,B,3], z_rot [S,B,3], z_tor [S,B*R]) CPU tensors, deterministic in `seed`."""
    import copy
    from .hetero import Batch
    from .sampling import randomize_position
    state = (np.random.get_state(), torch.random.get_rng_state())
    try:
        torch.manual_seed(seed)
        np.random.seed(seed)
        dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(samples)]
        randomize_position(dl, False, False, margs.tr_sigma_max)             # prior centred on the receptor centroid (no pocket knowledge)
        p0 = torch.stack([d["ligand"].pos for d in dl])
        pocket = cplx["ligand"].pos.mean(0)
        p0 = p0 + (pocket - cplx["receptor"].pos.mean(0))                    # the same prior, centred on the pocket
        S, R = len(sched), int(cplx["ligand"].edge_mask.sum())
        torch.randn(S, samples, 3)                                           # (the draw bench.py makes for the free-running variant)
        z_tr = ideal_path_noise(p0, pocket, sched, margs)
        return p0.contiguous(), z_tr, torch.randn(S, samples, 3), torch.randn(S, samples * R)
    finally:
        np.random.set_state(state[0])
        torch.random.set_rng_state(state[1])


# ---- BASELINE.json configs[2]: a heterogeneous set of complexes ---------------------------------
def complex_set_sizes(n, seed=7):
    """(Nl, Nr, R) of n complexes: log-normal around the DockGen-median complex (Nl 28, Nr 384), clipped to Nl in [6, 80] and
    Nr in [64, 1500]; R ~ Nl / 5 rotatable bonds, at most 14."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        nl = int(np.clip(round(28 * np.exp(0.42 * rng.normal())), 6, 80))
        nr = int(np.clip(round(384 * np.exp(0.50 * rng.normal())), 64, 1500))
        out.append((nl, nr, int(np.clip(nl // 5, 0 if nl < 8 else 1, 14))))
    return out


def make_set_complex(i, size, seed=7, all_atoms=True):
    """complex i of the set (deterministic in (seed, i)); falls back to fewer rotatable bonds when a small ligand cannot realise R"""
    nl, nr, R = size
    for r in range(R, -1, -1):
        try:
            c = make_complex(Nl=nl, Nr=nr, R=r, knn=24, seed=1000 * seed + i, name=f"set{i}")
            return add_atoms(c, seed=1000 * seed + i) if all_atoms else c
        except RuntimeError:
            continue
    raise RuntimeError(f"could not build complex {i} of size {size}")
