"""Generate tests/golden/g9_symm_rmsd.npz by RUNNING the reference's vendored spyrmsd (this container only):
spyrmsd.rmsd.symmrmsd(coords_ref, [coords...], atomicnums, atomicnums2, adjacency, adjacency2, return_permutation=True)
exactly as utils/molecules_utils.py:9-17 calls it.  TEST INFRASTRUCTURE ONLY.   python oracle/make_golden_rmsd.py"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def molecules():
    """(name, atomic numbers, adjacency) of a few graphs with different symmetry."""
    out = []
    # para-substituted benzene: ring of 6 C, two identical substituents (O) opposite each other -> 4 automorphisms
    n = 8
    am = np.zeros((n, n), int)
    for i in range(6):
        am[i, (i + 1) % 6] = am[(i + 1) % 6, i] = 1
    am[0, 6] = am[6, 0] = 1
    am[3, 7] = am[7, 3] = 1
    out.append(("para_benzene", np.array([6] * 6 + [8, 8]), am))
    # tert-butyl-like star: centre C with three identical CH3 arms and one N -> 6 automorphisms
    n = 5
    am = np.zeros((n, n), int)
    for i in range(1, 5):
        am[0, i] = am[i, 0] = 1
    out.append(("star", np.array([6, 6, 6, 6, 7]), am))
    # biphenyl-like: two 6-rings joined -> 8 automorphisms
    n = 12
    am = np.zeros((n, n), int)
    for r in (0, 6):
        for i in range(6):
            am[r + i, r + (i + 1) % 6] = am[r + (i + 1) % 6, r + i] = 1
    am[0, 6] = am[6, 0] = 1
    out.append(("biphenyl", np.array([6] * 12), am))
    # asymmetric chain with distinct elements -> identity only
    n = 6
    am = np.zeros((n, n), int)
    for i in range(5):
        am[i, i + 1] = am[i + 1, i] = 1
    out.append(("chain", np.array([6, 7, 8, 16, 6, 9]), am))
    return out


def main():
    from oracle import ref_import
    from oracle.make_golden import npz
    ref_import.install(load_tables=False)
    from spyrmsd import rmsd as ref_rmsd
    rng = np.random.default_rng(5)
    arrs = {}
    for name, nums, am in molecules():
        n = len(nums)
        ref = rng.normal(size=(n, 3)) * 2.0
        poses = [ref + rng.normal(scale=s, size=(n, 3)) for s in (0.05, 0.5, 2.0)]
        # a pose that is the reference with a symmetric relabelling applied: must give ~0
        import networkx as nx
        gm = nx.algorithms.isomorphism.GraphMatcher(nx.Graph(am), nx.Graph(am))
        last = None
        for m in gm.isomorphisms_iter():
            if all(nums[k] == nums[v] for k, v in m.items()):
                last = m
        perm = np.array([last[i] for i in range(n)])
        poses.append(ref[perm] + 1e-3 * rng.normal(size=(n, 3)))
        # second molecule with a different atom order
        order = rng.permutation(n)
        nums2, am2 = nums[order], am[np.ix_(order, order)]
        poses2 = [p[order] for p in poses]
        r1, p1 = ref_rmsd.symmrmsd(ref, poses, nums, nums, am, am, return_permutation=True)
        r2, p2 = ref_rmsd.symmrmsd(ref, poses2, nums, nums2, am, am2, return_permutation=True)
        G = ref_rmsd.graph.match_graphs(ref_rmsd.graph.graph_from_adjacency_matrix(am, nums), ref_rmsd.graph.graph_from_adjacency_matrix(am, nums))
        arrs.update({f"{name}_nums": nums, f"{name}_am": am, f"{name}_ref": ref, f"{name}_poses": np.stack(poses),
                     f"{name}_order": order, f"{name}_rmsd": np.array(r1), f"{name}_rmsd_reordered": np.array(r2),
                     f"{name}_n_iso": np.array(len(G)), f"{name}_perm_ref": np.array([p[0] for p in p1]), f"{name}_perm_pos": np.array([p[1] for p in p1])})
        print(name, "isomorphisms", len(G), "rmsd", np.round(r1, 4), np.round(r2, 4))
    npz("g9_symm_rmsd.npz", **arrs)


if __name__ == "__main__":
    main()
