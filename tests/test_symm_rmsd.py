"""Symmetry-corrected RMSD against tests/golden/g9_symm_rmsd.npz, produced by running the reference's
vendored spyrmsd (oracle/make_golden_rmsd.py).  CPU: the host isomorphism enumeration; GPU: cbd_symm_rmsd."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_symm_rmsd.npz")
NAMES = ("para_benzene", "star", "biphenyl", "chain")


def test_isomorphism_enumeration_matches_spyrmsd():
    from confidence_bootstrapping_amd.molecules_utils import graph_isomorphisms
    g = np.load(GOLD)
    for name in NAMES:
        nums, am = g[f"{name}_nums"], g[f"{name}_am"]
        idx1, idx2 = graph_isomorphisms(nums, am)
        assert idx1.shape == (int(g[f"{name}_n_iso"]), len(nums))
        assert all(np.array_equal(nums[i1], nums[i2]) for i1, i2 in zip(idx1, idx2))          # label preserving
        assert all(np.array_equal(am[np.ix_(i1, i1)], am[np.ix_(i2, i2)]) for i1, i2 in zip(idx1, idx2))   # edge preserving
        order = g[f"{name}_order"]
        j1, j2 = graph_isomorphisms(nums, am, nums[order], am[np.ix_(order, order)])
        assert j1.shape == idx1.shape
    other = g["chain_nums"].copy()
    other[0] = 35                      # different element: no label-preserving isomorphism
    with pytest.raises(ValueError):
        graph_isomorphisms(g["chain_nums"], g["chain_am"], other, g["chain_am"])


@pytest.mark.gpu
def test_symmetry_rmsd_matches_spyrmsd():
    from confidence_bootstrapping_amd.molecules_utils import symmetry_rmsd, get_symmetry_rmsd
    g = np.load(GOLD)
    for name in NAMES:
        nums, am, ref, poses, order = (g[f"{name}_{k}"] for k in ("nums", "am", "ref", "poses", "order"))
        got, perms = symmetry_rmsd(ref, [p for p in poses], nums, am, return_permutation=True)
        np.testing.assert_allclose(got, g[f"{name}_rmsd"], rtol=2e-5, atol=2e-6)
        for (i1, i2), w1, w2 in zip(perms, g[f"{name}_perm_ref"], g[f"{name}_perm_pos"]):
            # same minimising relabelling reference atom -> pose atom (no ties between isomorphisms for the noisy poses)
            assert dict(zip(i1, i2)) == dict(zip(w1.tolist(), w2.tolist()))
        got2 = symmetry_rmsd(ref, torch.from_numpy(poses[:, order]).float().cuda(), nums, am, nums[order], am[np.ix_(order, order)])
        np.testing.assert_allclose(got2, g[f"{name}_rmsd_reordered"], rtol=2e-5, atol=2e-6)

        class Mol:   # spyrmsd-Molecule-like duck type accepted by get_symmetry_rmsd
            atomicnums, adjacency_matrix = nums, am
        single = get_symmetry_rmsd(Mol, ref, poses[1])
        assert isinstance(single, float) and abs(single - float(g[f"{name}_rmsd"][1])) < 2e-5


def test_isomorphism_enumeration_is_bounded():
    """A highly symmetric ligand (factorially many automorphisms) must raise quickly instead of stalling / exhausting the host: the
    callers then fall back to the uncorrected RMSD like the reference after its time_limit(10) (inference.py:511-520)."""
    import time
    from confidence_bootstrapping_amd.molecules_utils import graph_isomorphisms, IsomorphismLimit
    n = 9
    adj = np.ones((n, n), dtype=int) - np.eye(n, dtype=int)          # K9, one element: 9! = 362 880 automorphisms
    nums = np.full(n, 6)
    t0 = time.monotonic()
    with pytest.raises(IsomorphismLimit):
        graph_isomorphisms(nums, adj, max_isomorphisms=500)
    with pytest.raises(IsomorphismLimit):
        graph_isomorphisms(nums, adj, time_limit_s=0.05)
    assert time.monotonic() - t0 < 5.0
    i1, i2 = graph_isomorphisms(np.array([6, 6, 8]), np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]]))      # C-C-O: only the identity
    assert i1.shape == (1, 3) and (i1 == i2).all()
