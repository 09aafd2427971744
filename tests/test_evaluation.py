"""evaluation.performance_metrics against the dictionary produced by EXECUTING the reference's own aggregate block
(inference.py:593-885, oracle/make_golden_eval.py); pose_metrics against direct numpy formulas (inference.py:505-548)."""
import os

import numpy as np

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g13_eval.npz")


def test_performance_metrics_match_reference_block():
    from confidence_bootstrapping_amd.evaluation import performance_metrics
    g = np.load(G)
    for tag, with_conf in (("n12", True), ("n6", True), ("n3_noconf", False)):
        pm = performance_metrics(g[f"{tag}_rmsds"], g[f"{tag}_centroid"], g[f"{tag}_self"], g[f"{tag}_conf"] if with_conf else None,
                                 run_times=g[f"{tag}_times"], without_rec_overlap=g[f"{tag}_overlap"])
        keys = [str(k) for k in g[f"{tag}_keys"]]
        assert set(pm) == set(keys), set(pm) ^ set(keys)
        for k, v in zip(keys, g[f"{tag}_vals"]):
            assert abs(float(pm[k]) - float(v)) <= 1e-9 * max(1.0, abs(float(v))), (tag, k, pm[k], v)


def test_pose_metrics_without_symmetry():
    from confidence_bootstrapping_amd.evaluation import pose_metrics
    rng = np.random.default_rng(0)
    ref = rng.normal(size=(9, 3)).astype(np.float32)
    poses = ref[None] + rng.normal(scale=0.5, size=(4, 9, 3)).astype(np.float32)
    rmsd, cent, selfd = pose_metrics(poses, ref)
    np.testing.assert_allclose(rmsd, np.sqrt(((poses - ref) ** 2).sum(-1).mean(-1)), rtol=1e-6)
    np.testing.assert_allclose(cent, np.linalg.norm(poses.mean(1) - ref.mean(0), axis=1), rtol=1e-5)
    d = np.linalg.norm(poses[:, :, None] - poses[:, None], axis=-1)
    d = np.where(np.eye(9, dtype=bool), np.inf, d)
    np.testing.assert_allclose(selfd, d.min(axis=(1, 2)), rtol=1e-5)
