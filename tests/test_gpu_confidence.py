"""GPU parity of the all-atom confidence engine (cbd_conf_* C ABI -> HIP kernels) against the golden produced by running
the reference (tests/golden/g8_confidence.npz) and against the CPU oracle (oracle/confidence_ref.py).
Tolerances: fp32 throughout; per-layer features 1e-4 relative to the layer's max magnitude, confidences 2e-5 absolute."""
import copy
import os

import numpy as np
import pytest
import torch

from tests.helpers import to_aacx

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def conf_model():
    from confidence_bootstrapping_amd.utils import make_confidence_model
    model, args = make_confidence_model(device="cuda:0", seed=5)
    return model, args


def _poses(cplx, B, seed, spread):
    g = torch.Generator().manual_seed(seed)
    base = cplx["ligand"].pos
    return torch.stack([base + spread * torch.randn(1, 3, generator=g) + 0.4 * torch.randn(base.shape, generator=g) for _ in range(B)])


def _oracle(model, cplx, pos, crop=20.0):
    from oracle import confidence_ref as cr
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = cr.ConfConfig(crop_beyond=crop if crop else 1e9)
    return cr.confidence_forward(sd, to_aacx(cplx), pos, cfg, record=True)


def _check_layers(eng, ref, n_lig, tol=1e-4):
    for l in range(1, 6):
        got = eng.fetch(f"lig_layer{l}").reshape(n_lig, 84)
        want = ref[f"node_attr{l}"][:n_lig].numpy()
        d = want.shape[1]
        err = np.abs(got[:, :d] - want).max() / max(1.0, np.abs(want).max())
        assert err < tol, (l, err)
        assert np.all(got[:, d:] == 0)


def test_confidence_matches_reference_golden(conf_model):
    from confidence_bootstrapping_amd.synthetic import make_workload
    model, _ = conf_model
    g = np.load(os.path.join(GOLD, "g8_confidence.npz"))
    cplx = make_workload("tiny", all_atoms=True)
    eng = model.engine()
    eng.set_complex(cplx)
    eng.set_option("debug", 1)
    pos = torch.from_numpy(g["tiny_pos"])
    conf, atom = eng.score(pos.cuda(), crop_beyond=20.0)
    eng.set_option("debug", 0)
    B, Nl = pos.shape[:2]
    keep = eng.fetch("keep_res").reshape(B, -1)
    assert keep.sum(1).astype(int).tolist() == g["tiny_n_res"].tolist()
    for l in range(1, 6):
        got = eng.fetch(f"lig_layer{l}").reshape(B * Nl, 84)
        want = g[f"tiny_lig_layer{l}"]
        err = np.abs(got[:, :want.shape[1]] - want).max() / max(1.0, np.abs(want).max())
        assert err < 1e-4, (l, err)
    assert np.abs(conf.cpu().numpy() - g["tiny_confidence"]).max() < 2e-5
    assert np.abs(atom.cpu().numpy() - g["tiny_atom_confidence"]).max() < 2e-5


@pytest.mark.parametrize("workload,B,spread", [("tiny", 5, 6.0), ("c2_dockgen_median", 4, 5.0)])
def test_confidence_matches_oracle(conf_model, workload, B, spread):
    from confidence_bootstrapping_amd.synthetic import make_workload
    model, _ = conf_model
    cplx = make_workload(workload, all_atoms=True)
    pos = _poses(cplx, B, 3, spread)
    ref = _oracle(model, cplx, pos)
    eng = model.engine()
    eng.set_complex(cplx)
    eng.set_option("debug", 1)
    conf, atom = eng.score(pos.cuda(), crop_beyond=20.0)
    eng.set_option("debug", 0)
    counts = eng.edge_counts()
    want = dict(zip(("ll", "lr", "la", "rr", "rl", "ra", "aa", "al", "ar"), ref["edge_counts"].tolist()))
    assert counts == want, (counts, want)
    _check_layers(eng, ref, B * pos.shape[1])
    assert (conf.cpu() - ref["confidence"]).abs().max() < 2e-5
    assert (atom.cpu() - ref["atom_confidence"]).abs().max() < 2e-5
    # bitwise reproducible (fixed-order segmented reduction); a pose scored alone agrees up to the summation order
    # inside a node's edge run (its edges fall on different 32-edge tile boundaries)
    conf2, atom2 = eng.score(pos.cuda(), crop_beyond=20.0)
    assert torch.equal(conf, conf2) and torch.equal(atom, atom2)
    conf1, _ = eng.score(pos[1:2].cuda(), crop_beyond=20.0)
    assert abs(float(conf1[0] - conf[1])) < 2e-6


def test_confidence_edge_cases(conf_model):
    """Pose far from the receptor (crop keeps nothing: ligand-only graph), mixed with a normal pose; no crop at all."""
    from confidence_bootstrapping_amd.synthetic import make_workload
    model, _ = conf_model
    cplx = make_workload("tiny", all_atoms=True)
    pos = _poses(cplx, 3, 11, 2.0)
    pos[1] += 500.0
    ref = _oracle(model, cplx, pos)
    assert ref["n_res"][1] == 0 and ref["n_res"][0] > 0
    eng = model.engine()
    eng.set_complex(cplx)
    conf, atom = eng.score(pos.cuda(), crop_beyond=20.0)
    assert (conf.cpu() - ref["confidence"]).abs().max() < 2e-5
    assert (atom.cpu() - ref["atom_confidence"]).abs().max() < 2e-5
    ref_nc = _oracle(model, cplx, pos[:1], crop=None)
    conf_nc, _ = eng.score(pos[:1].cuda(), crop_beyond=None)
    assert (conf_nc.cpu() - ref_nc["confidence"]).abs().max() < 2e-5
    with pytest.raises(RuntimeError):
        eng.score(torch.zeros(eng.max_batch + 1, eng.Nl, 3).cuda())
    bad = copy.deepcopy(cplx)
    bad["atom"].x = bad["atom"].x.clone()
    bad["atom"].x[3, 2] = 23.0          # atom_type_2 has 23 classes: index 23 is out of range
    with pytest.raises(RuntimeError, match="out of range"):
        eng.set_complex(bad)
    eng.set_complex(cplx)


def test_confidence_forward_api(conf_model):
    """forward(batch) with the reference's contract: Batch of poses of one complex -> (confidence [B], atom_confidence [B*Nl,1])."""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    model, args = conf_model
    cplx = make_workload("tiny", all_atoms=True)
    pos = _poses(cplx, 3, 5, 3.0)
    graphs = []
    for b in range(3):
        gph = copy.deepcopy(cplx)
        gph["ligand"].pos = pos[b].clone()
        graphs.append(gph)
    batch = Batch.from_data_list(graphs).to("cuda:0")
    model.crop_beyond = args.crop_beyond
    conf, atom = model(batch)
    ref = _oracle(model, cplx, pos)
    assert conf.shape == (3,) and atom.shape == (3 * pos.shape[1], 1)
    assert (conf.cpu() - ref["confidence"]).abs().max() < 2e-5


def test_sampling_returns_confidence(conf_model):
    """sampling(..., confidence_model=, filtering_data_list=, filtering_model_args=) like inference.py:537-560: the
    confidences returned are those of the FINAL poses (oracle on the returned coordinates)."""
    from functools import partial
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_score_model
    from confidence_bootstrapping_amd.sampling import sampling, randomize_position
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule, t_to_sigma
    cmodel, cargs = conf_model
    smodel, sargs = make_score_model(device="cuda:0", seed=0)
    cplx = make_workload("tiny", all_atoms=True)
    N, S = 5, 4
    torch.manual_seed(3)
    np.random.seed(3)
    data_list = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(N)]
    filt_list = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(N)]
    randomize_position(data_list, False, False, 3.0)    # stay near the pocket so that the crop keeps residues
    sched = get_t_schedule("expbeta", S)
    out, conf = sampling(data_list, smodel, S, sched, sched, sched, torch.device("cuda:0"), partial(t_to_sigma, args=sargs), sargs,
                         batch_size=3, confidence_model=cmodel, filtering_data_list=filt_list, filtering_model_args=cargs)
    assert conf.shape == (N,) and torch.isfinite(conf).all()
    final = torch.stack([d["ligand"].pos.cpu() for d in out])
    ref = _oracle(cmodel, cplx, final)
    assert (conf.cpu() - ref["confidence"]).abs().max() < 2e-5


def test_end_to_end_demo_runs():
    """tools/dock_demo.py: randomize -> sampling with confidence -> ranking -> symmetric RMSD, on the tiny complex."""
    from tools.dock_demo import main
    conf, rmsds = main(["--samples", "6", "--steps", "3", "--workload", "tiny", "--batch-size", "4"])
    assert conf.shape == (6,) and torch.isfinite(conf).all() and torch.isfinite(rmsds).all()


def test_confidence_score_multi_equals_separate_calls(conf_model):
    """cbd_conf_score_multi: the pose batches of several complexes (different sizes, different batch sizes) in one set of fused-conv
    launches give bitwise the confidences and atom confidences of separate cbd_conf_score calls; an engine may appear only once."""
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.engine import ConfidenceEngine
    model, _ = conf_model
    main = model.engine()
    engines = [main] + model.co_engines(2, main)
    cplxs = [make_workload("tiny", all_atoms=True), make_workload("c2_dockgen_median", seed=77, all_atoms=True),
             make_workload("tiny", seed=5, all_atoms=True)]
    poses = [_poses(c, B, 20 + k, 1.5).cuda() for k, (c, B) in enumerate(zip(cplxs, (3, 5, 2)))]
    for e, c in zip(engines, cplxs):
        e.set_complex(c)
    single = [e.score(p, crop_beyond=20.0) for e, p in zip(engines, poses)]
    multi = ConfidenceEngine.score_multi(engines, poses, crop_beyond=20.0)
    for (c1, a1), (c2, a2) in zip(single, multi):
        assert torch.isfinite(c1).all() and torch.equal(c1, c2) and torch.equal(a1, a2)
    with pytest.raises(RuntimeError, match="once per call"):
        ConfidenceEngine.score_multi([main, main], poses[:1] * 2, crop_beyond=20.0)
