"""The oracle's all-atom confidence forward (oracle/confidence_ref.py) against the golden produced by RUNNING the
reference's model class + crop_beyond + set_time (oracle/make_golden_confidence.py -> tests/golden/g8_confidence.npz)."""
import os

import numpy as np
import torch

from tests.helpers import to_aacx

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_confidence_oracle_matches_reference_run():
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_confidence_model
    from oracle import confidence_ref as cr
    g = np.load(os.path.join(GOLD, "g8_confidence.npz"))
    model, _ = make_confidence_model(seed=5)
    sd = {k: v.detach() for k, v in model.state_dict().items()}
    cx = to_aacx(make_workload("tiny", all_atoms=True))
    pos = torch.from_numpy(g["tiny_pos"])
    out = cr.confidence_forward(sd, cx, pos, record=True)
    assert out["n_res"].tolist() == g["tiny_n_res"].tolist() and out["n_atom"].tolist() == g["tiny_n_atom"].tolist()
    n_lig = pos.shape[0] * pos.shape[1]
    for l in range(1, 6):
        ref = torch.from_numpy(g[f"tiny_lig_layer{l}"])
        err = (out[f"node_attr{l}"][:n_lig] - ref).abs().max().item()
        assert err < 2e-5 * max(1.0, ref.abs().max().item()), (l, err)
    assert np.abs(out["confidence"].numpy() - g["tiny_confidence"]).max() < 2e-6
    assert np.abs(out["atom_confidence"].numpy() - g["tiny_atom_confidence"]).max() < 2e-6


def test_confidence_model_state_dict_layout():
    """264 entries / 3 883 676 parameters, the counts SURVEY.md 8f-1 records for the reference class."""
    from confidence_bootstrapping_amd.utils import make_confidence_model
    model, args = make_confidence_model(seed=5)
    sd = model.state_dict()
    assert len(sd) == 264 and sum(p.numel() for p in model.parameters()) == 3883676
    assert sd["conv_layers.3.fc.8.3.weight"].shape == (1944, 72) and sd["conv_layers.0.fc.0.3.weight"].shape == (720, 72)
    assert sd["conv_layers.4.batch_norm.weight"].shape == (60,) and "conv_layers.4.fc.3.0.weight" not in sd
    assert sd["atom_confidence_predictor.8.weight"].shape == (25, 24) and "atom_confidence_predictor.1.num_batches_tracked" in sd
