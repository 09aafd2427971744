"""The bf16 role split experiment against the single-chain bf16 kernel -- runs on the DIAGNOSTIC library
(tools/diag_lib.py builds experiments/libcbdock_diag.so with -DCBD_DIAG -DCBD_EXPERIMENTS), not part of the product's test suite:
    python -m pytest experiments/test_role_split.py -q          (on a GPU box)"""
import copy
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.diag_lib import use_diag_library

pytestmark = pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")


@pytest.fixture(scope="module")
def model_args():
    use_diag_library()
    from confidence_bootstrapping_amd.utils import make_score_model
    return make_score_model(device="cuda:0", seed=0)


def test_bf16_role_split_options_agree(model_args):
    """cbd_set_option("bf16_roles", 1 | 2) (experimental): the cross / receptor groups as three weight-tile
    slices -- through the streaming kernel (1) or with the two 0e slices in the persistent LDS-resident kernel (2, tp_conv_bf16p.hip).
    Same products, summed slice by slice: 1e-5 relative to the single chain; the two forms are bitwise equal; both repeatable."""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.engine import make_steps
    from confidence_bootstrapping_amd.sampling import randomize_position
    model, args = model_args
    cplx = make_workload("c2_dockgen_median")
    torch.manual_seed(5); np.random.seed(5)
    dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(5)]
    randomize_position(dl, False, False, 5.0)
    pos = torch.stack([d["ligand"].pos for d in dl]).cuda()
    eng = model.engine()
    eng.set_complex(cplx)
    step = make_steps(np.array([0.5]), args, model.timestep_emb_func)[0]
    res = {}
    try:
        eng.set_option("bf16", 1)
        for mode in (0, 1, 2):
            eng.set_option("bf16_roles", mode)
            res[mode] = [x.clone() for x in eng.score(pos, step)]
            again = eng.score(pos, step)
            assert all(torch.equal(p, q) for p, q in zip(res[mode], again)), mode
    finally:
        eng.set_option("bf16_roles", 0)
        eng.set_option("bf16", 0)
    for p, q in zip(res[1], res[0]):
        assert float((p - q).abs().max()) <= 1e-5 * float(q.abs().max())
    assert not all(torch.equal(p, q) for p, q in zip(res[1], res[0]))      # the slices really ran
    assert all(torch.equal(p, q) for p, q in zip(res[2], res[1]))
