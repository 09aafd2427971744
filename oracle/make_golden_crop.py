"""Golden vectors for the score model's `crop_beyond` (reference utils/sampling.py:101-108, utils/utils.py:395-420): the REFERENCE's own
`utils.sampling.sampling()` run here with `model_args.crop_beyond` set, noise recorded.

TEST INFRASTRUCTURE ONLY (needs /root/reference and the cached tables of oracle/gen_tables.py).  Output:
  tests/golden/g18_sampling_crop.npz   pos0, schedule, the drawn noise, per-step scores, the number of residues the model saw per pose and
                                       step, final poses (tiny complex, B = 3, S = 6, crop_beyond = 8 A, translation head scaled by 0.02 as in bench.py so
                                       that the poses stay at the receptor: the crop bites from the middle steps on, differently per pose)
What is NOT the reference's code: torch_geometric's `subgraph` and `Batch.to_data_list` (absent here), restated from their published
semantics (oracle/ref_import.py::subgraph for a boolean mask with relabel_nodes; confidence_bootstrapping_amd/hetero.py::to_data_list).
Usage: python oracle/make_golden_crop.py"""
from __future__ import annotations

import copy
import os
import sys
from functools import partial

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def main():
    from oracle import ref_import
    from oracle.make_golden import npz
    hetero = ref_import.install()
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_score_model
    from utils.diffusion_utils import get_t_schedule, t_to_sigma
    import utils.sampling as ref_sampling
    import utils.utils as ref_utils
    ref_utils.subgraph = ref_import.subgraph
    torch.set_num_threads(8)
    from confidence_bootstrapping_amd.synthetic import scale_tr_head
    mine, margs = make_score_model(seed=0)
    scale_tr_head(mine)          # random-init weights: without it the translation score is a random walk that leaves the receptor at once
    sd = {k: v.clone() for k, v in mine.state_dict().items()}
    ref_model, _ = ref_import.reference_score_model(sd)
    cplx = make_workload("tiny")
    cplx["receptor"].side_chain_vecs = torch.zeros(cplx["receptor"].pos.shape[0], 4, 3)     # carried through the crop, not read by the model
    B, S, crop = 3, 6, 8.0
    torch.manual_seed(654)
    np.random.seed(654)
    data_list = [hetero.Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
    ref_sampling.randomize_position(data_list, False, False, 3.0)       # poses near the pocket: the crop then keeps a pose-dependent subset
    pos0 = torch.stack([d["ligand"].pos for d in data_list])
    # the late part of the reverse process (t = 0.3 -> 0.05: sigma_tr 0.48 -> 0.13 A): with random-init weights nothing pulls a pose back,
    # and at t ~ 1 a single step's noise (25 A) throws it off the receptor, where the crop keeps nothing
    sched = np.linspace(0.3, 0.05, S)
    args = copy.deepcopy(margs)
    args.crop_beyond = crop
    args.all_atoms = False
    drawn, step_scores, n_res = [], [], []
    real_normal = torch.normal

    def rec_normal(*a, **k):
        z = real_normal(*a, **k)
        drawn.append(z.clone())
        return z
    ref_sampling.DataLoader = hetero.DataLoader
    ref_sampling.Batch = hetero.Batch
    ref_sampling.crop_beyond = ref_utils.crop_beyond
    orig_forward = ref_model.forward

    def spy(batch):
        n_res.append(torch.bincount(batch["receptor"].batch, minlength=B).clone())
        out = orig_forward(batch)
        step_scores.append([o.clone() for o in out[:3]])
        return out
    torch.manual_seed(79)
    torch.normal = rec_normal
    try:
        out_list, conf = ref_sampling.sampling([copy.deepcopy(d) for d in data_list], spy, S, sched, sched, sched, torch.device("cpu"),
                                               partial(t_to_sigma, args=args), args, batch_size=B)
    finally:
        torch.normal = real_normal
    assert conf is None and len(drawn) == 3 * S
    n_res = torch.stack(n_res)
    print("residues seen per step and pose (of %d):" % cplx["receptor"].pos.shape[0], n_res.tolist())
    assert 0 < int(n_res.min()) < cplx["receptor"].pos.shape[0] and any(len(set(r)) > 1 for r in n_res.tolist())
    npz("g18_sampling_crop.npz", pos0=pos0, schedule=sched, crop_beyond=np.float64(crop),
        noise_tr=torch.stack(drawn[0::3]), noise_rot=torch.stack(drawn[1::3]), noise_tor=torch.stack(drawn[2::3]),
        final_pos=torch.stack([d["ligand"].pos for d in out_list]), n_res=n_res,
        step_tr=torch.stack([s[0] for s in step_scores]), step_rot=torch.stack([s[1] for s in step_scores]),
        step_tor=torch.stack([s[2] for s in step_scores]))


if __name__ == "__main__":
    main()
