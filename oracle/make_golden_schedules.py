"""Golden vectors for --different_schedules (reference inference.py:375-383, utils/sampling.py:93-144): the REFERENCE's own
`utils.sampling.sampling()` run here with three different time grids for translation / rotation / torsion, noise recorded.

TEST INFRASTRUCTURE ONLY (needs /root/reference and the cached tables of oracle/gen_tables.py).  Output:
  tests/golden/g14_sampling_schedules.npz   pos0, the three schedules, the drawn noise, per-step scores, final poses (tiny, B = 3, S = 8)
  tests/golden/g16_sampling_async.npz       the same for a model BUILT with asyncronous_noise_schedule (score_model.py:85) and the
                                            schedules of inference.py:384-388: a common time grid t and its beta-quantile images
                                            (get_inverse_schedule) for translation / rotation / torsion
Usage: python oracle/make_golden_schedules.py"""
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
    torch.set_num_threads(8)
    mine, margs = make_score_model(seed=0)
    sd = {k: v.clone() for k, v in mine.state_dict().items()}
    ref_model, _ = ref_import.reference_score_model(sd)
    cplx = make_workload("tiny")
    B, S = 3, 8
    torch.manual_seed(321)
    np.random.seed(321)
    data_list = [hetero.Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
    ref_sampling.randomize_position(data_list, False, False, margs.tr_sigma_max)
    pos0 = torch.stack([d["ligand"].pos for d in data_list])
    # the schedule family of inference.py's --sigma_schedule expbeta with different (alpha, beta) per component
    tr_s = get_t_schedule(sigma_schedule="expbeta", inference_steps=S, inf_sched_alpha=1, inf_sched_beta=1)
    rot_s = get_t_schedule(sigma_schedule="expbeta", inference_steps=S, inf_sched_alpha=2, inf_sched_beta=1)
    tor_s = get_t_schedule(sigma_schedule="expbeta", inference_steps=S, inf_sched_alpha=1, inf_sched_beta=3)
    assert not np.array_equal(tr_s, rot_s) and not np.array_equal(tr_s, tor_s)
    drawn, step_scores = [], []
    real_normal = torch.normal

    def rec_normal(*a, **k):
        z = real_normal(*a, **k)
        drawn.append(z.clone())
        return z
    ref_sampling.DataLoader = hetero.DataLoader
    ref_sampling.Batch = hetero.Batch
    orig_forward = ref_model.forward

    def spy(batch):
        out = orig_forward(batch)
        step_scores.append([o.clone() for o in out[:3]])
        return out
    torch.manual_seed(77)
    torch.normal = rec_normal
    try:
        out_list, conf = ref_sampling.sampling([copy.deepcopy(d) for d in data_list], spy, S, tr_s, rot_s, tor_s, torch.device("cpu"),
                                               partial(t_to_sigma, args=margs), margs, batch_size=B)
    finally:
        torch.normal = real_normal
    assert conf is None and len(drawn) == 3 * S
    npz("g14_sampling_schedules.npz", pos0=pos0, tr_schedule=tr_s, rot_schedule=rot_s, tor_schedule=tor_s,
        noise_tr=torch.stack(drawn[0::3]), noise_rot=torch.stack(drawn[1::3]), noise_tor=torch.stack(drawn[2::3]),
        final_pos=torch.stack([d["ligand"].pos for d in out_list]),
        step_tr=torch.stack([s[0] for s in step_scores]), step_rot=torch.stack([s[1] for s in step_scores]),
        step_tor=torch.stack([s[2] for s in step_scores]))

    # ---- asyncronous noise schedule: the reference model class built with the flag (same weights: it adds no parameters)
    from utils.diffusion_utils import get_inverse_schedule
    async_model, _ = ref_import.reference_score_model(sd, asyncronous_noise_schedule=True)
    t_s = get_t_schedule(sigma_schedule="expbeta", inference_steps=S, inf_sched_alpha=1, inf_sched_beta=1)
    tr_a, rot_a, tor_a = get_inverse_schedule(t_s, 1.0, 1.0), get_inverse_schedule(t_s, 2.0, 1.0), get_inverse_schedule(t_s, 1.0, 3.0)
    assert np.allclose(tr_a, t_s) and not np.allclose(rot_a, t_s)
    tr_a = get_inverse_schedule(t_s, 1.5, 1.0)          # all three differ from the common grid
    drawn.clear(); step_scores.clear()
    orig_async = async_model.forward

    def spy_async(batch):
        assert "t" in batch.complex_t
        out = orig_async(batch)
        step_scores.append([o.clone() for o in out[:3]])
        return out
    torch.manual_seed(78)
    torch.normal = rec_normal
    try:
        out_list, conf = ref_sampling.sampling([copy.deepcopy(d) for d in data_list], spy_async, S, tr_a, rot_a, tor_a, torch.device("cpu"),
                                               partial(t_to_sigma, args=margs), margs, batch_size=B,
                                               asyncronous_noise_schedule=True, t_schedule=t_s)
    finally:
        torch.normal = real_normal
    assert len(drawn) == 3 * S
    npz("g16_sampling_async.npz", pos0=pos0, t_schedule=t_s, tr_schedule=tr_a, rot_schedule=rot_a, tor_schedule=tor_a,
        noise_tr=torch.stack(drawn[0::3]), noise_rot=torch.stack(drawn[1::3]), noise_tor=torch.stack(drawn[2::3]),
        final_pos=torch.stack([d["ligand"].pos for d in out_list]),
        step_tr=torch.stack([s[0] for s in step_scores]), step_rot=torch.stack([s[1] for s in step_scores]),
        step_tor=torch.stack([s[2] for s in step_scores]))


if __name__ == "__main__":
    main()
