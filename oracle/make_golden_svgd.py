"""Golden vectors for SVGD sampling (reference utils/sampling.py:169-218, utils/torsion.py:121-185, utils/geometry.py:279-314): the
REFERENCE's own `utils.sampling.sampling()` run here with the svgd_* arguments set, noise recorded.

TEST INFRASTRUCTURE ONLY (needs /root/reference and the cached tables of oracle/gen_tables.py).  Output:
  tests/golden/g19_sampling_svgd.npz   pos0, schedule, the svgd arguments, the drawn noise, per-step scores and final poses of two runs on
                                       the tiny complex (N = 5 samples in one batch, S = 5 steps): `svgd_use_x0` False and True
Usage: python oracle/make_golden_svgd.py"""
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

SVGD = dict(svgd_weight_log_0=-0.5, svgd_weight_log_1=0.0, svgd_repulsive_weight_log_0=0.0, svgd_repulsive_weight_log_1=0.5,
            svgd_kernel_size_log_0=0.0, svgd_kernel_size_log_1=0.3, svgd_langevin_weight_log_0=-1.0, svgd_langevin_weight_log_1=-0.3,
            svgd_rot_log_rel_weight=0.3, svgd_tor_log_rel_weight=0.1)


def main():
    from oracle import ref_import
    from oracle.make_golden import npz
    hetero = ref_import.install()
    from confidence_bootstrapping_amd.synthetic import make_workload, scale_tr_head
    from confidence_bootstrapping_amd.utils import make_score_model
    from utils.diffusion_utils import t_to_sigma
    import utils.sampling as ref_sampling
    torch.set_num_threads(8)
    mine, margs = make_score_model(seed=0)
    scale_tr_head(mine)
    sd = {k: v.clone() for k, v in mine.state_dict().items()}
    ref_model, _ = ref_import.reference_score_model(sd)
    cplx = make_workload("tiny")
    N, S = 5, 5
    torch.manual_seed(321)
    np.random.seed(321)
    data_list = [hetero.Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(N)]
    ref_sampling.randomize_position(data_list, False, False, 3.0)
    pos0 = torch.stack([d["ligand"].pos for d in data_list])
    sched = np.linspace(0.5, 0.1, S)
    ref_sampling.DataLoader = hetero.DataLoader
    ref_sampling.Batch = hetero.Batch
    out = dict(pos0=pos0, schedule=sched, **{k: np.float64(v) for k, v in SVGD.items()})
    real_normal = torch.normal
    for tag, use_x0 in (("a", False), ("b", True)):
        drawn, step_scores = [], []

        def rec_normal(*a, **k):
            z = real_normal(*a, **k)
            drawn.append(z.clone())
            return z

        def spy(batch):
            o = ref_model(batch)
            step_scores.append([x.clone() for x in o[:3]])
            return o
        torch.manual_seed(97)
        torch.normal = rec_normal
        try:
            out_list, conf = ref_sampling.sampling([copy.deepcopy(d) for d in data_list], spy, S, sched, sched, sched, torch.device("cpu"),
                                                   partial(t_to_sigma, args=margs), margs, batch_size=N, svgd_use_x0=use_x0, **SVGD)
        finally:
            torch.normal = real_normal
        assert conf is None and len(drawn) == 3 * S
        final = torch.stack([d["ligand"].pos for d in out_list])
        print(tag, "use_x0", use_x0, "mean displacement", float((final - pos0).norm(dim=-1).mean()))
        out.update({f"noise_tr_{tag}": torch.stack(drawn[0::3]), f"noise_rot_{tag}": torch.stack(drawn[1::3]),
                    f"noise_tor_{tag}": torch.stack(drawn[2::3]), f"final_pos_{tag}": final,
                    f"step_tr_{tag}": torch.stack([s[0] for s in step_scores]), f"step_rot_{tag}": torch.stack([s[1] for s in step_scores]),
                    f"step_tor_{tag}": torch.stack([s[2] for s in step_scores])})
    # and the same noise WITHOUT svgd: the goldens must differ from the plain sampler by far more than the test tolerance
    torch.manual_seed(97)
    plain, _ = ref_sampling.sampling([copy.deepcopy(d) for d in data_list], ref_model, S, sched, sched, sched, torch.device("cpu"),
                                     partial(t_to_sigma, args=margs), margs, batch_size=N)
    out["final_pos_plain"] = torch.stack([d["ligand"].pos for d in plain])
    print("svgd vs plain, mean atom distance:", float((out["final_pos_a"] - out["final_pos_plain"]).norm(dim=-1).mean()))
    npz("g19_sampling_svgd.npz", **out)


if __name__ == "__main__":
    main()
