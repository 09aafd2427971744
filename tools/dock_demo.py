"""End-to-end use of the reference-shaped API on one (synthetic) complex, the flow of the reference's inference.py:409-590
without its dataset/rdkit plumbing: N copies of the complex -> randomize_position -> sampling() with the score model and the
all-atom confidence model -> poses ranked by confidence -> symmetry-corrected RMSD of every pose to the input ("crystal") pose.

  python tools/dock_demo.py [--samples 40] [--steps 20] [--workload c2_dockgen_median]"""
import argparse
import copy
import os
import sys
import time
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=40)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--workload", default="c2_dockgen_median")
    ap.add_argument("--batch-size", type=int, default=10, help="the reference's default; batches of one complex are merged on the GPU")
    a = ap.parse_args(argv)
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model
    from confidence_bootstrapping_amd.sampling import sampling, randomize_position
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule, t_to_sigma
    from confidence_bootstrapping_amd.molecules_utils import symmetry_rmsd
    dev = torch.device("cuda:0")
    score_model, score_args = make_score_model(device=dev, seed=0)          # stand-ins for the two checkpoints
    conf_model, conf_args = make_confidence_model(device=dev, seed=5)
    cplx = make_workload(a.workload, all_atoms=True)
    crystal = cplx["ligand"].pos.clone()
    torch.manual_seed(0); np.random.seed(0)
    data_list = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(a.samples)]
    conf_list = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(a.samples)]
    randomize_position(data_list, score_args.no_torsion, False, score_args.tr_sigma_max)
    sched = get_t_schedule("expbeta", a.steps)
    # first call: the engines are created lazily (weights re-packed into MFMA tile streams and uploaded, complex uploaded)
    warm = [copy.deepcopy(d) for d in data_list[:2]]
    t0 = time.perf_counter()
    sampling(data_list=warm, model=score_model, inference_steps=a.steps, tr_schedule=sched, rot_schedule=sched, tor_schedule=sched,
             device=dev, t_to_sigma=partial(t_to_sigma, args=score_args), model_args=score_args, confidence_model=conf_model,
             filtering_data_list=conf_list[:2], filtering_model_args=conf_args, batch_size=a.batch_size)
    torch.cuda.synchronize()
    print(f"engine set-up + first call: {(time.perf_counter() - t0) * 1e3:.0f} ms")
    t0 = time.perf_counter()
    data_list, confidence = sampling(data_list=data_list, model=score_model, inference_steps=a.steps, tr_schedule=sched,
                                     rot_schedule=sched, tor_schedule=sched, device=dev, t_to_sigma=partial(t_to_sigma, args=score_args),
                                     model_args=score_args, confidence_model=conf_model, filtering_data_list=conf_list,
                                     filtering_model_args=conf_args, batch_size=a.batch_size, no_final_step_noise=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    poses = torch.stack([d["ligand"].pos for d in data_list])
    order = torch.argsort(confidence, descending=True)
    # molecular graph of the synthetic ligand: atomic-number feature + bond list
    nums = cplx["ligand"].x[:, 0].numpy()
    ei = cplx["ligand", "ligand"].edge_index.numpy()
    am = np.zeros((len(nums), len(nums)), dtype=int)
    am[ei[0], ei[1]] = 1
    rmsds = torch.tensor(symmetry_rmsd(crystal, poses, nums, am))
    print(f"{a.samples} poses x {a.steps} steps + confidence in {dt * 1e3:.1f} ms ({a.samples / dt:.1f} poses/s incl. host glue)")
    for rank, i in enumerate(order[:5].tolist()):
        print(f"  rank {rank + 1}: pose {i:2d}  confidence {confidence[i]:+.4f}  symmetric RMSD to the input pose {rmsds[i]:.2f} A")
    return confidence, rmsds


if __name__ == "__main__":
    main()
