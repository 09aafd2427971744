"""One rank of `distributed.sampling_distributed` on the REAL engine (north-star split of ONE complex: samples round-robin over the
ranks, one confidence-ranked gather).  Started by tests/test_gpu_distributed.py as fresh processes:

  python tools/dist_sampling_check.py --out ref.npz                                   (world 1)
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
         tools/dist_sampling_check.py --backend nccl|gloo --out w2.npz                  (world 2)

backend nccl: one GPU per rank, the gather runs over RCCL; backend gloo: every rank uses cuda:0 (works on a 1-GPU box: the engines
of the two processes share the device, the gather runs on CPU tensors).  Rank 0 writes the ranked poses / confidences / indices.
"""
import argparse
import copy
import os
import sys
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--out", required=True)
    ap.add_argument("--samples", type=int, default=7)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--workload", default="tiny")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch.distributed as dist
    dev = torch.device("cuda", local if (a.backend == "nccl" and world > 1) else 0)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(dev)
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule, t_to_sigma
    from confidence_bootstrapping_amd.sampling import randomize_position
    from confidence_bootstrapping_amd.distributed import sampling_distributed
    smodel, sargs = make_score_model(device=dev, seed=0)
    cmodel, cargs = make_confidence_model(device=dev, seed=5)
    cplx = make_workload(a.workload, all_atoms=True)
    sched = get_t_schedule("expbeta", a.steps)
    torch.manual_seed(31)
    np.random.seed(31)
    dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(a.samples)]          # the same list on every rank
    randomize_position(dl, False, False, sargs.tr_sigma_max)
    torch.manual_seed(99)
    out = sampling_distributed(dl, smodel, a.steps, sched, sched, sched, dev, partial(t_to_sigma, args=sargs), sargs,
                               confidence_model=cmodel, filtering_model_args=cargs, batch_size=3)
    if rank == 0:
        np.savez(a.out, pos=out["pos"].cpu().numpy(), confidence=out["confidence"].cpu().numpy(), index=out["index"].cpu().numpy(),
                 world=world)
    else:
        assert out is None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
