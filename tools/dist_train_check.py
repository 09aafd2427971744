"""One rank of the fine-tuning loop on the REAL HIP training step (training.train_epoch -> train_forward + tp_train kernels, the flat
gradient all-reduce that replaces the reference's DataParallel, utils/utils.py:285-286 / utils/training.py:184-233).  Started by
tests/test_gpu_train_distributed.py as fresh processes:

  python tools/dist_train_check.py --mode grads --out w1.npz                                   (world 1: the concatenated batches)
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
         tools/dist_train_check.py --backend gloo|nccl --mode grads|nan|full --out w2.npz        (world 2: every rank its half)

modes
  grads : per-sample-independent configuration (dropout 0, the e3nn BatchNorm layers on their running statistics) so that the mean of
          the two ranks' gradients IS the gradient of the concatenated batch: plain SGD, two steps of four complexes (two per rank);
          rank 0 writes the parameters after each step (delta / lr = the averaged gradient).
  nan   : the same, but rank 1's first batch carries a NaN score: both ranks must skip step 1 and take step 2.
  full  : the shipped configuration (train-mode BatchNorm, dropout 0.1, Adam + EMA): the ranks must end with identical parameters
          and identical BatchNorm running statistics (averaged at the end of the epoch).
backend gloo: both ranks on cuda:0 (runs on a 1-GPU box; the all-reduce goes through host memory); nccl: one GPU per rank (RCCL)."""
import argparse
import copy
import os
import sys
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LR = 0.05


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--mode", default="grads", choices=["grads", "nan", "full"])
    ap.add_argument("--out", required=True)
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
    from confidence_bootstrapping_amd.synthetic import make_complex, WORKLOADS
    from confidence_bootstrapping_amd.utils import make_score_model, load_model_args, ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function, train_epoch
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    from confidence_bootstrapping_amd.score_model import IrrepsBatchNorm
    margs = load_model_args()
    if a.mode != "full":
        margs.dropout = 0.0
    model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False)
    model.train()
    t2s = partial(t_to_sigma, args=margs)
    loss_fn = partial(loss_function, tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
    # two steps x four complexes of the same size and torsion count (mean of the ranks' means = mean over the concatenated batch)
    base = [make_complex(name=f"c{i}", seed=700 + i, **WORKLOADS["tiny"]) for i in range(8)]
    nt = NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False)
    np.random.seed(5)
    torch.manual_seed(5)
    noised = [nt(copy.deepcopy(c)) for c in base]                           # identical on every rank
    steps = [noised[0:4], noised[4:8]]
    if a.mode == "nan":
        bad = copy.deepcopy(steps[0][3])                                    # the last complex of step 1 = rank 1's second one
        bad.tr_score = torch.full_like(torch.as_tensor(bad.tr_score), float("nan"))
        steps[0] = steps[0][:3] + [bad]
    if world == 1:
        loader_steps = steps if a.mode != "nan" else steps[1:]              # the one-rank reference of `nan`: step 2 alone
    else:
        loader_steps = [s[2 * rank:2 * rank + 2] for s in steps]
    if a.mode == "full":
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    else:
        opt = torch.optim.SGD(model.parameters(), lr=LR)
        ema = None

        def keep_bn_on_running_stats():
            for m in model.modules():
                if isinstance(m, IrrepsBatchNorm):
                    m.eval()
    snaps = [torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy()]
    summaries = []
    if a.mode != "full":
        orig_train = model.train

        def train_keep(mode=True):                                          # train_epoch calls model.train(): re-pin the BatchNorms
            r = orig_train(mode)
            keep_bn_on_running_stats()
            return r
        model.train = train_keep
    for st in loader_steps:                                                 # one train_epoch call per step: a snapshot after each
        summaries.append(train_epoch(model, [st], opt, dev, t2s, loss_fn, ema))
        snaps.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy())
    torch.cuda.synchronize()
    bn = torch.cat([b.detach().reshape(-1).float() for n, b in model.named_buffers() if n.endswith(("running_mean", "running_var"))]).cpu()
    final = torch.from_numpy(snaps[-1])
    if world > 1:
        both_p = [torch.zeros_like(final) for _ in range(world)]
        both_b = [torch.zeros_like(bn) for _ in range(world)]
        if a.backend == "nccl":
            gp, gb = [t.to(dev) for t in both_p], [t.to(dev) for t in both_b]
            dist.all_gather(gp, final.to(dev))
            dist.all_gather(gb, bn.to(dev))
            both_p, both_b = [t.cpu() for t in gp], [t.cpu() for t in gb]
        else:
            dist.all_gather(both_p, final)
            dist.all_gather(both_b, bn)
    else:
        both_p, both_b = [final], [bn]
    if rank == 0:
        np.savez(a.out, snaps=np.stack(snaps), world=world, lr=LR, losses=np.asarray([s["loss"] for s in summaries]),
                 params_by_rank=np.stack([t.numpy() for t in both_p]), bn_by_rank=np.stack([t.numpy() for t in both_b]))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
