"""C3-style run (BASELINE.json configs[2]): a SET of heterogeneous complexes, 40 poses x 20 steps each + confidence
ranking, sharded over ranks by longest-processing-time greedy on Nl*Nr (no collective inside the loop; one gather of the
per-complex top pose at the end).  Synthetic complexes (sizes drawn from a DockGen-like spread); reports whole-job poses/s
including the per-complex set-up (graph upload, receptor embedding, all-atom static tables).

  python tools/run_set.py --complexes 24            (1 GPU)
  python -m torch.distributed.run --nproc-per-node N tools/run_set.py --complexes 189
"""
import argparse
import copy
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--complexes", type=int, default=24)
    ap.add_argument("--samples", type=int, default=40)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_complex, add_atoms
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model
    from confidence_bootstrapping_amd.engine import make_steps
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
    from confidence_bootstrapping_amd.sampling import randomize_position
    from confidence_bootstrapping_amd.distributed import shard_lpt

    rng = np.random.default_rng(a.seed)
    sizes = [(int(rng.integers(14, 56)), int(rng.integers(140, 720))) for _ in range(a.complexes)]
    parts = shard_lpt([nl * nr for nl, nr in sizes], world)
    mine = parts[rank]
    smodel, sargs = make_score_model(device=dev, seed=0)
    cmodel, cargs = make_confidence_model(device=dev, seed=5)
    from confidence_bootstrapping_amd.engine import DockEngine
    seng = DockEngine.from_model(smodel, dev, max_batch=a.samples)
    partners = []                                         # engines for the co-scheduled complexes (same weights)
    for _ in range(3):
        p_ = DockEngine(dev, max_batch=a.samples)
        p_.share_weights_from(seng)
        partners.append(p_)
    ceng = cmodel.engine(max_batch=a.samples)
    sched = get_t_schedule("expbeta", a.steps)
    steps = make_steps(sched, sargs, smodel.timestep_emb_func)
    # host-side synthesis of this rank's complexes is data loading, outside the timed region
    todo = []
    for i in mine:
        nl, nr = sizes[i]
        R = max(1, min(nl // 5, 10))
        c = add_atoms(make_complex(Nl=nl, Nr=nr, R=R, knn=24, seed=1000 + i, name=f"set{i}"), seed=1000 + i)
        torch.manual_seed(i); np.random.seed(i)
        dl = [Batch.from_data_list([copy.deepcopy(c)]) for _ in range(a.samples)]
        randomize_position(dl, False, False, sargs.tr_sigma_max)
        pos0 = torch.stack([d["ligand"].pos for d in dl]).contiguous()
        Rr = int(c["ligand"].edge_mask.sum())
        noise = (torch.randn(a.steps, a.samples, 3), torch.randn(a.steps, a.samples, 3), torch.randn(a.steps, a.samples * Rr))
        todo.append((i, c, pos0, noise))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t_setup = t_sample = t_conf = 0.0
    results = []
    t0 = time.perf_counter()
    k = 0
    while k < len(todo):
        group = todo[k:k + 4]           # consecutive complexes are co-scheduled four at a time (cbd_sample_multi)
        ta = time.perf_counter()
        engines = ([seng] + partners)[:len(group)]
        staged = []
        for e, (i, c, pos0, noise) in zip(engines, group):
            e.set_complex(c)
            staged.append((pos0.to(dev), [z.to(dev) for z in noise]))
        torch.cuda.synchronize()
        tb = time.perf_counter()
        DockEngine.sample_multi(engines, [st_[0] for st_ in staged], steps, [st_[1] for st_ in staged])
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for (i, c, _, _), (pos, _) in zip(group, staged):
            ceng.set_complex(c)
            conf, _ = ceng.score(pos, cargs.crop_beyond)
            best = int(torch.argmax(conf))
            results.append((i, float(conf[best]), pos[best].cpu()))
        td = time.perf_counter()
        t_setup += tb - ta; t_sample += tc - tb; t_conf += td - tc
        k += len(group)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tm = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        elapsed = float(tm.item())
        gathered = [None] * world if rank == 0 else None
        dist.gather_object([(i, cf) for i, cf, _ in results], gathered, dst=0)
    if rank == 0:
        print(json.dumps({"what": "heterogeneous complex set, sampling + confidence ranking, set-up included", "complexes": a.complexes,
                          "samples": a.samples, "denoise_steps": a.steps, "n_gpus": world, "poses_per_s": round(a.complexes * a.samples / elapsed, 2),
                          "s_total": round(elapsed, 3), "rank0": {"complexes": len(mine), "setup_s": round(t_setup, 3),
                                                                  "sampling_s": round(t_sample, 3), "confidence_s": round(t_conf, 3)}}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
