"""Where does a fine-tuning step spend its HOST time?  (the step is host-bound at the reference's batch sizes.)
Phase timings with synchronisation between the phases, then a torch.profiler table of the ATen / custom ops by call count and self
CPU time.   python tools/train_profile.py [--batch 8]"""
import argparse
import copy
import os
import sys
import time
from functools import partial

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--rows", type=int, default=45)
    ap.add_argument("--cprofile", action="store_true", help="Python-level profile (cProfile) of 16 free-running steps instead of the op table")
    a = ap.parse_args()
    from confidence_bootstrapping_amd.synthetic import make_complex, WORKLOADS
    from confidence_bootstrapping_amd.utils import make_score_model, load_model_args, ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function
    from confidence_bootstrapping_amd.train_forward import forward, collate
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    dev = torch.device("cuda:0")
    margs = load_model_args()
    model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    t2s = partial(t_to_sigma, args=margs)
    loss_fn = partial(loss_function, tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
    base = [make_complex(name=f"cplx{i}", seed=1234 + i, **WORKLOADS["c2_dockgen_median"]) for i in range(a.batch)]
    nt = NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False)
    np.random.seed(0); torch.manual_seed(0)
    batches = [[nt(c.shallow_copy()) for c in base] for _ in range(8)]   # like CBBuffer.get: shallow copies sharing the complex's tensors
    sync = torch.cuda.synchronize

    def step(data, tm=None):
        def lap(name, t0):
            if tm is not None:
                sync()
                tm[name] = tm.get(name, 0.0) + time.perf_counter() - t0
            return time.perf_counter()
        t = time.perf_counter()
        opt.zero_grad()
        t = lap("zero_grad", t)
        out = forward(model, data)
        t = lap("forward (incl. collate)", t)
        lt = loss_fn(*out, data=data, t_to_sigma=t2s, device=dev)
        t = lap("loss", t)
        lt[0].backward()
        t = lap("backward", t)
        opt.step()
        t = lap("adam", t)
        ema.update(model.parameters())
        t = lap("ema", t)

    for k in range(12):             # the caching allocator needs a few steps of every size before it stops calling hipMalloc
        step(batches[k % 8])
    sync()
    for rep in range(3):
        t0 = time.perf_counter()
        for k in range(16):
            step(batches[k % 8])
        sync()
        print(f"batch {a.batch}: {(time.perf_counter() - t0) / 16 * 1e3:.1f} ms per step (free-running, 16 steps)")
    if a.cprofile:
        import cProfile, pstats
        pr = cProfile.Profile()
        pr.enable()
        for k in range(16):
            step(batches[k % 8])
        sync()
        pr.disable()
        st = pstats.Stats(pr)
        st.sort_stats("tottime").print_stats(a.rows)
        st.sort_stats("cumtime").print_stats(a.rows)
        return
    tm = {}
    for k in range(8):
        step(batches[k], tm)
    print("with a synchronisation after every phase (ms per step):", {k: round(v / 8 * 1e3, 2) for k, v in tm.items()})
    # host time of enqueueing alone: the same step with the GPU made irrelevant is not possible, so count ops instead
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        step(batches[7])
        sync()
    ka = prof.key_averages()
    n_ops = sum(e.count for e in ka if e.device_type == torch.autograd.DeviceType.CPU)
    print("CPU-side op calls in one step:", n_ops)
    launches = sum(e.count for e in ka if e.key in ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync"))
    print(f"GPU launches (kernels + async copies / memsets) in one step: {launches}")
    print(ka.table(sort_by="self_cpu_time_total", row_limit=a.rows, max_name_column_width=70))


if __name__ == "__main__":
    main()
