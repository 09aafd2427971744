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
    ap.add_argument("--fused-adam", action="store_true")
    ap.add_argument("--blas", default=None, help="torch.backends.cuda.preferred_blas_library: cublas (= rocBLAS) | cublaslt (= hipBLASLt)")
    ap.add_argument("--regions", action="store_true", help="GPU launches per forward region and per backward node type")
    ap.add_argument("--sites", action="store_true", help="non-cbd GPU kernels (ATen, rocPRIM ...) of one step by calling source line / backward node")
    ap.add_argument("--gc", default="default", help="default | off | freeze: Python cyclic GC during the timed steps")
    ap.add_argument("--plain", action="store_true", help="free-running steps only (for rocprofv3 --kernel-trace; see tools/gap_stats.py)")
    ap.add_argument("--cprofile", action="store_true", help="Python-level profile (cProfile) of 16 free-running steps instead of the op table")
    a = ap.parse_args()
    from confidence_bootstrapping_amd.synthetic import make_complex, WORKLOADS
    from confidence_bootstrapping_amd.utils import make_score_model, load_model_args, ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function
    from confidence_bootstrapping_amd.train_forward import forward, collate
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    dev = torch.device("cuda:0")
    torch.set_num_threads(max(1, int(os.environ.get("CBD_HOST_THREADS", "1"))))     # what training.train_epoch runs under (hostcfg.py)
    margs = load_model_args()
    model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False)
    model.train()
    if a.blas:
        torch.backends.cuda.preferred_blas_library(a.blas)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, **({"fused": True} if a.fused_adam else {}))
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    t2s = partial(t_to_sigma, args=margs)
    loss_fn = partial(loss_function, tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
    base = [make_complex(name=f"cplx{i}", seed=1234 + i, **WORKLOADS["c2_dockgen_median"]) for i in range(a.batch)]
    nt = NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False)
    np.random.seed(0); torch.manual_seed(0)
    batches = [[nt(c.shallow_copy()) for c in base] for _ in range(8)]   # like CBBuffer.get: shallow copies sharing the complex's tensors
    sync = torch.cuda.synchronize
    params = list(model.parameters())
    from contextlib import nullcontext
    from torch.profiler import record_function
    reg = (lambda n: record_function("R:" + n)) if a.regions else (lambda n: nullcontext())

    def step(data, tm=None):
        def lap(name, t0):
            if tm is not None:
                sync()
                tm[name] = tm.get(name, 0.0) + time.perf_counter() - t0
            return time.perf_counter()
        t = time.perf_counter()
        with reg("zero_grad"):
            opt.zero_grad()
        t = lap("zero_grad", t)
        with reg("forward_other"):
            out = forward(model, data)
        t = lap("forward (incl. collate)", t)
        with reg("loss"):
            lt = loss_fn(*out, data=data, t_to_sigma=t2s, device=dev)
        t = lap("loss", t)
        lt[0].backward()
        t = lap("backward", t)
        with reg("adam"):
            opt.step()
        t = lap("adam", t)
        with reg("ema"):
            ema.update(params)
        t = lap("ema", t)

    for k in range(12):             # the caching allocator needs a few steps of every size before it stops calling hipMalloc
        step(batches[k % 8])
    sync()
    import gc
    if a.gc == "off":
        gc.collect(); gc.disable()
    elif a.gc == "freeze":
        gc.collect(); gc.freeze()
    reps = []
    for rep in range(10 if a.plain else 3):
        t0 = time.perf_counter()
        for k in range(16):
            step(batches[k % 8])
        sync()
        reps.append((time.perf_counter() - t0) / 16 * 1e3)
        if not a.plain:
            print(f"batch {a.batch}: {reps[-1]:.1f} ms per step (free-running, 16 steps)")
    if a.plain:
        print(f"batch {a.batch}: free-running ms per step over {len(reps)} x 16 steps: min {min(reps):.1f} median {sorted(reps)[len(reps) // 2]:.1f} all",
              " ".join(f"{r:.1f}" for r in reps))
    if a.plain:
        return
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
    from torch.profiler import profile, ProfilerActivity
    if a.regions:
        import collections
        from torch.profiler import record_function
        import confidence_bootstrapping_amd.train_forward as tf
        import confidence_bootstrapping_amd.train_ops as to
        import confidence_bootstrapping_amd.training as tr

        def wrap(mod, name):
            fn = getattr(mod, name)

            def inner(*x, **k):
                with record_function("R:" + name):
                    return fn(*x, **k)
            setattr(mod, name, inner)
        for name in ("collate", "radius", "radius_graph", "conv_layer", "atom_encoder", "center_tensor_product", "bond_tensor_product",
                     "irreps_batch_norm", "gaussian_smearing", "edge_cat", "gather_pad", "take"):
            wrap(tf, name)
        wrap(to, "csr_of")

        to.StreamHub.pack = (lambda f: (lambda self: (lambda r: r)(f(self))))(to.StreamHub.pack)
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            with record_function("R:step"):
                step(batches[7])
            sync()
        LAUNCH = ("hipLaunchKernel", "hipExtModuleLaunchKernel", "hipMemcpyAsync", "hipMemsetAsync")
        evs = [e for e in prof.events()]
        launches = sorted((e.time_range.start, e) for e in evs if e.name in LAUNCH)
        named = [e for e in evs if e.name.startswith("R:") or e.name.startswith("autograd::engine::evaluate_function")]
        named.sort(key=lambda e: (e.time_range.start, -e.time_range.end))
        count = collections.Counter()
        ncall = collections.Counter()
        for e in named:
            ncall[e.name] += 1
        for t, l in launches:
            owner = None
            for e in named:                     # innermost enclosing named range on the same thread
                if e.thread == l.thread and e.time_range.start <= t <= e.time_range.end:
                    if owner is None or e.time_range.start >= owner.time_range.start:
                        owner = e
            count[owner.name if owner else "(outside)"] += 1
        print("GPU launches by innermost forward region / backward node type (one step):")
        for name, n in count.most_common(60):
            print(f"  {n:6d}  calls {ncall[name]:5d}  {name}")
        print("  total", sum(count.values()))
        return
    if a.sites:
        import collections
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            step(batches[7])
            sync()
        evs = list(prof.events())
        nodes = [e for e in evs if e.name.startswith("autograd::engine::evaluate_function")]
        cnt, tim = collections.Counter(), collections.Counter()
        tot_n = tot_t = cbd_t = 0.0
        for e in evs:
            for k in (e.kernels or []):
                if "cbd::" in k.name:
                    cbd_t += k.duration
                    continue
                site = None
                for fr in (e.stack or []):
                    if "confidence_bootstrapping_amd" in fr or "tools/train_profile" in fr:
                        site = fr.split("confidence_bootstrapping_amd/")[-1]
                        if "tools/train_profile" not in fr:
                            break
                if site is None or "train_profile" in site:
                    own = None
                    for n in nodes:
                        if n.thread == e.thread and n.time_range.start <= e.time_range.start <= n.time_range.end:
                            own = n
                    site = (own.name.replace("autograd::engine::evaluate_function: ", "bwd:") if own else (site or "(no python frame)")) + " <- " + e.name
                else:
                    site += " <- " + e.name
                cnt[site] += 1; tim[site] += k.duration
                tot_n += 1; tot_t += k.duration
        print(f"non-cbd GPU kernels in one step: {int(tot_n)} launches, {tot_t / 1e3:.2f} ms; cbd kernels {cbd_t / 1e3:.2f} ms")
        for site, n in cnt.most_common(a.rows * 2):
            print(f"  {n:4d}  {tim[site] / 1e3:7.3f} ms  {site[:170]}")
        return
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
