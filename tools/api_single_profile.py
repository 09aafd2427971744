"""Host timeline + cProfile of ONE sampling() call on ONE complex (the shape of the reference's inference.py loop; bench.py's `single_complex`
leg, api_per_call): where do the ~15 ms between the engine-level 183 ms and the call's 198 ms go?   python tools/api_single_profile.py"""
import cProfile
import io
import os
import pstats
import sys
import time
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload, ideal_path_noise, BENCH_GEOMETRY, scale_tr_head
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule, t_to_sigma
    from confidence_bootstrapping_amd.engine import DockEngine, ConfidenceEngine
    import confidence_bootstrapping_amd.sampling as smp
    import confidence_bootstrapping_amd.score_model as sm
    dev = torch.device("cuda:0")
    model, margs = make_score_model(device=dev, seed=0)
    scale_tr_head(model)
    cmodel, cargs = make_confidence_model(device=dev, seed=5)
    S, N = 20, 40
    sched = get_t_schedule("expbeta", S)
    t2s = partial(t_to_sigma, args=margs)
    base = make_workload("c2_dockgen_median", seed=1234, all_atoms=True, **BENCH_GEOMETRY)
    pocket = base["ligand"].pos.mean(0)
    R = int(base["ligand"].edge_mask.sum())
    T = {}

    def stamp(tag, fn):
        def w(*x, **k):
            if "t0" not in T:
                return fn(*x, **k)
            t1 = time.perf_counter()
            out = fn(*x, **k)
            t2 = time.perf_counter()
            print(f"  +{1e3 * (t1 - T['t0']):8.2f} ms  {tag:28s} {1e3 * (t2 - t1):7.2f} ms")
            return out
        return w
    DockEngine.set_complex = stamp("score set_complex", DockEngine.set_complex)
    DockEngine.sample = stamp("sample", DockEngine.sample)
    ConfidenceEngine.set_complex = stamp("confidence set_complex", ConfidenceEngine.set_complex)
    ConfidenceEngine.score_multi = staticmethod(stamp("confidence score_multi", ConfidenceEngine.score_multi))
    sm.weights_version = stamp("weights_version", sm.weights_version)
    smp.draw_noise_like_reference = stamp("draw_noise_like_reference", smp.draw_noise_like_reference)

    def one(k, trace=False, prof=None):
        c = base.shallow_copy()
        c.name = f"single{k}"
        torch.manual_seed(700 + k); np.random.seed(700 + k)
        b1 = Batch.from_data_list([c])
        dl = [b1.shallow_copy() for _ in range(N)]
        smp.randomize_position(dl, False, False, margs.tr_sigma_max)
        for g in dl:
            g["ligand"].pos = g["ligand"].pos + (pocket - base["receptor"].pos.mean(0))
        ztr = ideal_path_noise(torch.stack([g["ligand"].pos for g in dl]), pocket, sched, margs)
        torch.cuda.synchronize()
        if trace:
            T["t0"] = time.perf_counter()
        if prof:
            prof.enable()
        t0 = time.perf_counter()
        filt = [g.shallow_copy() for g in dl]
        ta = time.perf_counter()
        noise = smp.draw_noise_like_reference(len(dl), R, S, N)
        noise["tr"] = ztr
        tb = time.perf_counter()
        out, conf = smp.sampling(data_list=dl, model=model, inference_steps=S, tr_schedule=sched, rot_schedule=sched, tor_schedule=sched,
                                 device=dev, t_to_sigma=t2s, model_args=margs, confidence_model=cmodel, filtering_data_list=filt,
                                 filtering_model_args=cargs, batch_size=N, noise=noise)
        tc = time.perf_counter()
        torch.cuda.synchronize()
        td = time.perf_counter()
        if prof:
            prof.disable()
        if trace:
            print(f"  filtering copies {1e3 * (ta - t0):.2f} ms | noise {1e3 * (tb - ta):.2f} ms | sampling() {1e3 * (tc - tb):.2f} ms | final sync {1e3 * (td - tc):.2f} ms | total {1e3 * (td - t0):.2f} ms")
            T.pop("t0")
        return td - t0
    one(-1); one(-2)
    print("timeline of one call:")
    one(0, trace=True)
    print("timeline of the next call:")
    one(1, trace=True)
    pr = cProfile.Profile()
    one(2, prof=pr)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
    print(s.getvalue()[:6000])


if __name__ == "__main__":
    main()
