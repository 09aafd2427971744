"""GPU busy time of the python_api leg of bench.py (sampling() with the confidence model over 20 complexes), for rocprofv3 --kernel-trace:
   rocprofv3 --kernel-trace --output-format csv -d out -o p -- python3 tools/api_profile.py [--no-conf]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--complexes", type=int, default=20)
    ap.add_argument("--timeline", action="store_true", help="host timeline of the timed sampling() call: start / end of every engine call, "
                    "with a device synchronisation in front of each one when --sync (what the GPU still had to do shows as the wait)")
    ap.add_argument("--sync", action="store_true")
    ap.add_argument("--cprofile", action="store_true", help="host profile of the timed sampling() call (cumulative, top 45)")
    a = ap.parse_args()
    import bench
    from confidence_bootstrapping_amd.synthetic import scale_tr_head, BENCH_GEOMETRY
    from confidence_bootstrapping_amd.utils import make_score_model
    dev = torch.device("cuda:0")
    model, margs = make_score_model(seed=0)
    scale_tr_head(model)
    if a.timeline:
        import confidence_bootstrapping_amd.sampling as smp
        from confidence_bootstrapping_amd.engine import DockEngine, ConfidenceEngine
        orig, calls, T = smp.sampling, [], {}

        def stamp(tag, fn):
            def w(*x, **k):
                if "t0" not in T:
                    return fn(*x, **k)
                w0 = 0.0
                if a.sync:
                    ts = time.perf_counter(); torch.cuda.synchronize(); w0 = time.perf_counter() - ts
                t1 = time.perf_counter()
                out = fn(*x, **k)
                t2 = time.perf_counter()
                print(f"  +{1e3 * (t1 - T['t0']):8.1f} ms  {tag:28s} {1e3 * (t2 - t1):7.1f} ms" + (f"   (device was busy for {1e3 * w0:.1f} ms more)" if a.sync else ""))
                return out
            return w
        DockEngine.set_complex = stamp("score set_complex", DockEngine.set_complex)
        DockEngine.sample_multi = staticmethod(stamp("sample_multi", DockEngine.sample_multi))
        ConfidenceEngine.set_complex = stamp("confidence set_complex", ConfidenceEngine.set_complex)
        ConfidenceEngine.score_multi = staticmethod(stamp("confidence score_multi", ConfidenceEngine.score_multi))
        import confidence_bootstrapping_amd.score_model as sm
        sm.weights_version = stamp("weights_version", sm.weights_version)

        def wrapped(**kw):
            calls.append(1)
            if len(calls) == 3:
                torch.cuda.synchronize()
                T["t0"] = time.perf_counter()
                print("timed sampling() call:")
            out = orig(**kw)
            if len(calls) == 3:
                print(f"  +{1e3 * (time.perf_counter() - T['t0']):8.1f} ms  sampling() returns")
                T.pop("t0")
            return out
        smp.sampling = wrapped
    if a.cprofile:
        import cProfile
        import pstats
        import confidence_bootstrapping_amd.sampling as smp
        orig, calls = smp.sampling, []

        def wrapped(**kw):
            calls.append(1)
            if len(calls) != 3:              # the leg's third call is the timed one
                return orig(**kw)
            pr = cProfile.Profile()
            t = time.perf_counter()
            out = pr.runcall(orig, **kw)
            print("timed sampling() call under cProfile:", round(time.perf_counter() - t, 3), "s")
            pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
            return out
        smp.sampling = wrapped
    t0 = time.perf_counter()
    r = bench.python_api_leg(model, margs, dev, "c2_dockgen_median", 40, 20, dict(BENCH_GEOMETRY), a.complexes, 229.0)
    print({k: v for k, v in r.items() if k != "what"}, round(time.perf_counter() - t0, 1), flush=True)


if __name__ == "__main__":
    main()
