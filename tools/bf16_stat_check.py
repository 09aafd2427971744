"""bf16 policy through the register-stationary kernel (tp_conv_bf16s.hip, option "bf16_stationary") against the streaming bf16 kernel and
the fp32 policy: scores of one forward pass at two diffusion times, repeatability.   python tools/bf16_stat_check.py [--workload ...]"""
import argparse, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2_dockgen_median")
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    from confidence_bootstrapping_amd.synthetic import make_workload, BENCH_GEOMETRY
    from confidence_bootstrapping_amd.utils import make_score_model
    from confidence_bootstrapping_amd.engine import DockEngine, make_steps
    dev = torch.device("cuda:0")
    model, args = make_score_model(seed=0)
    kw = {} if a.workload == "tiny" else dict(seed=1234, **BENCH_GEOMETRY)
    cplx = make_workload(a.workload, **kw)
    eng = DockEngine.from_model(model, dev, max_batch=a.batch)
    eng.set_complex(cplx)
    g = torch.Generator().manual_seed(0)
    pos = (cplx["ligand"].pos[None] + 2.0 * torch.randn(a.batch, 1, 3, generator=g)).to(dev).contiguous()
    out = {"workload": a.workload, "batch": a.batch}
    for t in (0.9, 0.3):
        step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
        res = {}
        for name, opts in (("f32", {"bf16": 0}), ("stream", {"bf16": 1, "bf16_stationary": 0}), ("stat", {"bf16": 1, "bf16_stationary": 1})):
            for k, v in opts.items():
                eng.set_option(k, v)
            res[name] = [x.cpu().clone() for x in eng.score(pos, step)]
            res[name + "_again"] = [x.cpu().clone() for x in eng.score(pos, step)]
            torch.cuda.synchronize()
        eng.set_option("bf16", 0)
        rel = lambda x, y: max(float((p - q).abs().max() / q.abs().max()) for p, q in zip(x, y))
        out[str(t)] = {"stream_vs_f32": rel(res["stream"], res["f32"]), "stat_vs_f32": rel(res["stat"], res["f32"]),
                       "stat_vs_stream": rel(res["stat"], res["stream"]),
                       "finite": all(bool(torch.isfinite(p).all()) for p in res["stat"]),
                       "stat_repeatable": all(torch.equal(p, q) for p, q in zip(res["stat"], res["stat_again"]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
