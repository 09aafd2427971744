"""bf16 policy with and without the role split ("bf16_roles" option): scores of one forward pass on a workload, both against each other
and against the fp32 policy; timing of a 64 x 40 C4 run through bench.measure.   python tools/bf16_roles_check.py [--workload ...]"""
import argparse, copy, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from tools.diag_lib import use_diag_library
use_diag_library()      # the role split exists in experiments/libcbdock_diag.so only


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
    cplx = make_workload(a.workload, seed=1234, **BENCH_GEOMETRY)
    eng = DockEngine.from_model(model, dev, max_batch=a.batch)
    eng.set_complex(cplx)
    g = torch.Generator().manual_seed(0)
    pos = (cplx["ligand"].pos[None] + 2.0 * torch.randn(a.batch, 1, 3, generator=g)).to(dev).contiguous()
    out = {}
    for t in (0.9, 0.3):
        step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
        res = {}
        for name, opts in (("f32", {}), ("bf16", {"bf16": 1}), ("bf16_roles", {"bf16": 1, "bf16_roles": 1}), ("bf16_res", {"bf16": 1, "bf16_roles": 2})):
            eng.set_option("bf16", 0); eng.set_option("bf16_roles", 0)
            for k, v in opts.items():
                eng.set_option(k, v)
            res[name] = [x.cpu().clone() for x in eng.score(pos, step)]
            res[name + "_again"] = [x.cpu().clone() for x in eng.score(pos, step)]
        eng.set_option("bf16", 0); eng.set_option("bf16_roles", 0)
        rel = lambda x, y: max(float((p - q).abs().max() / q.abs().max()) for p, q in zip(x, y))
        out[str(t)] = {"bf16_vs_f32": rel(res["bf16"], res["f32"]), "roles_vs_f32": rel(res["bf16_roles"], res["f32"]),
                       "roles_vs_bf16": rel(res["bf16_roles"], res["bf16"]),
                       "roles_repeatable": all(torch.equal(p, q) for p, q in zip(res["bf16_roles"], res["bf16_roles_again"])),
                       "resident_vs_roles": rel(res["bf16_res"], res["bf16_roles"]), "resident_vs_f32": rel(res["bf16_res"], res["f32"]),
                       "resident_bitwise_roles": all(torch.equal(p, q) for p, q in zip(res["bf16_res"], res["bf16_roles"])),
                       "resident_repeatable": all(torch.equal(p, q) for p, q in zip(res["bf16_res"], res["bf16_res_again"]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
