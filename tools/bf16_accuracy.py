"""How far a reduced-operand policy (argv[1] = "bf16" (default) or "f32_split") is from the exact-fp32 path on the C2 complex: per-step score errors and
the RMSD between the two 20-step trajectories under identical noise.  Prints one JSON line."""
import copy, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from confidence_bootstrapping_amd import Batch
from confidence_bootstrapping_amd.synthetic import make_workload
from confidence_bootstrapping_amd.utils import make_score_model
from confidence_bootstrapping_amd.engine import make_steps
from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
from confidence_bootstrapping_amd.sampling import randomize_position

MODE = sys.argv[1] if len(sys.argv) > 1 else "bf16"
model, args = make_score_model(device="cuda:0", seed=0)
cplx = make_workload("c2_dockgen_median")
B, S = 16, 20
torch.manual_seed(0); np.random.seed(0)
dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
randomize_position(dl, False, False, args.tr_sigma_max)
pos0 = torch.stack([d["ligand"].pos for d in dl]).cuda()
sched = get_t_schedule("expbeta", S)
steps = make_steps(sched, args, model.timestep_emb_func)
eng = model.engine(); eng.set_complex(cplx)
rel = {"tr": [], "rot": [], "tor": []}
for i in (0, 6, 12, 19):
    eng.set_option(MODE, 0); a = [x.clone() for x in eng.score(pos0, steps[i])]
    eng.set_option(MODE, 1); b = eng.score(pos0, steps[i])
    for k, x, y in zip(rel, a, b):
        rel[k].append(float((x - y).abs().max() / x.abs().max()))
R = int(cplx["ligand"].edge_mask.sum())
g = torch.Generator().manual_seed(1)
noise = [torch.randn(S, B, 3, generator=g), torch.randn(S, B, 3, generator=g), torch.randn(S, B * R, generator=g)]
out = {}
for mode in (0, 1):
    eng.set_option(MODE, mode)
    p = pos0.clone(); eng.sample(p, steps, *noise); out[mode] = p
eng.set_option(MODE, 0)
rmsd = torch.sqrt(((out[0] - out[1]) ** 2).sum(-1).mean(-1))
print(json.dumps({"mode": MODE, "max_rel_err_per_t": {k: [float(f'{v:.3g}') for v in vs] for k, vs in rel.items()},
                  "traj_rmsd_A": {"median": float(f"{float(rmsd.median()):.3g}"), "max": float(f"{float(rmsd.max()):.3g}")}}))
