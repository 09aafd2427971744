"""Soak of the pipelined set-up in sampling() (run by hand on a GPU box): random sets of random complexes (sizes, rotatable bonds, poses per
complex, co-scheduling width) through sampling() -- engines re-used across waves and calls, set-up of wave k + 1 under wave k -- against
one fresh synchronous engine per complex; bitwise.      python tools/fuzz_pipeline.py [seed] [rounds]"""
import copy
import os
import sys
from functools import partial

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from confidence_bootstrapping_amd import Batch
from confidence_bootstrapping_amd.synthetic import make_complex
from confidence_bootstrapping_amd.utils import make_score_model
from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma, get_t_schedule
from confidence_bootstrapping_amd.engine import DockEngine, make_steps, _single_complex
from confidence_bootstrapping_amd.sampling import sampling, randomize_position, draw_noise_like_reference

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
model, margs = make_score_model(device=dev, seed=0)
bad = 0
for rd in range(rounds):
    n_c, per, S, R = int(rng.integers(2, 12)), int(rng.integers(1, 6)), int(rng.integers(2, 7)), int(rng.integers(0, 4))
    co = int(rng.integers(1, 5))
    cps = []
    while len(cps) < n_c:
        try:
            cps.append(make_complex(Nl=int(rng.integers(max(5, 2 * R + 3), 34)), Nr=int(rng.integers(12, 120)), R=R, knn=int(rng.integers(4, 12)),
                                    seed=int(rng.integers(1, 10 ** 6)), name=f"f{rd}_{len(cps)}"))
        except (RuntimeError, ValueError):
            continue
    torch.manual_seed(seed + rd); np.random.seed(seed + rd)
    base = [Batch.from_data_list([copy.deepcopy(c)]) for c in cps for _ in range(per)]
    randomize_position(base, False, False, margs.tr_sigma_max)
    sched = get_t_schedule("expbeta", S)
    noise = draw_noise_like_reference(per * n_c, R, S, per)
    steps = make_steps(sched, margs, model.timestep_emb_func)
    want = []
    for i in range(n_c):
        e = DockEngine.from_model(model, dev, max_batch=8)
        e.set_complex(_single_complex(base[i * per])[0])
        pos = torch.stack([d["ligand"].pos for d in base[i * per:(i + 1) * per]]).to(dev).contiguous()
        sl = slice(i * per, (i + 1) * per)
        e.sample(pos, steps, noise["tr"][:, sl].to(dev), noise["rot"][:, sl].to(dev),
                 None if R == 0 else noise["tor"][:, i * per * R:(i + 1) * per * R].to(dev))
        torch.cuda.synchronize()
        want.append(pos.cpu())
        del e
    for rep in range(2):
        out, _ = sampling(data_list=[copy.deepcopy(d) for d in base], model=model, inference_steps=S, tr_schedule=sched, rot_schedule=sched,
                          tor_schedule=sched, device=dev, t_to_sigma=partial(t_to_sigma, args=margs), model_args=margs, batch_size=per,
                          noise=noise, co_schedule=co)
        for i in range(n_c):
            got = torch.stack([d["ligand"].pos for d in out[i * per:(i + 1) * per]]).cpu()
            if not torch.equal(got, want[i]):
                bad += 1
                print(f"MISMATCH round {rd} rep {rep} complex {i}: {float((got - want[i]).abs().max()):.3e}")
    print(f"round {rd}: {n_c} complexes x {per} poses, S={S}, R={R}, co_schedule={co}: ok" if not bad else f"round {rd}: {bad} mismatches so far", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
