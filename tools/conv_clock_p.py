"""Diagnostic: per-wave phase clocks of the persistent bf16 kernel (tp_conv_bf16p.hip), CBD_BF16_DIAG=4 CBD_BF16_ROLES=2.
Cycles per 64-edge unit of a wave: gather + mids, first Linear (3 streamed tiles), resident 0e tiles + bias, the rest of the wave's
lifetime (message reduction, weight fill, role set-up) divided by its units."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.diag_lib import use_diag_library
use_diag_library()      # phase stamps / timing-only variants / role split exist in experiments/libcbdock_diag.so only
os.environ.setdefault("CBD_BF16_DIAG", "4"); os.environ.setdefault("CBD_BF16_ROLES", "2")
from confidence_bootstrapping_amd.synthetic import make_workload, BENCH_GEOMETRY
from confidence_bootstrapping_amd.utils import make_score_model
from confidence_bootstrapping_amd.engine import DockEngine, make_steps
from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
dev = torch.device("cuda:0")
model, args = make_score_model(seed=0)
cplx = make_workload("c4_large_pocket", seed=1234, **BENCH_GEOMETRY)
B = 64
eng = DockEngine(dev, max_batch=B); eng.load_state_dict(model.state_dict()); eng.set_complex(cplx)
eng.set_option("bf16", 1)
steps = make_steps(get_t_schedule("expbeta", 20), args, model.timestep_emb_func)
g = torch.Generator().manual_seed(0)
pos0 = (cplx["ligand"].pos[None].repeat(B, 1, 1) - cplx["ligand"].pos.mean(0) + cplx["ligand"].pos.mean(0) + 2 * torch.randn(B, 1, 3, generator=g)).to(dev)
noise = [torch.randn(20, B, 3, generator=g), torch.randn(20, B, 3, generator=g), torch.randn(20, B * eng.R, generator=g)]
t0 = time.time()
while time.time() - t0 < 3.0:
    p = pos0.clone(); eng.sample(p, steps, *noise); torch.cuda.synchronize()
ghz, dur_ns, n, pro, g1, tiles, fin, g1a = eng.fetch("conv_clock_ghz", 16)
units = g1a
print(f"clock {ghz:.3f} GHz, {int(n)} waves, median wave lifetime {dur_ns/1e3:.1f} us, median units per wave {units:.1f}")
print(f"median cycles per wave: gather {pro:.0f}, first Linear {g1:.0f}, 0e tiles {tiles:.0f}, rest {fin:.0f}; "
      f"per unit: gather {pro/units:.0f}, first Linear {g1/units:.0f}, 0e {tiles/units:.0f} ({tiles/units/19:.0f}/tile), rest {fin/units:.0f}")
