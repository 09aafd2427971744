"""Diagnostic: workgroup lifetimes of single launches of the register-stationary bf16 kernel (CBD_BF16_DIAG=5: start / end stamps only).
One forward pass of 64 C4 poses at several diffusion times (the cross-edge count, hence the launch size, grows with t); prints, for the
LAST 74 -> 74 launch of the pass, the launch span against the median / max / min workgroup lifetime."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.diag_lib import use_diag_library
use_diag_library()      # phase stamps / timing-only variants / role split exist in experiments/libcbdock_diag.so only
os.environ.setdefault("CBD_BF16_DIAG", "5")
from confidence_bootstrapping_amd.synthetic import make_workload, BENCH_GEOMETRY
from confidence_bootstrapping_amd.utils import make_score_model
from confidence_bootstrapping_amd.engine import DockEngine, make_steps
dev = torch.device("cuda:0")
model, args = make_score_model(seed=0)
cplx = make_workload("c4_large_pocket", seed=1234, **BENCH_GEOMETRY)
B = 64
eng = DockEngine(dev, max_batch=B); eng.load_state_dict(model.state_dict()); eng.set_complex(cplx)
eng.set_option("bf16", 1); eng.set_option("bf16_stationary", 1)
g = torch.Generator().manual_seed(0)
pos = (cplx["ligand"].pos[None].repeat(B, 1, 1) + 2 * torch.randn(B, 1, 3, generator=g)).to(dev)
for t in (1.0, 0.6, 0.3, 0.05):
    step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
    for _ in range(3):
        eng.score(pos, step)
    torch.cuda.synchronize()
    raw = eng.fetch("conv_clock_s", 64)
    print(f"t = {t}: edges {eng.edge_counts()}  launch span {raw[40] / 1e3:.1f} us; workgroup lifetime median {raw[41] / 1e3:.1f} / max {raw[42] / 1e3:.1f} / "
          f"min {raw[43] / 1e3:.1f} us; {int(raw[46])} workgroups; ns per unit by role (workgroups): " + ", ".join(f"{raw[48 + 2 * r]:.0f} ({int(raw[49 + 2 * r])})" for r in range(4)))
