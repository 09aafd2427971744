"""Diagnostic: in-kernel shader clock of tp_conv<3,3> (s_memtime / s_memrealtime stamps), run with CBD_CONV_VARIANT=8
after >= 2 s of back-to-back work on random data (MI355X_MICROARCH.md 'DVFS give-back' item 6)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.diag_lib import use_diag_library
use_diag_library()      # phase stamps / timing-only variants / role split exist in experiments/libcbdock_diag.so only
from confidence_bootstrapping_amd.synthetic import make_workload
from confidence_bootstrapping_amd.utils import make_score_model
from confidence_bootstrapping_amd.engine import DockEngine, make_steps
from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
BF16 = os.environ.get("CBD_BF16_DIAG") == "4"
assert BF16 or os.environ.get("CBD_CONV_VARIANT") in ("8", "13")
dev = torch.device("cuda:0")
model, args = make_score_model(seed=0)
cplx = make_workload("c4_large_pocket" if BF16 else "c2_dockgen_median")
eng = DockEngine(dev, max_batch=40); eng.load_state_dict(model.state_dict()); eng.set_complex(cplx)
if BF16:
    eng.set_option("bf16", 1)
steps = make_steps(get_t_schedule("expbeta", 20), args, model.timestep_emb_func)
g = torch.Generator().manual_seed(0)
B = 40
pos0 = (cplx["ligand"].pos[None].repeat(B, 1, 1) - cplx["ligand"].pos.mean(0) + 10 * torch.randn(B, 1, 3, generator=g)).to(dev)
noise = [torch.randn(20, B, 3, generator=g), torch.randn(20, B, 3, generator=g), torch.randn(20, B * eng.R, generator=g)]
t0 = time.time()
while time.time() - t0 < 4.0:
    p = pos0.clone(); eng.sample(p, steps, *noise); torch.cuda.synchronize()
ghz, dur_ns, n, pro, g1, tiles, fin, g1a = eng.fetch("conv_clock_ghz", 16)
print(f"in-kernel clock of tp_conv<3,3>: median {ghz:.3f} GHz over {int(n)} workgroups, median workgroup lifetime {dur_ns/1e3:.1f} us")
print(f"median cycles: prologue(gather+bias) {pro:.0f}, first Linear (3 tiles) {g1:.0f}, 54 weight tiles {tiles:.0f} ({tiles/54:.0f}/tile), "
      f"message reduce {fin:.0f}; total {pro+g1+tiles+fin:.0f}; first tile of the first Linear alone {g1a:.0f}")
if BF16:
    e0 = g1a - g1
    print(f"bf16: 0e block (38 scalar tiles) {e0:.0f} ({e0/38:.0f}/tile), vector blocks (16 tiles) {tiles-e0:.0f} ({(tiles-e0)/16:.0f}/tile)")
