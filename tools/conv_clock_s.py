"""Diagnostic: per-wave phase clocks of the register-stationary bf16 kernel (tp_conv_bf16s.hip), CBD_BF16_DIAG=4.
Median cycles per 32-edge unit of each of the four waves of a workgroup: index / gather issue, first Linear (wave 3), tiles, reduction
(wave 2), LDS writes of the gathers, barrier wait."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.diag_lib import use_diag_library
use_diag_library()      # phase stamps / timing-only variants / role split exist in experiments/libcbdock_diag.so only
os.environ.setdefault("CBD_BF16_DIAG", "4")
from confidence_bootstrapping_amd.synthetic import make_workload, BENCH_GEOMETRY
from confidence_bootstrapping_amd.utils import make_score_model
from confidence_bootstrapping_amd.engine import DockEngine, make_steps
from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
dev = torch.device("cuda:0")
model, args = make_score_model(seed=0)
cplx = make_workload("c4_large_pocket", seed=1234, **BENCH_GEOMETRY)
B = 64
eng = DockEngine(dev, max_batch=B); eng.load_state_dict(model.state_dict()); eng.set_complex(cplx)
eng.set_option("bf16", 1); eng.set_option("bf16_stationary", 1)
steps = make_steps(get_t_schedule("expbeta", 20), args, model.timestep_emb_func)
g = torch.Generator().manual_seed(0)
pos0 = (cplx["ligand"].pos[None].repeat(B, 1, 1) + 2 * torch.randn(B, 1, 3, generator=g)).to(dev)
noise = [torch.randn(20, B, 3, generator=g), torch.randn(20, B, 3, generator=g), torch.randn(20, B * eng.R, generator=g)]
t0 = time.time()
while time.time() - t0 < 3.0:
    p = pos0.clone(); eng.sample(p, steps, *noise); torch.cuda.synchronize()
raw = eng.fetch("conv_clock_s", 64)
r = raw[:40].reshape(4, 10)
print(f"launch span {raw[40] / 1e3:.1f} us; workgroup lifetime median {raw[41] / 1e3:.1f} / max {raw[42] / 1e3:.1f} / min {raw[43] / 1e3:.1f} us; units per workgroup {raw[45]:.0f} .. {raw[44]:.0f}; {int(raw[46])} workgroups")
names = ["idx+issue", "firstLin", "tiles", "reduce", "ldsWrite", "barrier"]
if os.environ.get("CBD_BF16_DIAG") == "6":      # finer clocks of waves 0 .. 2 (wave 3: not instrumented)
    names = ["prologue", "hs 0-2", "hs 3-7", "hs 8-14", "tail+red", "barrier"]
for w in range(4):
    tot, ghz, units, n = r[w, 0], r[w, 1], max(r[w, 8], 1.0), r[w, 9]
    per = "  ".join(f"{nm} {r[w, 2 + k] / units:7.0f}" for k, nm in enumerate(names))
    print(f"wave {w}: {int(n)} records, clock {ghz:.2f} GHz, units/wg {units:.0f}, lifetime/unit {tot / units:7.0f} cycles | {per}")
