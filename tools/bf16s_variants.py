"""Diagnostic: timing-only variants of the register-stationary bf16 kernel (library built with CBD_BF16S_VARIANTS=1; WRONG results).
Average duration of the tensor-product launches of one forward pass of 64 C4 poses at t = 0.6, per variant (one process per variant:
the variant is read once per process from CBD_BF16_DIAG).   python tools/bf16s_variants.py [variant]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
NAMES = {0: "full kernel", 16: "no barrier", 32: "no 0e epilogue", 64: "no gathers", 128: "no reduction", 256: "no first Linear", 512: "no vector epilogue",
         992: "none of 32 .. 512"}
if len(sys.argv) < 2:
    for v in NAMES:
        env = dict(os.environ, CBD_BF16_DIAG=str(v))
        subprocess.run([sys.executable, os.path.abspath(__file__), str(v)], env=env)
    sys.exit(0)
import numpy as np, torch
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
step = make_steps(np.array([0.6]), args, model.timestep_emb_func)[0]
for _ in range(2):
    eng.score(pos, step)
eng.kernel_timing(enable=True, reset=True)
for _ in range(5):
    eng.score(pos, step)
torch.cuda.synchronize()
avg, n, tot = eng.kernel_timing(enable=False)
v = int(sys.argv[1])
print(f"variant {v:4d} ({NAMES[v]:22s}): {tot / 5:.3f} ms of tensor-product launches per forward pass ({n // 5} launches)")
