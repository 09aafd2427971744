"""Diagnostic (diagnostic library): lifetime of EVERY workgroup of one launch of the register-stationary bf16 kernel -- is the launch time set
by a few stragglers, and what distinguishes them (units, role boundary, XCD = workgroup index mod 8)?   python tools/conv_span_wg.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.diag_lib import use_diag_library
use_diag_library()
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
for t in (0.6, 0.3):
    step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
    for rep in range(4):
        eng.score(pos, step)
        torch.cuda.synchronize()
        raw = eng.fetch("conv_span_wg", 1024).reshape(-1, 4)
        start, life, units, tag = raw[:, 0], raw[:, 1], raw[:, 2], raw[:, 3]
        role, wg = np.floor(tag).astype(int), np.rint((tag - np.floor(tag)) * 1024).astype(int)
        per = life / units
        order = np.argsort(-life)
        med = np.median(life)
        print(f"t = {t} rep {rep}: {len(life)} workgroups, units {units.min():.0f}..{units.max():.0f}; lifetime us median {med / 1e3:.1f} p90 {np.percentile(life, 90) / 1e3:.1f} "
              f"max {life.max() / 1e3:.1f} (+{100 * (life.max() / med - 1):.1f} %), min {life.min() / 1e3:.1f}; start offsets up to {start.max() / 1e3:.1f} us; "
              f"launch span {(start + life).max() / 1e3:.1f} us; mean lifetime {life.mean() / 1e3:.1f}")
        if rep == 3:
            print("   slowest ten: " + ", ".join(f"wg {wg[i]} (xcd {wg[i] % 8}, role {role[i]}, {units[i]:.0f} u, {life[i] / 1e3:.1f} us, {per[i]:.0f} ns/u)" for i in order[:10]))
            print("   ns per unit by last role: " + ", ".join(f"role {r}: {np.median(per[role == r]):.0f} (n {int((role == r).sum())})" for r in sorted(set(role))))
            print("   ns per unit by XCD: " + ", ".join(f"{x}: {np.median(per[wg % 8 == x]):.0f}" for x in range(8)))
            # workgroups in role-major order: where do the stragglers sit?
            by = np.argsort(wg)
            print("   lifetime by workgroup index (us, every 8th): " + " ".join(f"{life[i] / 1e3:.0f}" for i in by[::8]))
