import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
from confidence_bootstrapping_amd import train_ops as to
from confidence_bootstrapping_amd.score_model import FCBlock
torch.manual_seed(0)
sizes = [100, 3000, 33, 1]
E = sum(sizes)
fcs = [FCBlock(96, 96, 64, 0.0).to(dev) for _ in sizes]
x = torch.randn(E, 96, device=dev)
xa = x.clone().requires_grad_(); xb = x.clone().requires_grad_()
to.FUSED_FIRST_STAGE = True
ha = to.fc_first_stage(xa, sizes, fcs)
to.FUSED_FIRST_STAGE = False
hb = to.fc_first_stage(xb, sizes, fcs)
print("fwd max diff", float((ha - hb).abs().max()), "max", float(hb.abs().max()))
g = torch.randn_like(ha)
ga = torch.autograd.grad(ha, [xa] + [p for fc in fcs for p in (fc[0].weight, fc[0].bias)], g)
gb = torch.autograd.grad(hb, [xb] + [p for fc in fcs for p in (fc[0].weight, fc[0].bias)], g)
for k, (a, b) in enumerate(zip(ga, gb)):
    print(k, tuple(a.shape), "max diff", float((a - b).abs().max()), "scale", float(b.abs().max()))
# dropout statistics and determinism
for fc in fcs:
    fc[2].p = 0.25
    fc.train()
to.FUSED_FIRST_STAGE = True
seed = torch.tensor([12345], device=dev)
h1 = to.fc_first_stage(x, sizes, fcs, seed=seed, call=1)
h1b = to.fc_first_stage(x, sizes, fcs, seed=seed, call=1)
h2 = to.fc_first_stage(x, sizes, fcs, seed=seed, call=2)
for fc in fcs:
    fc[2].p = 0.0
h0 = to.fc_first_stage(x, sizes, fcs)
act = h0 > 0
kept = (h1 > 0) & act
print("repeatable", bool(torch.equal(h1, h1b)), "keep rate", float(kept.sum() / act.sum()), "scale ok", float((h1[kept] / h0[kept]).mean()),
      "calls differ", float(((h1 > 0) != (h2 > 0))[act].float().mean()))
