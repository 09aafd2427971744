"""Conditioning of the g11 training step: how much do the gradients move when ONE intermediate (the cross-edge embedding, [E, 32]) is
perturbed by 1e-7 relative -- the size of a single fp32 rounding?  (all-torch Linear layers, fused first stage)"""
import os, sys
from functools import partial
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
t = importlib.import_module("test_gpu_train_step")
from confidence_bootstrapping_amd import train_ops as to, train_forward as tf
from confidence_bootstrapping_amd.utils import make_score_model, load_model_args
from confidence_bootstrapping_amd.training import loss_function
from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
dev = torch.device("cuda:0")
from experiments.train_ops_reference import linear_reference
to.linear = lambda x, lin, act=0, p=0.0, seed=None, call=0: linear_reference(x, lin, act, p)      # all-torch Linear layers
orig = tf._mlp
EPS = [0.0]
ONLY = None
def pert(seq, x, seed=None, call=0):
    y = orig(seq, x, seed=seed, call=call)
    if EPS[0] and (ONLY is None or call == ONLY):
        gen = torch.Generator(device=dev).manual_seed(1 + call)
        y = y * (1 + EPS[0] * torch.randn(y.shape, device=dev, generator=gen))
    return y
tf._mlp = pert
def grads(eps):
    EPS[0] = eps
    margs = load_model_args(); margs.dropout = 0.0
    model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False); model.train()
    data = t._noised_batch()
    tr, rot, tor, _ = model(data)
    out = loss_function(tr, rot, tor, None, data=data, t_to_sigma=partial(t_to_sigma, args=margs), device=dev, tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
    out[0].backward()
    return {n: p.grad.double().cpu() for n, p in model.named_parameters() if p.grad is not None and p.numel()}
g0 = grads(0.0)
import builtins
for eps, only in ((1e-7, None), (1e-7, 120), (1e-7, 100), (1e-7, 150), (1e-7, 190), (1e-7, 160)):
    globals()["ONLY"] = only
    g1 = grads(eps)
    rel = sorted(((float((g1[n] - g0[n]).abs().max() / g0[n].abs().max()), n) for n in g0), reverse=True)
    print("eps", eps, "only", only, [("%.1e" % r, n) for r, n in rel[:5]], "median %.1e" % np.median([r for r, _ in rel]))
