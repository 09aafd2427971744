import copy, os, sys
from functools import partial
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
dev = torch.device("cuda:0")
from confidence_bootstrapping_amd.synthetic import make_complex, WORKLOADS
from confidence_bootstrapping_amd.utils import make_score_model, load_model_args
from confidence_bootstrapping_amd.training import loss_targets, loss_from_targets
from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
from confidence_bootstrapping_amd import train_forward as tf
margs = load_model_args(); margs.dropout = 0.0
t2s = partial(t_to_sigma, args=margs)
base = [make_complex(name=f"cplx{i}", seed=1234 + i, **WORKLOADS["tiny"]) for i in range(4)]
nt = NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False)
np.random.seed(0); torch.manual_seed(0)
data = [nt(c.shallow_copy()) for c in base]
model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False); model.train()
tg = loss_targets(data, t2s, dev)
prep = tf.prepare_batch(model, data, dev, pad=True)
print("pad", {k: v for k, v in prep.pad.items()})
for k, v in prep.g.__dict__.items():
    if torch.is_tensor(v) and v.is_floating_point():
        print("g.", k, tuple(v.shape), "nan" if torch.isnan(v).any() else "", "inf" if torch.isinf(v).any() else "")
print("t_ei tail", prep.g.t_ei[:, -3:].tolist(), "bonds", prep.g.bonds.tolist())
print("lig_ptr", prep.batch.lig_ptr.tolist(), "rec_ptr", prep.batch.rec_ptr.tolist())
# forward hooks on intermediate: monkeypatch conv_layer / irreps_batch_norm to report NaN
orig_bn = tf.irreps_batch_norm
def bn(bnm, x, *a, **k):
    out = orig_bn(bnm, x, *a, **k)
    print("BN in nan", bool(torch.isnan(x).any()), "out nan", bool(torch.isnan(out).any()), tuple(x.shape), k.get("exclude"))
    def hook(g):
        print("  BN grad_out nan", bool(torch.isnan(g).any()), tuple(g.shape))
    if out.requires_grad: out.register_hook(hook)
    return out
tf.irreps_batch_norm = bn
def wrap(name):
    orig = getattr(tf, name)
    def f(*a, **k):
        out = orig(*a, **k)
        ins = [("%s%s" % (tuple(x.shape), " NAN rows %s" % torch.isnan(x.reshape(x.shape[0], -1)).any(1).nonzero().flatten().tolist()[:6] if torch.isnan(x).any() else "")) for x in a if torch.is_tensor(x) and x.is_floating_point()]
        print(name, "in:", ins, "out nan rows:", torch.isnan(out.reshape(out.shape[0], -1)).any(1).nonzero().flatten().tolist()[:8] if torch.is_tensor(out) else None)
        return out
    setattr(tf, name, f)
for n in ("center_tensor_product", "bond_tensor_product", "scatter_mean"):
    wrap(n)
tr, rot, tor, _ = tf.forward(model, prep)
print("preds nan", [bool(torch.isnan(t).any()) for t in (tr, rot, tor)])
lt = loss_from_targets(tr, rot, tor, tg, tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
lt[0].backward()
bad = [n for n, p in model.named_parameters() if p.grad is not None and torch.isnan(p.grad).any()]
print("params with NaN grad:", len(bad), bad[:12])
