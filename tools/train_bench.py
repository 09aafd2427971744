"""Fine-tuning step benchmark (BASELINE.json configs[4], C5: confidence-bootstrapping fine-tune): one optimisation step =
NoiseTransform'ed batch -> train-mode forward (HIP tensor-product op) -> score-matching loss -> backward -> Adam -> EMA.

    python tools/train_bench.py [--batch 8] [--steps 10] [--warmup 3] [--workload c2_dockgen_median]

Prints one JSON line: complexes/s, ms/step, and the split forward / backward / optimizer measured with HIP events.
Synthetic complexes of the workload's size (different seeds = different receptors and ligands), random-init weights."""
import argparse
import copy
import json
import os
import sys
import time
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default="c2_dockgen_median")
    ap.add_argument("--dropout", type=float, default=None)
    a = ap.parse_args()
    from confidence_bootstrapping_amd.synthetic import make_complex, WORKLOADS
    from confidence_bootstrapping_amd.utils import make_score_model, load_model_args, ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function, train_step
    from confidence_bootstrapping_amd.train_forward import forward as forward_train
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma

    dev = torch.device("cuda:0")
    margs = load_model_args()
    if a.dropout is not None:
        margs.dropout = a.dropout
    model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)   # as utils.get_optimizer_and_scheduler builds it on a GPU
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    t2s = partial(t_to_sigma, args=margs)
    loss_fn = partial(loss_function, tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
    base = [make_complex(name=f"cplx{i}", seed=1234 + i, **WORKLOADS[a.workload]) for i in range(a.batch)]
    nt = NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False)
    np.random.seed(0)
    torch.manual_seed(0)
    batches = [[nt(c.shallow_copy()) for c in base] for _ in range(a.warmup + a.steps)]   # like CBBuffer.get
    params = list(model.parameters())
    for k in range(a.warmup):
        train_step(model, batches[k], opt, dev, t2s, loss_fn, ema)
    torch.cuda.synchronize()
    # (1) the number that counts: the product's own loop, training.train_epoch, over `steps` batches (look-ahead thread and all);
    #     the same batches through bare train_step calls beside it
    from confidence_bootstrapping_amd.training import train_epoch
    loader = [batches[a.warmup + k] for k in range(a.steps)]
    train_epoch(model, loader[:4], opt, dev, t2s, loss_fn, ema)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    train_epoch(model, loader, opt, dev, t2s, loss_fn, ema)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    t0 = time.perf_counter()
    for k in range(a.steps):
        train_step(model, batches[a.warmup + k], opt, dev, t2s, loss_fn, ema)
    torch.cuda.synchronize()
    el_bare = time.perf_counter() - t0
    # (2) the same steps once more, phase by phase, with HIP events on the compute stream and around the two training kernels
    #     (GPU-side durations: a phase that waits for the host shows up as a long phase)
    from confidence_bootstrapping_amd.train_ops import TIMER
    TIMER.enabled = True
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(a.steps)]
    for k in range(a.steps):
        data = batches[a.warmup + k]
        opt.zero_grad()
        ev[k][0].record()
        tr, rot, tor, sc = forward_train(model, data)
        loss = loss_fn(tr, rot, tor, sc, data=data, t_to_sigma=t2s, device=dev)[0]
        ev[k][1].record()
        loss.backward()
        ev[k][2].record()
        opt.step()
        ema.update(params)
        ev[k][3].record()
        torch.cuda.synchronize()
    f = np.mean([e[0].elapsed_time(e[1]) for e in ev])
    b = np.mean([e[1].elapsed_time(e[2]) for e in ev])
    o = np.mean([e[2].elapsed_time(e[3]) for e in ev])
    ks = TIMER.summary()
    TIMER.enabled = False
    peak = 157.3   # fp32 MFMA dense peak, TFLOP/s (MI355X_MICROARCH.md)
    roof = {k: {"ms_per_step": round(ms / a.steps, 3), "launches_per_step": n // a.steps, "achieved": round(fl / (ms * 1e-3) / 1e12, 2),
                "peak": peak, "unit": "TFLOP/s", "frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4)} for k, (ms, n, fl) in ks.items()}
    print(json.dumps({"metric": "fine-tuning complexes/s (1 GPU)", "roofline": {"bound": "mfma", "kernels": {"tp_train_fwd_kernel": roof.get("fwd"),
                      "tp_train_bwd_kernel (matrix-core work = the forward's, re-computed)": roof.get("bwd")}}, "value": round(a.batch * a.steps / el, 2), "unit": "complexes/s",
                      "ms_per_step": round(el / a.steps * 1e3, 2), "ms_per_step_bare_train_step_loop": round(el_bare / a.steps * 1e3, 2), "forward_ms": round(float(f), 2), "backward_ms": round(float(b), 2),
                      "optimizer_ema_ms": round(float(o), 2), "batch": a.batch, "workload": a.workload, "dropout": margs.dropout,
                      "final_loss": float(loss)}))


if __name__ == "__main__":
    main()
