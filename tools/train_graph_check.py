"""Checks and timing of the hipGraph-captured fine-tuning step (train_graph.py) on one MI355X:
  python tools/train_graph_check.py [--workload c2_dockgen_median] [--batch 8] [--steps 16]
1. gradients of the capacity-padded eager step vs the plain eager step (dropout 0): equal up to the association of sums;
2. gradients of the graph replay vs the padded eager step: bitwise;
3. ms per step of training.train_epoch, eager and hip_graph=True (shipped configuration: dropout 0.1, Adam, EMA)."""
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


def flat_grads(model):
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in model.parameters()]).clone()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2_dockgen_median")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--skip-timing", action="store_true")
    ap.add_argument("--skip-checks", action="store_true")
    ap.add_argument("--modes", default="eager,hip_graph")
    ap.add_argument("--no-gc", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from confidence_bootstrapping_amd.synthetic import make_complex, WORKLOADS
    from confidence_bootstrapping_amd.utils import make_score_model, load_model_args, ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function, train_epoch, loss_targets, loss_from_targets
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    from confidence_bootstrapping_amd import train_forward as tf
    from confidence_bootstrapping_amd.train_graph import GraphedStep
    out = {}
    margs = load_model_args()
    t2s = partial(t_to_sigma, args=margs)
    lw = dict(tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
    base = [make_complex(name=f"cplx{i}", seed=1234 + i, **WORKLOADS[a.workload]) for i in range(a.batch)]
    nt = NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False)
    np.random.seed(0)
    torch.manual_seed(0)
    batches = [[nt(c.shallow_copy()) for c in base] for _ in range(8)]

    if not a.skip_checks:
        checks(a, dev, batches, margs, t2s, lw, out)
    if a.skip_timing:
        return
    timing(a, dev, batches, margs, t2s, lw)


def checks(a, dev, batches, margs, t2s, lw, out):
    from confidence_bootstrapping_amd.utils import make_score_model
    from confidence_bootstrapping_amd.training import loss_targets, loss_from_targets
    from confidence_bootstrapping_amd import train_forward as tf
    from confidence_bootstrapping_amd.train_graph import GraphedStep
    # ---- 1 + 2: gradients (dropout 0)
    m0 = copy.deepcopy(margs)
    m0.dropout = 0.0
    model, _ = make_score_model(device=dev, seed=0, args=m0, eval_mode=False)
    model.train()
    bn0 = {n: b.clone() for n, b in model.named_buffers()}

    def reset_bn():
        with torch.no_grad():
            for n, b in model.named_buffers():
                b.copy_(bn0[n])

    def grads_of(prep_or_list, targets):
        reset_bn()
        model.zero_grad(set_to_none=True)
        tr, rot, tor, _ = tf.forward(model, prep_or_list)
        lt = loss_from_targets(tr, rot, tor, targets, **lw)
        lt[0].backward()
        torch.cuda.synchronize()
        return flat_grads(model), float(lt[0])
    data = batches[0]
    tg = loss_targets(data, t2s, dev)
    g_plain, l_plain = grads_of(data, tg)
    g_plain2, _ = grads_of(data, tg)
    out["plain_repeatable_bitwise"] = bool(torch.equal(g_plain, g_plain2))
    prep = tf.prepare_batch(model, data, dev, pad=True)
    out["pad"] = {k: v for k, v in prep.pad.items() if k in ("edges_real", "B_real", "T_real", "buckets")}
    out["padded_edges"] = {"ll": int(prep.g.l_ei.shape[1]), "lr": int(prep.g.lr.shape[1]), "t": int(prep.g.t_ei.shape[1])}
    g_pad, l_pad = grads_of(prep, tg)
    scale = float(g_plain.abs().max())
    out["padded_vs_plain"] = {"max_abs_diff": float((g_pad - g_plain).abs().max()), "max_grad": scale, "loss_plain": l_plain, "loss_padded": l_pad,
                              "rel": float((g_pad - g_plain).abs().max()) / scale}
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    trainer = GraphedStep(model, opt, dev, t2s, lw, None)
    res = []
    for k in range(3):
        reset_bn()
        item = trainer.launch(trainer.prepare(data))
        torch.cuda.synchronize()
        raw = [float(t.reshape(-1)[0]) for t in item["loss_tuple"]]
        lt = trainer.finish(item)
        torch.cuda.synchronize()
        print("iter", k, "stats", trainer.stats, "raw loss tuple", raw[:4], "finished", lt is not None, flush=True)
        res.append((flat_grads(model), raw[0]))
    out["graph_stats"] = dict(trainer.stats)
    out["graph_vs_padded_eager_bitwise"] = [bool(torch.equal(r[0], g_pad)) for r in res]
    out["graph_vs_padded_max_abs_diff"] = [float((r[0] - g_pad).abs().max()) for r in res]
    out["graph_loss"] = [r[1] for r in res]
    # a different batch of the same shape through the same graph vs its own padded eager step
    data2 = batches[1]
    tg2 = loss_targets(data2, t2s, dev)
    prep2 = tf.prepare_batch(model, data2, dev, pad=True)
    g_pad2, _ = grads_of(prep2, tg2)
    reset_bn()
    item = trainer.prepare(data2)
    out["second_batch_same_key"] = item["key"] in trainer.graphs
    trainer.finish(trainer.launch(item))
    torch.cuda.synchronize()
    out["second_batch_bitwise"] = bool(torch.equal(flat_grads(model), g_pad2))
    out["second_batch_max_abs_diff"] = float((flat_grads(model) - g_pad2).abs().max())
    # ---- where a graphed step's time goes (same batch, dropout 0): host time of prepare / launch, GPU time of one replay
    cap = trainer.graphs[trainer.prepare(data)["key"]] if trainer.prepare(data)["key"] in trainer.graphs else None
    if cap is not None:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            cap.graph.replay()
        torch.cuda.synchronize()
        out["replay_only_ms"] = round((time.perf_counter() - t0) / 10 * 1e3, 2)
        # does a replay need the host, and what does the side-stream work of the next prepare() cost it?
        def timed(after):
            ts = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                cap.graph.replay()
                torch.cuda.current_stream().query()
                after()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            return round(float(np.median(ts)), 2)
        out["replay_then_sleep_10ms"] = timed(lambda: time.sleep(0.010))
        out["replay_then_prepare"] = timed(lambda: trainer.prepare(data))
        out["replay_then_prepare_x2"] = timed(lambda: (trainer.prepare(data), trainer.prepare(data)))
        # does switching between captured graphs cost anything?
        k1, k2 = trainer.prepare(data)["key"], trainer.prepare(data2)["key"]
        if k1 != k2:
            for _ in range(2):
                trainer.step(data2)
            ca, cb = trainer.graphs[k1], trainer.graphs[k2]
            seq, ts = "AAABBBABABAABB", []
            for ch in seq:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                (ca if ch == "A" else cb).graph.replay()
                torch.cuda.synchronize()
                ts.append(round((time.perf_counter() - t0) * 1e3, 1))
            out["alternating_replays"] = {"sequence": seq, "ms": ts}
        hp, hl, hf = [], [], []
        for _ in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            item = trainer.prepare(data)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            trainer.launch(item)
            t3 = time.perf_counter()
            trainer.finish(item)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            hp.append(t1 - t0); hl.append(t3 - t2); hf.append(t4 - t3)
        out["host_ms"] = {"prepare": round(np.median(hp) * 1e3, 2), "prepare_drain": round((t2 - t1) * 1e3, 2), "launch_enqueue": round(np.median(hl) * 1e3, 2),
                          "finish_incl_gpu_wait": round(np.median(hf) * 1e3, 2)}
    print(json.dumps(out), flush=True)


def timing(a, dev, batches, margs, t2s, lw):
    from confidence_bootstrapping_amd.utils import make_score_model, ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function, train_epoch
    # ---- 3: timing, shipped configuration
    times = {}
    if a.no_gc:
        import gc
        gc.disable()
    for mode in a.modes.split(","):
        model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False)
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
        loss_fn = partial(loss_function, **lw)
        train_epoch(model, [batches[k % 8] for k in range(12)], opt, dev, t2s, loss_fn, ema, hip_graph=(mode == "hip_graph"))
        train_epoch(model, [batches[k % 8] for k in range(12)], opt, dev, t2s, loss_fn, ema, hip_graph=(mode == "hip_graph"))
        torch.cuda.synchronize()
        blocks = []
        for rep in range(3):
            loader = [batches[(k + rep) % 8] for k in range(a.steps)]
            t0 = time.perf_counter()
            s = train_epoch(model, loader, opt, dev, t2s, loss_fn, ema, hip_graph=(mode == "hip_graph"))
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / a.steps * 1e3)
        times[mode] = {"ms_per_step_blocks": [round(b, 2) for b in blocks], "loss": s["loss"]}
        if mode == "hip_graph":
            from confidence_bootstrapping_amd.training import _GRAPHED
            times[mode]["stats"] = {k: v for k, v in _GRAPHED[model][1].stats.items() if k != "host_s"}
            hs = _GRAPHED[model][1].stats.get("host_s")
            if hs:
                times[mode]["host_trace_last_32_steps_prepare_finish_launch_ms"] = hs["trace"][-32:]
            times[mode]["graphs"] = len(_GRAPHED[model][1].graphs)
    print(json.dumps({"batch": a.batch, "workload": a.workload, "timing": times}), flush=True)


if __name__ == "__main__":
    main()
