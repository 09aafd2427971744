"""Benchmark of the hot path: reverse-diffusion docking sampler on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

One "step" = one complex of the named workload: 40 poses x 20 denoise steps (BASELINE.json configs[1],
synthetic DockGen-median complex: Nl=28, Nr=384, R=6), i.e. the time-independent receptor embedding +
20 x (score-model forward + reverse-SDE perturbation + pose update) for a batch of 40 poses.
Inputs (weights, complex, initial poses, pre-drawn noise) are resident in HBM before the timed region.
Multi-GPU: complexes are independent -> every rank runs K complexes of its own (weak scaling), no collective in
the data path; the only exchange is the final gather of poses to rank 0 (kept inside the timed region).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = tp_conv,
fp32 MFMA bound, duration from HIP events on the launch stream) and `cpu_baseline` (the oracle's PyTorch-CPU
restatement of the same path, timed on a bounded sample on rank 0 at N=1 only).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time
from functools import partial

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = "c2_dockgen_median"
SAMPLES, DENOISE_STEPS = 40, 20
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0     # MI355X_MICROARCH.md: bf16 dense (~2.5 PF)


def flops_per_edge(in_level: int, out_level: int) -> float:
    """Algorithmic FLOPs of one edge of a tensor-product layer: 2(F*H + H*W) + 2*sum_blocks fan*m_out*dim."""
    ns, nv = 32, 6
    n1o, n1e, n0o = (nv if in_level >= 1 else 0), (nv if in_level >= 2 else 0), (nv if in_level >= 3 else 0)
    fan0e, fan1o = ns + n1o, ns + n1o + n1e
    fan1e = n1o + n1e + n0o if out_level >= 2 else 0
    fan0o = n1e + n0o if out_level >= 3 else 0
    W = fan0e * ns + fan1o * nv + fan1e * nv + fan0o * nv
    tp = fan0e * ns * 1 + fan1o * nv * 3 + fan1e * nv * 3 + fan0o * nv * 1
    return 2.0 * (96 * 96 + 96 * W) + 2.0 * tp


def cpu_baseline(model, cplx, args, sched):
    """Oracle (PyTorch-CPU port of the reference arithmetic) on a bounded sample of the same workload."""
    from oracle import score_ref as sr, pose_ref as pr
    from tests.helpers import to_cx
    d = os.path.join(ROOT, "confidence_bootstrapping_amd", "data")
    so3, torus = np.load(os.path.join(d, "so3_exp_score_norms.npy")), np.load(os.path.join(d, "torus_score_norm.npy"))
    cx = to_cx(cplx)
    b, steps = 2, 4
    g = torch.Generator().manual_seed(0)
    pos = cplx["ligand"].pos[None].repeat(b, 1, 1) - cplx["ligand"].pos.mean(0) + cplx["receptor"].pos.mean(0) \
        + 10 * torch.randn(b, 1, 3, generator=g)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = sr.ScoreConfig()
    # time `steps` of the 20 schedule points spread over the schedule (the cross graph shrinks with t)
    idx = np.linspace(0, len(sched) - 1, steps).round().astype(int)

    def timed():
        p = pos.clone()
        t0 = time.perf_counter()
        rec_cache = sr.receptor_embedding(sd, cx, cfg)
        t_rec = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in idx:
            t = sched[i]
            out = sr.score_forward(sd, cx, p, t, t, t, cfg, so3, torus, rec_cache=rec_cache)
            p = pr.modify_conformer_batch(p, cx, 0.01 * out["tr_pred"], 0.01 * out["rot_pred"], 0.01 * out["tor_pred"])
        t_steps = time.perf_counter() - t0
        per_pose = (t_steps / steps / b) * DENOISE_STEPS + t_rec / SAMPLES   # receptor embedding amortised over the 40 poses
        return 1.0 / per_pose, t_steps + t_rec

    all_threads = torch.get_num_threads()
    v_all, t_all = timed()
    # the reference's own --restrict_cpu setting (inference.py:225-234) is 16 threads: report that figure too
    torch.set_num_threads(min(16, all_threads))
    try:
        v16, t16 = timed()
    finally:
        torch.set_num_threads(all_threads)
    n16 = min(16, all_threads)
    best, cores = (v16, n16) if v16 >= v_all else (v_all, all_threads)
    return {"value": round(best, 5), "unit": "poses/s", "cores": cores, "kind": "port",
            "by_threads": {str(all_threads): round(v_all, 5), str(n16): round(v16, 5)},
            "sample": f"{b} poses x {steps} of {DENOISE_STEPS} denoise steps (+ receptor embedding) of {WORKLOAD}, "
                      f"oracle PyTorch-CPU fp32, {t_all:.1f}s measured with {all_threads} threads and {t16:.1f}s with 16, "
                      f"extrapolated to 20 steps/pose"}


def hbm_secondary(st, eng, poses, elapsed):
    edge_visits = st["conv_edge_visits"] + 3 * st["ll_edges"]
    node_visits = poses * DENOISE_STEPS * (8 * eng.Nl + 4 * eng.engines[0].Nr)   # 3 + 5 ligand layers, 4 receptor layers
    nbytes = 432.0 * edge_visits + 592.0 * node_visits
    gbps = nbytes / elapsed / 1e9
    return {"algorithmic_mb_per_pose_step": round(nbytes / (poses * DENOISE_STEPS) / 1e6, 2), "achieved_gbps": round(gbps, 1),
            "peak_gbps": 8000.0, "frac": round(gbps / 8000.0, 4)}


def confidence_leg(cplx_seed, final_pos, dev):
    """All-atom confidence scoring of the 40 final poses of the last complex, measured OUTSIDE the timed
    region of the headline metric: ms per 40-pose batch and the fused conv kernel's algorithmic TFLOP/s (HIP events)."""
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_confidence_model
    from tools.conf_bench import flops_per_edge as cflops
    cmodel, cargs = make_confidence_model(device=dev, seed=5)
    ceng = cmodel.engine(max_batch=SAMPLES)
    ceng.set_complex(make_workload(WORKLOAD, seed=cplx_seed, all_atoms=True))
    for _ in range(2):
        ceng.score(final_pos, cargs.crop_beyond)
    counts = ceng.edge_counts()
    torch.cuda.synchronize()
    ceng.kernel_timing(enable=True, reset=True)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ceng.score(final_pos, cargs.crop_beyond, check=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    _, n, tot_ms = ceng.kernel_timing(enable=False)
    e_all, e_last = sum(counts.values()), counts["ll"] + counts["lr"] + counts["la"]
    fl = e_all * (cflops(0, 1) + cflops(1, 2) + cflops(2, 3) + cflops(3, 3)) + e_last * cflops(3, 3)
    tf = fl * reps / (tot_ms * 1e-3) / 1e12
    return {"what": "all-atom confidence model on the 40 final poses (crop 20 A, t=0), not part of `value`", "ms_per_40_poses": round(dt * 1e3, 3),
            "edges_per_layer": e_all, "kernel": "fctp_conv_kernel", "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4)}


def finetune_leg(dev, batch=8, warm=3, steps=6):
    """One confidence-bootstrapping fine-tuning step (BASELINE.json configs[4]) measured OUTSIDE the timed region of
    the headline metric: train-mode forward on the HIP tensor-product op, score-matching loss, HIP backward kernels, Adam, EMA on a
    batch of `batch` different C2-sized complexes noised by NoiseTransform (same code path as tools/train_bench.py)."""
    from confidence_bootstrapping_amd.synthetic import make_complex, WORKLOADS
    from confidence_bootstrapping_amd.utils import make_score_model, load_model_args, ExponentialMovingAverage
    from confidence_bootstrapping_amd.training import loss_function, train_step
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    margs = load_model_args()
    model, _ = make_score_model(device=dev, seed=0, args=margs, eval_mode=False)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    t2s = partial(t_to_sigma, args=margs)
    loss_fn = partial(loss_function, tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
    base = [make_complex(name=f"cplx{i}", seed=1234 + i, **WORKLOADS["c2_dockgen_median"]) for i in range(batch)]
    nt = NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False)
    state = (np.random.get_state(), torch.random.get_rng_state())
    np.random.seed(0)
    torch.manual_seed(0)
    batches = [[nt(copy.deepcopy(c)) for c in base] for _ in range(warm + steps)]
    np.random.set_state(state[0])
    torch.random.set_rng_state(state[1])
    for k in range(warm):
        train_step(model, batches[k], opt, dev, t2s, loss_fn, ema)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(warm, warm + steps):
        out = train_step(model, batches[k], opt, dev, t2s, loss_fn, ema)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"what": "fine-tuning step (train-mode forward + HIP backward kernels + Adam + EMA), not part of `value`", "batch": batch,
            "ms_per_step": round(dt * 1e3, 2), "complexes_per_s": round(batch / dt, 1), "loss": round(float(out[0]), 4), "dtype": "f32"}


def main():
    global WORKLOAD, SAMPLES, DENOISE_STEPS
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the secondary legs (confidence, other operand modes, fine-tuning, "
                    "CPU baseline): the command the rocprofv3 --pmc passes under profiles/ are taken over")
    ap.add_argument("--graph", type=int, default=0, help="1: replay the 20-step loop as one hipGraph (no per-kernel HIP events)")
    ap.add_argument("--streams", type=int, default=1, help="concurrent HIP streams the 40-pose batch is split over")
    ap.add_argument("--workload", default=WORKLOAD, help="synthetic complex: c2_dockgen_median (headline) or c4_large_pocket")
    ap.add_argument("--samples", type=int, default=SAMPLES)
    ap.add_argument("--denoise-steps", type=int, default=DENOISE_STEPS)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f32_split"],
                    help="bf16: FCBlock GEMMs on bf16 MFMA (configs[3]); f32_split: fp32 operands as three bf16 planes on the bf16 MFMA")
    ap.add_argument("--pair", type=int, default=4, help="co-schedule consecutive complexes (cbd_sample_multi: one tensor-product "
                    "launch covers the 40-pose batches of several complexes): 0 = one complex at a time, 1 = two, 2..4 = that many")
    a = ap.parse_args()
    WORKLOAD, SAMPLES, DENOISE_STEPS = a.workload, a.samples, a.denoise_steps

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_score_model
    from confidence_bootstrapping_amd.engine import DockEngine, DockEnginePool, make_steps
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
    from confidence_bootstrapping_amd.sampling import randomize_position
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.distributed import gather_poses

    model, margs = make_score_model(seed=0)
    cplx = make_workload(WORKLOAD, seed=1234)
    eng = DockEnginePool.from_model(model, dev, n=a.streams, max_batch=SAMPLES)
    eng.set_complex(cplx)
    eng.set_option("graph", a.graph)
    eng.set_option("bf16", int(a.dtype == "bf16"))
    if a.dtype == "f32_split":
        eng.set_option("f32_split", 1)
    cosched = (2 if a.pair == 1 else max(1, min(a.pair, 8))) if (a.pair and a.streams == 1 and not a.graph) else 1
    pair = cosched > 1
    extra = []   # further engines (own workspace, same device-resident weights) for the complexes that are co-scheduled
    for _ in range(min(cosched, 7) if pair else 0):      # one spare partner: a group may carry cosched + 1 complexes (see run())
        e2 = DockEnginePool.from_model(model, dev, n=1, max_batch=SAMPLES, share_from=eng)
        e2.set_complex(cplx)
        e2.set_option("bf16", int(a.dtype == "bf16"))
        if a.dtype == "f32_split":
            e2.set_option("f32_split", 1)
        extra.append(e2)
    eng2 = extra[0] if extra else None
    sched = get_t_schedule("expbeta", DENOISE_STEPS)
    steps = make_steps(sched, margs, model.timestep_emb_func)
    R = eng.R
    n_runs = a.warmup + a.steps
    # initial poses via the reference's randomisation, noise pre-drawn (seeded per (rank, complex)), all resident in HBM
    pos0, noise = [], []
    for k in range(n_runs):
        torch.manual_seed(42 + 1000 * rank + k)
        np.random.seed(42 + 1000 * rank + k)
        dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(SAMPLES)]
        randomize_position(dl, False, False, margs.tr_sigma_max)
        pos0.append(torch.stack([d["ligand"].pos for d in dl]).to(dev).contiguous())
        noise.append((torch.randn(DENOISE_STEPS, SAMPLES, 3).to(dev), torch.randn(DENOISE_STEPS, SAMPLES, 3).to(dev),
                      torch.randn(DENOISE_STEPS, SAMPLES * R).to(dev)))

    def one_complex(k):
        eng.recompute_receptor()
        eng.sample(pos0[k], steps, *noise[k])

    def run(lo, hi):
        """complexes lo..hi-1, two at a time when pairing is on (each complex still gets its own receptor embedding pass)"""
        k = lo
        while k < hi:
            if pair and k + 1 < hi:
                left = hi - k
                # no complex runs alone: one more than the nominal group rides along (5 -> 5), otherwise balanced groups (9 -> 3 x 3)
                m = left if left <= min(cosched + 1, 8) else -(-left // -(-left // cosched))
                pools = [eng] + extra[:m - 1]
                for p_ in pools:
                    p_.recompute_receptor()
                DockEngine.sample_multi([p_.engines[0] for p_ in pools], [pos0[k + q] for q in range(m)], steps,
                                        [noise[k + q] for q in range(m)])
                k += m
            else:
                one_complex(k)
                k += 1

    run(0, a.warmup)
    torch.cuda.synchronize()
    alt_k = list(range(max(a.warmup, n_runs - 2 * cosched), n_runs))     # complexes re-run in the other operand modes afterwards
    alt_init = {k: pos0[k].clone() for k in alt_k}
    eng.kernel_timing(enable=not a.graph, reset=True)
    eng.stats(reset=True)
    for e2 in extra:
        e2.kernel_timing(enable=True, reset=True)
        e2.stats(reset=True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.warmup, n_runs)
    final = gather_poses(pos0[n_runs - 1], world, rank)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    avg_ms, n_launch, total_ms = eng.kernel_timing(enable=False)
    st = eng.stats()
    for e2 in extra:   # merged launches are timed by whichever engine arrived last at the rendezvous
        _, n2, t2 = e2.kernel_timing(enable=False)
        n_launch, total_ms = n_launch + n2, total_ms + t2
        avg_ms = total_ms / max(n_launch, 1)
        st = {k: st[k] + v for k, v in e2.stats().items()}
    assert torch.isfinite(pos0[n_runs - 1]).all(), "non-finite poses"

    if rank == 0:
        poses = SAMPLES * a.steps * world
        f33 = flops_per_edge(3, 3)
        femb = flops_per_edge(0, 1) + flops_per_edge(1, 2) + flops_per_edge(2, 3)
        total_flops = st["conv_edge_visits"] * f33 + st["ll_edges"] * femb      # algorithmic work of the timed tp_conv launches
        flops_per_launch = total_flops / max(n_launch, 1)
        traffic = None   # HBM bytes per tp_conv<3,3> launch from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE), see profiles/
        tp = next((q for q in (os.path.join(ROOT, "profiles", f"r01_{t}_traffic.json") for t in "mlhf") if os.path.exists(q)), "")   # same command (defaults), mean over ALL tp_conv launches like `achieved`
        if os.path.exists(tp):
            traffic = round(json.load(open(tp))["hbm_bytes_per_launch_all_tp_conv"])
        achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        algorithmic = achieved
        if a.dtype == "f32_split":
            achieved *= 6.0     # every fp32 product is issued as 6 bf16 plane products; price the ISSUED flops against the bf16 peak
        headline = (WORKLOAD, SAMPLES, DENOISE_STEPS, a.dtype) == ("c2_dockgen_median", 40, 20, "f32")
        peak = PEAK_FP32_MFMA_TFLOPS if a.dtype == "f32" else PEAK_BF16_MFMA_TFLOPS
        if not headline:
            traffic = None   # the PMC traffic figure under profiles/ belongs to the headline configuration
        out = {
            "metric": "poses/sec (whole node), 40-sample x 20-step diffusion, DockGen median complex" if headline else
                      f"poses/sec (whole node), {SAMPLES}-sample x {DENOISE_STEPS}-step diffusion, {WORKLOAD}",
            "value": round(poses / elapsed, 3), "unit": "poses/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": WORKLOAD, "samples_per_complex": SAMPLES, "denoise_steps": DENOISE_STEPS, "streams": a.streams,
                       "co_scheduled_complexes": cosched,
                       "Nl": eng.Nl, "Nr": eng.engines[0].Nr, "R": eng.R, "weights": "random-init, reference state_dict layout",
                       "sharding": f"{world} rank(s) x {a.steps} complexes each, no data-path collective"},
            "roofline": {"bound": "mfma", "kernel": {"f32": "tp_conv_kernel<OpsF32>", "bf16": "tp_conv_kernel<OpsBf16>", "f32_split": "tp_conv_kernel<OpsBf16x3>"}[a.dtype],
                         "achieved": round(achieved, 3), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                         "avg_launch_ms": round(avg_ms, 4), "launches": n_launch,
                         "algorithmic_gflop_per_launch": round(flops_per_launch / 1e9, 3),
                         "algorithmic_tflops": round(algorithmic, 3),
                         "tp_conv_share_of_wall": round(total_ms * 1e-3 / elapsed, 4),
                         "pose_steps_per_s": round(poses * DENOISE_STEPS / elapsed, 1),
                         # secondary: fused-ideal algorithmic bytes = 432 B per edge-layer visit + 592 B per
                         # node-layer visit, against the 8 TB/s HBM3E peak -- the path is far from HBM-bound
                         "hbm_secondary": hbm_secondary(st, eng, poses, elapsed)},
        }
        extras = world == 1 and headline and not a.headline_only
        if extras:
            out["confidence"] = confidence_leg(1234, pos0[n_runs - 1], dev)
        if extras and pair:
            # The same complexes in the two other operand modes of the same kernel (NOT part of `value`): f32_split = fp32 operands as
            # three exact bf16 planes on the bf16 matrix cores (fp32-grade results, tests/test_gpu_bf16.py); bf16 = configs[3].
            out["other_operand_modes"] = {}
            for mode in ("f32_split", "bf16"):
                for p_ in [eng] + extra:
                    p_.set_option("bf16", int(mode == "bf16"))
                    p_.set_option("f32_split", int(mode == "f32_split"))
                for timed in (False, True):
                    for k in alt_k:
                        pos0[k].copy_(alt_init[k])
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    run(alt_k[0], n_runs)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter() - t1
                out["other_operand_modes"][mode] = {"value": round(SAMPLES * len(alt_k) / t1, 1), "unit": "poses/s", "complexes": len(alt_k)}
            for p_ in [eng] + extra:
                p_.set_option("bf16", 0)
                p_.set_option("f32_split", 0)
        if extras:
            try:
                out["finetune"] = finetune_leg(dev)
            except Exception as e:      # a secondary leg must never cost the headline line
                out["finetune"] = {"error": repr(e)[:200]}
        if extras and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, cplx, margs, sched)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
