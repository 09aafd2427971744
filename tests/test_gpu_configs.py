"""Every configuration of BASELINE.json at its stated size on one MI355X (configs[0] is covered by test_gpu_parity.py::test_c1_*,
configs[3] by test_gpu_bf16.py::test_config_c4_bf16_as_specified):

  configs[1]  the HEADLINE workload of bench.py itself -- globular receptor, pocket at 0.7 R, poses on the ideal reverse path, the
              translation head scaled by 0.02 -- against the fp32 oracle (scores at two times, a 3-step trajectory prefix), then the full
              40 x 20 run through the bench's own code path (four complexes per hipGraph launch) with its size-independent properties;
  configs[2]  the full 189-complex heterogeneous set x 40 samples x 20 steps through distributed.run_complex_set (world 1) with
              confidence ranking, three complexes re-checked against the oracle, properties on all, bitwise repeat of the whole set;
  configs[4]  one confidence-bootstrapping round on a cluster of C2-sized complexes with inference_samples = 8 and 20 steps:
              inference_epoch -> CBBuffer -> train_epoch (reference finetune_train.py:133-249, utils/training.py:184-233).

fp32 tolerances (stated): scores rel <= 2e-5 of the largest component (as test_gpu_parity.py), trajectory RMSD <= 1e-3 A (north star).
"""
import copy
from argparse import Namespace
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bond_lengths(cplx, pos):
    ei = cplx["ligand", "ligand"].edge_index
    d0 = (cplx["ligand"].pos[ei[0]] - cplx["ligand"].pos[ei[1]]).norm(dim=-1)
    d1 = (pos[:, ei[0]] - pos[:, ei[1]]).norm(dim=-1)
    return float((d1 - d0[None]).abs().max())


def test_config_c2_headline_workload_vs_oracle(tables):
    from confidence_bootstrapping_amd.synthetic import make_workload, scale_tr_head, ideal_path_inputs, BENCH_GEOMETRY
    from confidence_bootstrapping_amd.utils import make_score_model
    from confidence_bootstrapping_amd.engine import DockEngine, make_steps
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
    from oracle import score_ref as sr, pose_ref as pr
    from tests.helpers import to_cx
    so3, torus = tables
    dev = torch.device("cuda:0")
    model, args = make_score_model(seed=0)
    scale_tr_head(model)                                     # bench.py --poses ideal
    cplx = make_workload("c2_dockgen_median", seed=1234, **BENCH_GEOMETRY)
    cx = to_cx(cplx)
    B, S = 40, 20
    sched = get_t_schedule("expbeta", S)
    steps = make_steps(sched, args, model.timestep_emb_func)
    pos0, z_tr, z_rot, z_tor = ideal_path_inputs(cplx, args, sched, B, seed=42)      # bench.py's complex 0 on rank 0
    R = int(cplx["ligand"].edge_mask.sum())
    pocket = cplx["ligand"].pos.mean(0)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = sr.ScoreConfig()
    rec_cache = sr.receptor_embedding(sd, cx, cfg)
    eng = DockEngine.from_model(model, dev, max_batch=B)
    eng.set_complex(cplx)
    # ---- scores of the whole batch at two diffusion times; oracle on poses 3 and 28, placed where the ideal path has them at that time
    pick = [3, 28]
    eps = (pos0.mean(1, keepdim=True) - pocket) / args.tr_sigma_max
    for t in (1.0, 0.3):
        sig = args.tr_sigma_min ** (1 - t) * args.tr_sigma_max ** t
        p = pos0 - pos0.mean(1, keepdim=True) + pocket + sig * eps
        step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
        tr, rot, tor = [x.cpu() for x in eng.score(p.to(dev), step)]
        ref = sr.score_forward(sd, cx, p[pick], t, t, t, cfg, so3, torus, rec_cache=rec_cache)
        for got, want in ((tr[pick], ref["tr_pred"]), (rot[pick], ref["rot_pred"]), (tor.reshape(B, R)[pick].reshape(-1), ref["tor_pred"])):
            err, scale = float((got - want).abs().max()), float(want.abs().max())
            assert err <= 2e-5 * scale, (t, err, scale)
        counts = eng.edge_counts()
        assert counts["rr"] == B * 24 * 384 and counts["lr"] == counts["rl"] and counts["lr"] > 0
    # ---- the first three steps of the bench trajectory of two poses against the oracle's trajectory on the same noise
    sel = torch.tensor(pick)
    cols = (sel[:, None] * R + torch.arange(R)[None]).reshape(-1)
    n4 = {"tr": z_tr[:4, sel], "rot": z_rot[:4, sel], "tor": z_tor[:4, cols]}
    _, trace = pr.sampling_ref(sd, cx, pos0[sel], sched[:4], cfg, so3, torus, noise=n4, record=True)
    p3 = pos0[sel].to(dev).contiguous()
    eng.sample(p3, (type(steps[0]) * 3)(*[steps[i] for i in range(3)]), n4["tr"][:3], n4["rot"][:3], n4["tor"][:3])
    r3 = float(torch.sqrt(((p3.cpu() - trace[2]["pos"]) ** 2).sum(-1).mean(-1)).max())
    assert r3 < 1e-3, r3
    # ---- the full 40 x 20 run the way bench.py runs it: four complexes (here: the same complex, four different seeds) per hipGraph
    #      launch; = the eager single-engine run, bitwise; stays on the ideal path; the blueprint's work (Elr ~ 6 200)
    engs = [eng]
    for _ in range(3):
        e = DockEngine(dev, max_batch=B)
        e.share_weights_from(eng)
        e.set_complex(cplx)
        engs.append(e)
    inputs = [(pos0, z_tr, z_rot, z_tor)] + [ideal_path_inputs(cplx, args, sched, B, seed=43 + k) for k in range(3)]
    outs = []
    for graph in (1, 1, 0):
        for e in engs:
            e.set_option("graph", graph)
            e.recompute_receptor()
            e.stats(reset=True)
        ps = [x[0].to(dev).contiguous() for x in inputs]
        DockEngine.sample_multi(engs, ps, steps, [[z.to(dev) for z in x[1:]] for x in inputs])
        torch.cuda.synchronize()
        outs.append(ps)
    for e in engs:
        e.set_option("graph", 0)
    st = {k: sum(e.stats()[k] for e in engs) for k in ("ll_edges", "conv_edge_visits")}
    pose_steps = 4 * B * S
    elr_mean = (st["conv_edge_visits"] - 5 * st["ll_edges"] - 4 * pose_steps * 24 * 384) / 9.0 / pose_steps
    assert 5800 < elr_mean < 6800, elr_mean                    # the work behind bench.py's `value`
    for a, b, c in zip(*outs):
        assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, c)   # replayed graph = first launch = eager
    single = pos0.to(dev).contiguous()
    eng.sample(single, steps, z_tr, z_rot, z_tor)
    assert torch.equal(single, outs[0][0])                      # co-scheduled = one complex at a time
    final = outs[0][0].cpu()
    assert float((final.mean(1) - pocket).norm(dim=1).max()) < 1.0      # on the ideal path: centroid within 1 A of the pocket
    assert _bond_lengths(cplx, final) < 1e-3


def test_config_c3_full_complex_set(tables):
    from confidence_bootstrapping_amd.synthetic import complex_set_sizes, make_set_complex
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model
    from confidence_bootstrapping_amd.engine import make_steps
    from confidence_bootstrapping_amd.distributed import run_complex_set, shard_lpt
    from confidence_bootstrapping_amd.complex_set import ComplexSetRunner
    from oracle import score_ref as sr, pose_ref as pr
    from tests.helpers import to_cx
    so3, torus = tables
    dev = torch.device("cuda:0")
    N, B, S = 189, 40, 20
    sizes = complex_set_sizes(N, seed=7)
    smodel, sargs = make_score_model(device=dev, seed=0)
    cmodel, cargs = make_confidence_model(device=dev, seed=5)
    runner = ComplexSetRunner(smodel, sargs, cmodel, cargs, dev, samples=B, denoise_steps=S, group=4, keep_poses=True)
    cps = [make_set_complex(i, sizes[i], seed=7) for i in range(N)]
    for i, c in enumerate(cps):
        runner.prepare(i, c)
    res = run_complex_set(cps, runner.sample_group, world=1, rank=0, group=4)
    assert [r["complex"] for r in res] == list(range(N))
    for i, r in enumerate(res):
        allp = r["all_pos"]
        assert allp.shape == (B, sizes[i][0], 3) and torch.isfinite(allp).all(), i
        assert _bond_lengths(cps[i], allp) < 2e-3, i
        assert np.isfinite(r["confidence"]) and sorted(r["order"].tolist()) == list(range(B))
    # ---- three complexes drawn at random (seeded): engine vs oracle -- scores of two initial poses at t = 1 and the first two steps
    sd = {k: v.detach().cpu() for k, v in smodel.state_dict().items()}
    cfg = sr.ScoreConfig()
    rng = np.random.default_rng(3)
    eng = runner.engines[0]
    for i in rng.choice(N, size=3, replace=False).tolist():
        cplx, pos0, noise = runner.prepared[i]
        cx = to_cx(cplx)
        R = int(cplx["ligand"].edge_mask.sum())
        eng.set_complex(cplx)
        pick = [1, 17]
        step = make_steps(np.array([1.0]), sargs, smodel.timestep_emb_func)[0]
        tr, rot, tor = [x.cpu() for x in eng.score(pos0.to(dev), step)]
        ref = sr.score_forward(sd, cx, pos0[pick], 1.0, 1.0, 1.0, cfg, so3, torus)
        pairs = [(tr[pick], ref["tr_pred"]), (rot[pick], ref["rot_pred"])]
        if R > 0:
            pairs.append((tor.reshape(B, R)[pick].reshape(-1), ref["tor_pred"]))
        for got, want in pairs:
            err, scale = float((got - want).abs().max()), float(want.abs().max())
            assert err <= 2e-5 * scale, (i, err, scale)
        sel = torch.tensor(pick)
        cols = (sel[:, None] * R + torch.arange(R)[None]).reshape(-1)
        n3 = {"tr": noise[0][:3, sel], "rot": noise[1][:3, sel], "tor": noise[2][:3, cols] if R > 0 else None}
        _, trace = pr.sampling_ref(sd, cx, pos0[sel], runner.sched[:3], cfg, so3, torus, noise=n3, record=True)
        p2 = pos0[sel].to(dev).contiguous()
        eng.sample(p2, (type(runner.steps[0]) * 2)(runner.steps[0], runner.steps[1]), n3["tr"][:2], n3["rot"][:2], n3["tor"][:2] if R > 0 else None)
        r2 = float(torch.sqrt(((p2.cpu() - trace[1]["pos"]) ** 2).sum(-1).mean(-1)).max())
        assert r2 < 1e-3, (i, r2)
    # ---- the whole set a second time: bitwise the same poses, confidences and ranking (no atomics anywhere on the path)
    res2 = run_complex_set(cps, runner.sample_group, world=1, rank=0, group=4)
    for a, b in zip(res, res2):
        assert torch.equal(a["all_pos"], b["all_pos"]) and a["confidence"] == b["confidence"] and np.array_equal(a["order"], b["order"])
    # ---- the 8-GPU partition of this set (the driver's node): every complex exactly once, LPT loads within 6 % of each other
    parts = shard_lpt([nl * nr for nl, nr, _ in sizes], 8)
    assert sorted(sum(parts, [])) == list(range(N))
    loads = [sum(sizes[i][0] * sizes[i][1] for i in p) for p in parts]
    assert max(loads) / min(loads) < 1.06


def test_config_c5_bootstrapping_round_at_size():
    """configs[4] on one GPU: a cluster of six C2-sized complexes (Nl 28, Nr 384, R 6, all-atom stores), inference_samples = 8,
    20 denoising steps, inference batch 4 (README), training batch 5: inference_epoch (sampling + confidence + symmetry-corrected RMSD)
    -> CBBuffer -> one train_epoch on the HIP training path -> the engine samples with the updated weights."""
    from confidence_bootstrapping_amd.synthetic import make_complex, add_atoms, WORKLOADS
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model, load_model_args, ExponentialMovingAverage
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    from confidence_bootstrapping_amd.bootstrapping.buffer import CBBuffer
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.finetune_train import inference_epoch, _Loader
    from confidence_bootstrapping_amd.training import loss_function, train_epoch
    dev = torch.device("cuda:0")
    margs = load_model_args()
    model, _ = make_score_model(device=dev, seed=0, args=margs)
    conf_model, conf_args = make_confidence_model(device=dev, seed=5)
    names = [f"{i}abc_A_l{i}" for i in range(6)]
    targets = []
    for i, n in enumerate(names):
        g = add_atoms(make_complex(seed=500 + i, name=n, **WORKLOADS["c2_dockgen_median"]), seed=500 + i)
        g["ligand"].orig_pos = g["ligand"].pos.numpy() + g.original_center.numpy()
        nums = np.minimum(g["ligand"].x[:, 0].numpy() + 1, 118)     # synthetic atom types as "atomic numbers" (0 would be filtered as H)
        g["ligand"].x[:, 0] = torch.from_numpy(nums)
        ei = g["ligand", "ligand"].edge_index.numpy()
        am = np.zeros((len(nums), len(nums)), dtype=int)
        am[ei[0], ei[1]] = 1
        g.mol = Namespace(atomicnums=nums, adjacency_matrix=am)
        targets.append(g)
    args = copy.copy(margs)
    args.__dict__.update(inference_steps=20, inference_samples=8, inference_batch_size=4, batch_size=5)
    t2s = partial(t_to_sigma, args=margs)
    torch.manual_seed(0); np.random.seed(0)
    metrics, kept, top = inference_epoch(model, conf_model, targets, None, dev, t2s, args, conf_args, confidence_cutoff=-1e9)
    assert len(kept) == 6 * 8 and len(top) == 6 and np.isfinite(metrics["avg_confidence"]) and metrics["rmsds_lt5"] is not None
    for g, c in kept:
        p = g["ligand"].pos
        assert p.shape == (28, 3) and torch.isfinite(p).all() and np.isfinite(c)
    # same seeds -> the same poses and confidences (the sampler and the confidence engine are deterministic)
    torch.manual_seed(0); np.random.seed(0)
    metrics2, kept2, _ = inference_epoch(model, conf_model, targets, None, dev, t2s, args, conf_args, confidence_cutoff=-1e9)
    assert metrics2 == metrics and all(torch.equal(a[0]["ligand"].pos, b[0]["ligand"].pos) and a[1] == b[1] for a, b in zip(kept, kept2))
    buf = CBBuffer(cluster_name="c", cluster_to_ligands={"c": names}, max_complexes_per_couple=20,
                   transform=NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False))
    buf.add_complexes(kept)
    assert len(buf.complexes) == 48
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    w0 = model.conv_layers[0].fc[0][3].weight.detach().clone()
    loss_fn = partial(loss_function, tr_weight=0.33, rot_weight=0.33, tor_weight=0.33, no_torsion=False)
    losses = train_epoch(model, _Loader(buf, 5, shuffle=True), opt, dev, t2s, loss_fn, ema)
    assert np.isfinite(losses["loss"]) and ema.num_updates == 10            # ceil(48 / 5) steps
    assert not torch.equal(w0, model.conv_layers[0].fc[0][3].weight)
    model.eval()
    m3, kept3, _ = inference_epoch(model, conf_model, targets[:2], None, dev, t2s, args, conf_args, confidence_cutoff=-1e9)
    assert len(kept3) == 16 and np.isfinite(m3["avg_confidence"])
