"""One confidence-bootstrapping round trip on the MI355X (reference finetune_train.py:248-349): sample with the fused engine ->
confidence model -> symmetry-corrected RMSD metrics -> CBBuffer -> NoiseTransform -> train_epoch on the HIP training path -> EMA ->
sample again with the updated weights."""
import copy
from argparse import Namespace
from functools import partial

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_confidence_bootstrapping_round_trip():
    from confidence_bootstrapping_amd.synthetic import make_complex, add_atoms
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model, load_model_args, ExponentialMovingAverage
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    from confidence_bootstrapping_amd.bootstrapping.buffer import CBBuffer
    from confidence_bootstrapping_amd.datasets.pdbbind import NoiseTransform
    from confidence_bootstrapping_amd.finetune_train import inference_finetune, inference_epoch
    dev = torch.device("cuda:0")
    margs = load_model_args()
    model, _ = make_score_model(device=dev, seed=0, args=margs)
    conf_model, conf_args = make_confidence_model(device=dev, seed=5)
    names = ["1abc_A_l0", "2xyz_B_l1", "3pqr_C_l2"]
    targets = []
    for i, n in enumerate(names):
        g = add_atoms(make_complex(Nl=9 + i, Nr=36 + 4 * i, R=1 + i % 2, knn=8, seed=40 + i, name=n), seed=40 + i)
        g["ligand"].orig_pos = g["ligand"].pos.numpy() + g.original_center.numpy()
        nums = np.minimum(g["ligand"].x[:, 0].numpy() + 1, 118)     # synthetic atom types as "atomic numbers" (0 would be filtered as H)
        g["ligand"].x[:, 0] = torch.from_numpy(nums)
        ei = g["ligand", "ligand"].edge_index.numpy()
        am = np.zeros((len(nums), len(nums)), dtype=int)
        am[ei[0], ei[1]] = 1
        g.mol = Namespace(atomicnums=nums, adjacency_matrix=am)
        targets.append(g)
    args = copy.copy(margs)
    args.__dict__.update(inference_steps=4, inference_samples=4, inference_batch_size=4, n_epochs=2, cb_inference_freq=1,
                         initial_iterations=1, inference_iterations=1, num_inference_complexes=3, batch_size=4, use_ema=True,
                         tr_weight=0.33, rot_weight=0.33, tor_weight=0.33)
    t2s = partial(t_to_sigma, args=margs)
    torch.manual_seed(0); np.random.seed(0)
    metrics, kept, top = inference_epoch(model, conf_model, targets, None, dev, t2s, args, conf_args, confidence_cutoff=-1e9)
    assert len(kept) == 12 and len(top) == 3 and metrics["rmsds_lt5"] is not None and np.isfinite(metrics["avg_confidence"])
    buf = CBBuffer(cluster_name="c", cluster_to_ligands={"c": names}, max_complexes_per_couple=6,
                   transform=NoiseTransform(t_to_sigma=t2s, no_torsion=False, all_atom=False))
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    ema = ExponentialMovingAverage(model.parameters(), decay=0.999)
    w0 = model.conv_layers[0].fc[0][3].weight.detach().clone()
    hist = inference_finetune(args, model, conf_model, conf_args, None, -1e9, opt, ema, buf, targets, t2s, dev, log=lambda s: None)
    assert len(hist) == 2 and all(np.isfinite(h["train_loss"]) for h in hist)
    assert hist[0]["buffer"] == 12 and hist[1]["buffer"] == 18          # 3 couples x top-6 after the second round
    assert not torch.equal(w0, model.conv_layers[0].fc[0][3].weight)    # the fine-tuning steps moved the weights
    assert ema.num_updates == 3 + 5
    # the inference engine picks the updated weights up
    model.eval()
    m2, kept2, _ = inference_epoch(model, conf_model, targets[:1], None, dev, t2s, args, conf_args, confidence_cutoff=-1e9)
    assert len(kept2) == 4 and np.isfinite(m2["avg_confidence"])


def test_inference_epoch_fix_equals_inference_epoch_without_a_confidence_model(monkeypatch):
    """utils/training.py:292-373 (validation by docking during training) against inference_epoch with filtering_model=None on the same
    seeds: same poses, so the same RMSD percentages; a complex whose sampling keeps failing counts as 100 A."""
    from confidence_bootstrapping_amd.synthetic import make_complex, add_atoms
    from confidence_bootstrapping_amd.utils import make_score_model, load_model_args
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma
    from confidence_bootstrapping_amd.finetune_train import inference_epoch
    from confidence_bootstrapping_amd.training import inference_epoch_fix
    dev = torch.device("cuda:0")
    margs = load_model_args()
    model, _ = make_score_model(device=dev, seed=0, args=margs)
    targets = []
    for i in range(2):
        g = make_complex(Nl=9 + i, Nr=36 + 4 * i, R=1 + i, knn=8, seed=50 + i, name=f"c{i}")
        g["ligand"].orig_pos = g["ligand"].pos.numpy() + g.original_center.numpy()
        nums = np.minimum(g["ligand"].x[:, 0].numpy() + 1, 118)
        g["ligand"].x[:, 0] = torch.from_numpy(nums)
        ei = g["ligand", "ligand"].edge_index.numpy()
        am = np.zeros((len(nums), len(nums)), dtype=int)
        am[ei[0], ei[1]] = 1
        g.mol = Namespace(atomicnums=nums, adjacency_matrix=am)
        targets.append(g)
    args = copy.copy(margs)
    args.__dict__.update(inference_steps=4, inference_samples=4, inference_batch_size=4, inf_pocket_knowledge=False, inf_pocket_cutoff=7)
    t2s = partial(t_to_sigma, args=margs)
    torch.manual_seed(7); np.random.seed(7)
    a = inference_epoch_fix(model, targets, dev, t2s, args)
    assert set(a) == {"rmsds_lt2", "rmsds_lt5", "min_rmsds_lt2", "min_rmsds_lt5"} and all(0 <= v <= 100 for v in a.values())
    torch.manual_seed(7); np.random.seed(7)
    b, kept, _ = inference_epoch(model, None, [targets[0]], None, dev, t2s, args, None, confidence_cutoff=-1e9)
    torch.manual_seed(7); np.random.seed(7)
    a0 = inference_epoch_fix(model, [targets[0]], dev, t2s, args)
    assert kept == [] and all(a0[k] == pytest.approx(b[k]) for k in a0)
    # failure protocol: a model whose sampling raises -> every pose of the complex counts as 100 A
    import confidence_bootstrapping_amd.sampling as smp
    calls = []

    def broken(**kw):
        calls.append(1)
        raise RuntimeError("no convergence")
    monkeypatch.setattr(smp, "sampling", broken)
    out = inference_epoch_fix(model, [targets[0]], dev, t2s, args)
    assert len(calls) == 6
    assert out == {"rmsds_lt2": 0.0, "rmsds_lt5": 0.0, "min_rmsds_lt2": 0.0, "min_rmsds_lt5": 0.0}


def test_inference_epoch_with_an_asyncronous_noise_schedule():
    """finetune_train.py:137-140,185-186: with args.asyncronous_noise_schedule the three component schedules are the Beta quantiles of the
    common time grid and sampling() gets the grid itself for the model's time embedding.  The epoch's sampling call must equal a direct
    sampling() call with those schedules (same seeds), and differ from the synchronous one."""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_complex, add_atoms
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model, load_model_args
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma, get_t_schedule, get_inverse_schedule
    from confidence_bootstrapping_amd.finetune_train import inference_epoch
    from confidence_bootstrapping_amd.sampling import sampling, randomize_position
    dev = torch.device("cuda:0")
    margs = load_model_args()
    margs.asyncronous_noise_schedule = True
    margs.sampling_alpha, margs.sampling_beta, margs.rot_alpha, margs.rot_beta, margs.tor_alpha, margs.tor_beta = 1.0, 1.0, 2.0, 1.0, 1.0, 2.0
    model, _ = make_score_model(device=dev, seed=0, args=margs)
    assert model.asyncronous_noise_schedule
    conf_model, conf_args = make_confidence_model(device=dev, seed=5)
    g = add_atoms(make_complex(Nl=10, Nr=40, R=2, knn=8, seed=41, name="1abc_A_l0"), seed=41)
    args = copy.copy(margs)
    args.__dict__.update(inference_steps=5, inference_samples=4, inference_batch_size=4)
    t2s = partial(t_to_sigma, args=margs)
    model.eval()
    torch.manual_seed(3); np.random.seed(3)
    metrics, kept, _ = inference_epoch(model, conf_model, [g], None, dev, t2s, args, conf_args, confidence_cutoff=-1e9)
    assert len(kept) == 4 and np.isfinite(metrics["avg_confidence"])
    pos_epoch = np.stack([k[0]["ligand"].pos.cpu().numpy() for k in kept])
    conf_epoch = np.array([float(k[1]) for k in kept])

    t = get_t_schedule("expbeta", 5)
    sched = dict(tr_schedule=get_inverse_schedule(t, 1.0, 1.0), rot_schedule=get_inverse_schedule(t, 2.0, 1.0), tor_schedule=get_inverse_schedule(t, 1.0, 2.0))
    assert np.allclose(sched["tr_schedule"], t) and not np.allclose(sched["rot_schedule"], t)

    def direct(**kw):
        torch.manual_seed(3); np.random.seed(3)
        dl = [Batch.from_data_list([copy.deepcopy(g)]) for _ in range(4)]
        randomize_position(dl, args.no_torsion, False, args.tr_sigma_max)
        out, conf = sampling(data_list=dl, model=model, inference_steps=5, device=dev, t_to_sigma=t2s, model_args=args, confidence_model=conf_model,
                             filtering_model_args=conf_args, batch_size=4, asyncronous_noise_schedule=True, t_schedule=t, **kw)
        return np.stack([d["ligand"].pos.cpu().numpy() for d in out]), conf.cpu().numpy().reshape(-1)

    pos, conf = direct(**sched)
    order = np.argsort(-conf_epoch, kind="stable")
    assert np.array_equal(np.sort(conf), np.sort(conf_epoch))
    for k in range(4):      # same poses (the epoch keeps them with their confidences; order may be the ranked one)
        assert any(np.array_equal(pos[k], pe) for pe in pos_epoch)
    pos_sync, _ = direct(tr_schedule=t, rot_schedule=t, tor_schedule=t)
    assert not np.allclose(pos_sync, pos, atol=1e-3)


def test_sampling_co_schedules_complexes_identically():
    """sampling() over the poses of six different complexes: any co-scheduling (cbd_sample_multi with 6 or 4 engines) gives bitwise the
    poses and confidences of one complex at a time."""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_complex, add_atoms
    from confidence_bootstrapping_amd.utils import make_score_model, make_confidence_model
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from confidence_bootstrapping_amd.sampling import sampling, randomize_position
    dev = torch.device("cuda:0")
    model, margs = make_score_model(device=dev, seed=0)
    conf_model, conf_args = make_confidence_model(device=dev, seed=5)
    cps = [add_atoms(make_complex(Nl=9 + i, Nr=36 + 4 * i, R=1 + i % 2, knn=8, seed=40 + i, name=f"c{i}"), seed=40 + i) for i in range(6)]
    torch.manual_seed(1); np.random.seed(1)
    base = [Batch.from_data_list([copy.deepcopy(c)]) for c in cps for _ in range(2)]
    randomize_position(base, False, False, margs.tr_sigma_max)
    sched = get_t_schedule("expbeta", 5)
    outs = []
    for co in (None, 4, 1):     # None: sampling() picks the group size itself (here 6 complexes of 2 poses -> one 6-engine launch group)
        dl = [copy.deepcopy(d) for d in base]
        torch.manual_seed(7)
        res, conf = sampling(data_list=dl, model=model, inference_steps=5, tr_schedule=sched, rot_schedule=sched, tor_schedule=sched,
                             device=dev, t_to_sigma=partial(t_to_sigma, args=margs), model_args=margs, confidence_model=conf_model,
                             filtering_model_args=conf_args, batch_size=2, co_schedule=co)
        outs.append(([d["ligand"].pos.clone() for d in res], conf.clone()))
    for other in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0][0], other[0]))
        assert torch.equal(outs[0][1], other[1]) and other[1].shape == (12,)


def test_pipelined_set_up_of_the_next_wave_gives_the_poses_of_synchronous_engines():
    """sampling() sets the complexes of wave k + 1 up while wave k runs (two alternating engine sets, cbd_set_complex with "async_setup" on a
    stream of its own, uploads on a side stream).  Seven different complexes in waves of two (engines re-used for other complexes while
    the other set is busy, a last wave of one) must give bitwise the poses of one fresh, synchronous engine per complex -- twice in a row
    (the second call finds the engines holding the LAST complexes of the first)."""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_complex
    from confidence_bootstrapping_amd.utils import make_score_model
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from confidence_bootstrapping_amd.engine import DockEngine, make_steps, _single_complex
    from confidence_bootstrapping_amd.sampling import sampling, randomize_position, draw_noise_like_reference
    dev = torch.device("cuda:0")
    model, margs = make_score_model(device=dev, seed=0)
    S, per, R = 6, 3, 2
    cps = [make_complex(Nl=8 + 2 * i, Nr=30 + 7 * i, R=R, knn=8, seed=60 + i, name=f"p{i}") for i in range(7)]
    torch.manual_seed(2); np.random.seed(2)
    base = [Batch.from_data_list([copy.deepcopy(c)]) for c in cps for _ in range(per)]
    randomize_position(base, False, False, margs.tr_sigma_max)
    sched = get_t_schedule("expbeta", S)
    torch.manual_seed(5)
    noise = draw_noise_like_reference(per * len(cps), R, S, per)
    steps = make_steps(sched, margs, model.timestep_emb_func)
    want = []
    for i in range(len(cps)):       # reference: one fresh engine (synchronous set-up, no partners) per complex
        e = DockEngine.from_model(model, dev, max_batch=8)
        e.set_complex(_single_complex(base[i * per])[0])
        pos = torch.stack([d["ligand"].pos for d in base[i * per:(i + 1) * per]]).to(dev).contiguous()
        sl = slice(i * per, (i + 1) * per)
        e.sample(pos, steps, noise["tr"][:, sl].to(dev), noise["rot"][:, sl].to(dev), noise["tor"][:, i * per * R:(i + 1) * per * R].to(dev))
        torch.cuda.synchronize()
        want.append(pos.cpu())
    for rep in range(2):
        out, _ = sampling(data_list=[copy.deepcopy(d) for d in base], model=model, inference_steps=S, tr_schedule=sched, rot_schedule=sched,
                          tor_schedule=sched, device=dev, t_to_sigma=partial(t_to_sigma, args=margs), model_args=margs, batch_size=per,
                          noise=noise, co_schedule=2)
        for i in range(len(cps)):
            got = torch.stack([d["ligand"].pos for d in out[i * per:(i + 1) * per]]).cpu()
            assert torch.equal(got, want[i]), (rep, i, float((got - want[i]).abs().max()))
    # the pipelined set-up is a property of the call: the model's cached engines are handed back as they were
    assert model.engine().get_option("async_setup", 0) == 0


def test_pipelined_uploads_wait_for_pending_work_on_the_callers_stream():
    """the pipelined uploads of sampling() take their device buffers from the CURRENT stream's allocator pool and fill them
    on a side stream.  A block the host has already freed may still be written by work queued on the current stream; the fills must be
    ordered behind it (an event on the current stream at wave 0 / right before each wave's launch).  Here the caller leaves exactly that
    behind: tensors of the uploads' sizes, a long spin kernel, then fill_(NaN) on each -- queued, not executed -- and frees them on the
    host right before the call.  Unordered side-stream copies into those blocks would be overwritten with NaN when the spin ends; the
    poses must come out bitwise those of fresh synchronous engines.
    (Measured, tools/check_upload_ordering.py: with today's library the scenario does not corrupt even with the ordering switched off --
    0 of 6 runs -- because cbd_set_complex of wave 0 synchronises the device and cbd_sample* synchronise their stream after the sigma
    upload; the explicit ordering is what keeps the uploads correct if those internal synchronisations are ever removed, and this test is
    its regression guard.)"""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_complex
    from confidence_bootstrapping_amd.utils import make_score_model
    from confidence_bootstrapping_amd.diffusion_utils import t_to_sigma, get_t_schedule
    from confidence_bootstrapping_amd.engine import DockEngine, make_steps, _single_complex
    from confidence_bootstrapping_amd.sampling import sampling, randomize_position, draw_noise_like_reference
    dev = torch.device("cuda:0")
    model, margs = make_score_model(device=dev, seed=0)
    S, per, R = 5, 3, 2
    cps = [make_complex(Nl=9 + i, Nr=28 + 5 * i, R=R, knn=8, seed=160 + i, name=f"q{i}") for i in range(5)]
    torch.manual_seed(4); np.random.seed(4)
    base = [Batch.from_data_list([copy.deepcopy(c)]) for c in cps for _ in range(per)]
    randomize_position(base, False, False, margs.tr_sigma_max)
    sched = get_t_schedule("expbeta", S)
    torch.manual_seed(6)
    noise = draw_noise_like_reference(per * len(cps), R, S, per)
    steps = make_steps(sched, margs, model.timestep_emb_func)
    want = []
    for i, c in enumerate(cps):
        e = DockEngine.from_model(model, dev, max_batch=8)
        e.set_complex(_single_complex(base[i * per])[0])
        pos = torch.stack([d["ligand"].pos for d in base[i * per:(i + 1) * per]]).to(dev).contiguous()
        sl = slice(i * per, (i + 1) * per)
        e.sample(pos, steps, noise["tr"][:, sl].to(dev), noise["rot"][:, sl].to(dev), noise["tor"][:, i * per * R:(i + 1) * per * R].to(dev))
        torch.cuda.synchronize()
        want.append(pos.cpu())
    for rep in range(3):
        torch.cuda.synchronize()
        shapes = [(per, c["ligand"].pos.shape[0], 3) for c in cps] + [(S, per, 3)] * (2 * len(cps)) + [(S, per * R)] * len(cps)
        junk = [torch.empty(sh, device=dev) for sh in shapes for _ in range(3)]
        torch.cuda._sleep(40_000_000)                    # tens of milliseconds of GPU spin in front of the fills
        for t in junk:
            t.fill_(float("nan"))
        del junk, t                                      # freed on the host, the fills still queued behind the spin
        out, _ = sampling(data_list=[copy.deepcopy(d) for d in base], model=model, inference_steps=S, tr_schedule=sched, rot_schedule=sched,
                          tor_schedule=sched, device=dev, t_to_sigma=partial(t_to_sigma, args=margs), model_args=margs, batch_size=per,
                          noise=noise, co_schedule=2)
        for i in range(len(cps)):
            got = torch.stack([d["ligand"].pos for d in out[i * per:(i + 1) * per]]).cpu()
            assert torch.isfinite(got).all() and torch.equal(got, want[i]), (rep, i)
