"""bf16-operand variant of the tensor-product layers (cbd_set_option("bf16", 1), BASELINE.json configs[3]) against the fp32
path / the fp32 oracle.  bf16 has 8 significant bits: with fp32 accumulation over K = 96 and 8 stacked layers the scores
agree with fp32 to ~1e-2 relative (tolerance below: 2e-2 of the largest component; measured 1e-3..5e-3, and 0.003 A median RMSD between 20-step trajectories, tools/bf16_accuracy.py); the pose update itself stays fp32."""
import copy

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def model_args():
    from confidence_bootstrapping_amd.utils import make_score_model
    return make_score_model(device="cuda:0", seed=0)


@pytest.mark.parametrize("workload,B", [("tiny", 3), ("c2_dockgen_median", 4)])
def test_bf16_scores_close_to_fp32(model_args, workload, B):
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.engine import make_steps
    from confidence_bootstrapping_amd.sampling import randomize_position
    model, args = model_args
    cplx = make_workload(workload)
    torch.manual_seed(4); np.random.seed(4)
    dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
    randomize_position(dl, False, False, 5.0)
    pos = torch.stack([d["ligand"].pos for d in dl]).cuda()
    eng = model.engine()
    eng.set_complex(cplx)
    for t in (1.0, 0.3):
        step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
        eng.set_option("bf16", 0)
        tr32, rot32, tor32 = [x.clone() for x in eng.score(pos, step)]
        eng.set_option("bf16", 1)
        tr16, rot16, tor16 = eng.score(pos, step)
        eng.set_option("bf16", 0)
        for a, b in ((tr16, tr32), (rot16, rot32), (tor16, tor32)):
            assert torch.isfinite(a).all()
            assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()), (t, float((a - b).abs().max()), float(b.abs().max()))
        assert not torch.equal(tr16, tr32)      # the bf16 kernels really ran


def test_bf16_trajectory_properties(model_args):
    """40-step-free sanity of a full sampling run in bf16: finite, bond lengths preserved, deterministic run to run."""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.engine import make_steps
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
    from confidence_bootstrapping_amd.sampling import randomize_position
    model, args = model_args
    cplx = make_workload("c2_dockgen_median")
    B, S = 8, 20
    torch.manual_seed(9); np.random.seed(9)
    dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
    randomize_position(dl, False, False, args.tr_sigma_max)
    pos0 = torch.stack([d["ligand"].pos for d in dl]).cuda()
    steps = make_steps(get_t_schedule("expbeta", S), args, model.timestep_emb_func)
    R = int(cplx["ligand"].edge_mask.sum())
    g = torch.Generator().manual_seed(1)
    noise = [torch.randn(S, B, 3, generator=g), torch.randn(S, B, 3, generator=g), torch.randn(S, B * R, generator=g)]
    eng = model.engine()
    eng.set_complex(cplx)
    eng.set_option("bf16", 1)
    try:
        outs = []
        for _ in range(2):
            p = pos0.clone()
            eng.sample(p, steps, *noise)
            outs.append(p)
    finally:
        eng.set_option("bf16", 0)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
    ei = cplx["ligand", "ligand"].edge_index
    d0 = (cplx["ligand"].pos[ei[0]] - cplx["ligand"].pos[ei[1]]).norm(dim=-1)
    d1 = (outs[0][:, ei[0]] - outs[0][:, ei[1]]).norm(dim=-1).cpu()
    assert float((d1 - d0[None]).abs().max()) < 1e-3


def test_f32_split_is_fp32_grade(model_args):
    """cbd_set_option("f32_split", 1): fp32 operands as the exact sum of three bf16 planes on the bf16 matrix cores (6 of the 9
    plane products, fp32 accumulate).  Scores agree with the exact-fp32 kernel to fp32 rounding level (stated: 2e-5 of the largest
    component, the tolerance of the fp32 parity tests; measured 2e-7..2e-6), the reference's golden 20-step trajectory is met
    within the north-star 1e-3 A, and repeated trajectories are bitwise identical."""
    import os
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.engine import make_steps
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
    from confidence_bootstrapping_amd.sampling import randomize_position
    model, args = model_args
    cplx = make_workload("c2_dockgen_median")
    B = 6
    torch.manual_seed(4); np.random.seed(4)
    dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
    randomize_position(dl, False, False, 5.0)
    pos = torch.stack([d["ligand"].pos for d in dl]).cuda()
    eng = model.engine()
    eng.set_complex(cplx)
    try:
        for t in (1.0, 0.3, 0.05):
            step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
            eng.set_option("f32_split", 0)
            ref = [x.clone() for x in eng.score(pos, step)]
            eng.set_option("f32_split", 1)
            got = eng.score(pos, step)
            for a, b in zip(got, ref):
                assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
        # the reference's own 20-step trajectory (tests/golden/g6_sampling.npz) in this mode
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g6_sampling.npz"))
        tiny = make_workload("tiny")
        eng.set_complex(tiny)
        steps = make_steps(g["schedule"], args, model.timestep_emb_func)
        outs = []
        for _ in range(3):
            p = torch.from_numpy(g["pos0"]).cuda().contiguous()
            eng.sample(p, steps, torch.from_numpy(g["noise_tr"]), torch.from_numpy(g["noise_rot"]), torch.from_numpy(g["noise_tor"]))
            outs.append(p)
        rmsd = torch.sqrt(((outs[0].cpu() - torch.from_numpy(g["final_pos"])) ** 2).sum(-1).mean(-1))
        assert float(rmsd.max()) < 1e-3
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    finally:
        eng.set_option("f32_split", 0)


def test_config_c4_bf16_as_specified(model_args, tables):
    """BASELINE.json configs[3] at its stated shape: large-pocket complex (Nl = 64, Nr = 1024, R = 16), 64 samples x 40 denoise steps,
    bf16 operands / fp32 accumulate -- against the fp32 ORACLE (not the fp32 HIP path).
      * scores of the 64-pose batch at t = 1.0 and t = 0.3: oracle on poses 0 and 37 of the same batch (samples are independent);
        stated tolerance 2e-2 of the largest component of each output (bf16 has 8 significant bits; measured ~3e-3);
      * trajectory: the first 3 of the 40 steps of pose 0 against the oracle's fp32 trajectory on the same noise, RMSD < 0.05 A,
        and the full 64 x 40 run against the fp32 HIP run (itself oracle-checked on this workload in test_gpu_parity.py)
        median RMSD < 0.05 A (measured ~0.005 A);
      * size-independent properties of the full run: finite, bond lengths preserved to 5e-3 A, bitwise repeatable."""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.engine import DockEngine, make_steps
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
    from confidence_bootstrapping_amd.sampling import randomize_position
    from oracle import score_ref as sr, pose_ref as pr
    from tests.helpers import to_cx
    model, args = model_args
    so3, torus = tables
    cplx = make_workload("c4_large_pocket")
    cx = to_cx(cplx)
    B, S = 64, 40
    dev = torch.device("cuda:0")
    eng = DockEngine.from_model(model, dev, max_batch=B)
    eng.set_complex(cplx)
    torch.manual_seed(12); np.random.seed(12)
    dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
    randomize_position(dl, False, False, args.tr_sigma_max)
    pos0 = torch.stack([d["ligand"].pos for d in dl])
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = sr.ScoreConfig()
    rec_cache = sr.receptor_embedding(sd, cx, cfg)
    pick = [0, 37]
    R = eng.R
    eng.set_option("bf16", 1)
    try:
        for t in (1.0, 0.3):
            step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
            tr, rot, tor = [x.cpu() for x in eng.score(pos0.to(dev), step)]
            ref = sr.score_forward(sd, cx, pos0[pick], t, t, t, cfg, so3, torus, rec_cache=rec_cache)
            for got, want in ((tr[pick], ref["tr_pred"]), (rot[pick], ref["rot_pred"]),
                              (tor.reshape(B, R)[pick].reshape(-1), ref["tor_pred"])):
                assert torch.isfinite(got).all()
                err, scale = float((got - want).abs().max()), float(want.abs().max())
                assert err <= 2e-2 * scale, (t, err, scale)
        sched = get_t_schedule("expbeta", S)
        steps = make_steps(sched, args, model.timestep_emb_func)
        g = torch.Generator().manual_seed(5)
        noise = [torch.randn(S, B, 3, generator=g), torch.randn(S, B, 3, generator=g), torch.randn(S, B * R, generator=g)]
        # first 3 steps of pose 0 vs the oracle's fp32 trajectory (same noise)
        n3 = {"tr": noise[0][:3, :1], "rot": noise[1][:3, :1], "tor": noise[2][:3, :R]}
        n4 = {"tr": noise[0][:4, :1], "rot": noise[1][:4, :1], "tor": noise[2][:4, :R]}
        _, trace = pr.sampling_ref(sd, cx, pos0[:1], sched[:4], cfg, so3, torus, noise=n4, record=True)   # steps 0..2 of the prefix = steps 0..2 of the run
        ref3 = trace[2]["pos"]
        p3 = pos0[:1].to(dev).contiguous()
        eng.sample(p3, (type(steps[0]) * 3)(*[steps[i] for i in range(3)]), n3["tr"], n3["rot"], n3["tor"])
        r3 = float(torch.sqrt(((p3.cpu() - ref3) ** 2).sum(-1).mean(-1)).max())
        assert r3 < 0.05, r3
        outs = []
        for _ in range(2):
            p = pos0.to(dev).contiguous()
            eng.sample(p, steps, *noise)
            outs.append(p)
        eng.set_option("bf16", 0)
        p32 = pos0.to(dev).contiguous()
        eng.sample(p32, steps, *noise)
    finally:
        eng.set_option("bf16", 0)
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
    rm = torch.sqrt(((outs[0] - p32) ** 2).sum(-1).mean(-1)).cpu()
    assert float(rm.median()) < 0.05, (float(rm.median()), float(rm.max()))
    ei = cplx["ligand", "ligand"].edge_index
    d0 = (cplx["ligand"].pos[ei[0]] - cplx["ligand"].pos[ei[1]]).norm(dim=-1)
    d1 = (outs[0][:, ei[0]] - outs[0][:, ei[1]]).norm(dim=-1).cpu()
    assert float((d1 - d0[None]).abs().max()) < 5e-3


def test_product_library_refuses_the_role_split_experiment(model_args):
    """The bf16 role split (correct, slower) is compiled into the diagnostic library only (tools/diag_lib.py,
    experiments/); the product library refuses the option loudly instead of silently ignoring it.  Its own test: experiments/test_role_split.py."""
    from confidence_bootstrapping_amd.synthetic import make_workload
    model, _ = model_args
    eng = model.engine()
    eng.set_complex(make_workload("tiny"))
    eng.set_option("bf16_roles", 0)
    with pytest.raises(RuntimeError, match="diagnostic library"):
        eng.set_option("bf16_roles", 1)


def test_bf16_stationary_kernel_agrees_with_the_streaming_kernel(model_args):
    """cbd_set_option("bf16_stationary", 1) (tp_conv_bf16s.hip): the 74 -> 74 layers of the bf16 policy through persistent
    four-wave workgroups that keep a whole FCBlock in registers.  Same bf16 products as the streaming kernel; a message's tile
    contributions are added per wave and then across waves, so the two kernels agree to fp32 rounding of those sums (2e-5 of the largest
    component here; measured 5e-7 .. 3e-6), not bitwise.  Deterministic.  Covered: a tiny complex (ragged last units, roles with a handful
    of units), a C2-sized one, three co-scheduled complexes in one launch (several entries per role; R = 0 ligand) eager and under
    hipGraph replay, and a whole trajectory against the streaming kernel."""
    from confidence_bootstrapping_amd import Batch
    from confidence_bootstrapping_amd.synthetic import make_workload, make_complex
    from confidence_bootstrapping_amd.engine import DockEngine, make_steps
    from confidence_bootstrapping_amd.diffusion_utils import get_t_schedule
    from confidence_bootstrapping_amd.sampling import randomize_position
    model, args = model_args
    dev = torch.device("cuda:0")
    for workload, B in (("tiny", 3), ("c2_dockgen_median", 4), ("c4_large_pocket", 8)):      # C4: hundreds of units per workgroup, several segments
        cplx = make_workload(workload)
        torch.manual_seed(6); np.random.seed(6)
        dl = [Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
        randomize_position(dl, False, False, 5.0)
        pos = torch.stack([d["ligand"].pos for d in dl]).cuda()
        eng = model.engine()
        eng.set_complex(cplx)
        try:
            eng.set_option("bf16", 1)
            for t in (1.0, 0.4):
                step = make_steps(np.array([t]), args, model.timestep_emb_func)[0]
                eng.set_option("bf16_stationary", 0)
                ref = [x.clone() for x in eng.score(pos, step)]
                eng.set_option("bf16_stationary", 1)
                got = [x.clone() for x in eng.score(pos, step)]
                again = eng.score(pos, step)
                assert all(torch.equal(p, q) for p, q in zip(got, again)), (workload, t)
                for p, q in zip(got, ref):
                    assert torch.isfinite(p).all()
                    assert float((p - q).abs().max()) <= 2e-5 * float(q.abs().max()), (workload, t, float((p - q).abs().max()), float(q.abs().max()))
                assert not all(torch.equal(p, q) for p, q in zip(got, ref))      # the other kernel really ran
        finally:
            eng.set_option("bf16_stationary", 1)      # the default
            eng.set_option("bf16", 0)
    # three different complexes in one launch, S steps: stationary eager = stationary under hipGraph replay (bitwise), close to streaming
    cps = [make_workload("tiny"), make_complex(Nl=17, Nr=60, R=3, knn=10, seed=77), make_complex(Nl=9, Nr=33, R=0, knn=8, seed=78)]
    Bs, S = [5, 3, 6], 4
    steps = make_steps(get_t_schedule("expbeta", S), args, model.timestep_emb_func)
    engs = [DockEngine.from_model(model, dev, max_batch=8)]
    for _ in range(2):
        e = DockEngine(dev, max_batch=8)
        e.share_weights_from(engs[0])
        engs.append(e)
    g = torch.Generator().manual_seed(4)
    inputs = []
    for e, c, B in zip(engs, cps, Bs):
        e.set_complex(c)
        torch.manual_seed(B); np.random.seed(B)
        dl = [Batch.from_data_list([copy.deepcopy(c)]) for _ in range(B)]
        randomize_position(dl, False, False, 5.0)
        R = int(c["ligand"].edge_mask.sum())
        inputs.append((torch.stack([d["ligand"].pos for d in dl]).to(dev).contiguous(),
                       [torch.randn(S, B, 3, generator=g).to(dev), torch.randn(S, B, 3, generator=g).to(dev), torch.randn(S, B * R, generator=g).to(dev)]))

    def run(stationary, graph):
        for e in engs:
            e.set_option("bf16", 1); e.set_option("bf16_stationary", stationary); e.set_option("graph", graph)
        ps = [p.clone() for p, _ in inputs]
        for _ in range(2 if graph else 1):          # second call replays the captured graph
            ps = [p.clone() for p, _ in inputs]
            DockEngine.sample_multi(engs, ps, steps, [nz for _, nz in inputs])
        torch.cuda.synchronize()
        return ps
    try:
        stream = run(0, 0)
        eager = run(1, 0)
        replay = run(1, 1)
    finally:
        for e in engs:
            e.set_option("bf16_stationary", 1); e.set_option("bf16", 0); e.set_option("graph", 1)
    for a, b, c in zip(eager, replay, stream):
        assert torch.equal(a, b)
        assert torch.isfinite(a).all() and float((a - c).norm(dim=-1).max()) < 2e-3      # positions after S steps, in Angstrom
