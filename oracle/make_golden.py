"""Generate tests/golden/*.npz by RUNNING THE REFERENCE's own modules (this container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python oracle/make_golden.py   (needs /root/reference and the cached
score-normaliser tables produced by oracle/gen_tables.py in /tmp/cb_tables).

Every array stored is an input or an output of reference code (see oracle/ref_import.py for exactly which
third-party pieces are shims).  The fixtures are data; no reference source text is stored.

  g1_faster_tp.npz     FasterTensorProduct.forward for the four layer shapes     tensor_layers.py:66-117
  g2_conv_layer.npz    TensorProductConvLayer.forward, 4 edge groups + last-layer 2 groups   tensor_layers.py:195-217
  g3_small_ops.npz     GaussianSmearing, sinusoidal_embedding, AtomEncoder, get_t_schedule, t_to_sigma,
                       so3.score_norm / torus.score_norm at the schedule points
  g4_pose.npz          axis_angle_to_matrix, Kabsch batch, torsion batch update, modify_conformer_batch
  g6_forward.npz       TensorProductScoreModel.forward on a synthetic batch (B=3) at three times
  g6_sampling.npz      utils.sampling.sampling(): 20 steps, B=3, with the drawn noise recorded
  g7_randomize.npz     randomize_position with recorded numpy/torch RNG seeds
"""
from __future__ import annotations

import copy
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KB, {len(arrs)} arrays", flush=True)


def main():
    from oracle import ref_import
    hetero = ref_import.install()
    from confidence_bootstrapping_amd.synthetic import make_workload
    from confidence_bootstrapping_amd.utils import make_score_model

    import utils.so3 as ref_so3
    import utils.torus as ref_torus
    from models.tensor_layers import FasterTensorProduct, TensorProductConvLayer, get_irrep_seq
    from models.score_model import GaussianSmearing, AtomEncoder
    from utils.diffusion_utils import (sinusoidal_embedding, get_t_schedule, t_to_sigma, modify_conformer_batch,
                                       set_time)
    from utils.geometry import axis_angle_to_matrix, rigid_transform_Kabsch_3D_torch_batch
    from utils.torsion import modify_conformer_torsion_angles_batch
    import utils.sampling as ref_sampling

    torch.set_num_threads(8)
    g = torch.Generator().manual_seed(2024)
    rn = lambda *s: torch.randn(*s, generator=g)

    # ------------------------------------------------------------------ G1
    seq = get_irrep_seq(32, 6, False, True)
    arrs = {}
    for i in range(4):
        a, b = seq[min(i, 3)], seq[min(i + 1, 3)]
        tp = FasterTensorProduct(a, "1x0e+1x1o", b)
        E = 12
        from oracle.e3nn_ref import Irreps, sh_l1
        x = rn(E, Irreps(a).dim)
        sh = sh_l1(rn(E, 3))
        w = torch.randint(-32, 33, (E, tp.weight_numel), generator=g).float() / 16  # coarse grid: compresses well
        arrs[f"x{i}"], arrs[f"sh{i}"], arrs[f"w{i}"] = x, sh, w
        arrs[f"out{i}"] = tp(x, sh, w)
        arrs[f"numel{i}"] = np.array(tp.weight_numel)
    npz("g1_faster_tp.npz", **arrs)

    # ------------------------------------------------------------------ G2
    arrs = {}
    irr = seq[3]
    for tag, groups in (("g4", 4), ("g2", 2)):
        torch.manual_seed(11 + groups)
        layer = TensorProductConvLayer(irr, "1x0e+1x1o", irr, 96, hidden_features=96, residual=True, batch_norm=True,
                                       dropout=0.1, faster=True, edge_groups=groups)
        layer.eval()
        with torch.no_grad():
            for p_ in layer.fc.parameters():  # coarse-grid weights so the fixture compresses (values are still generic)
                p_.copy_(torch.randint(-32, 33, p_.shape, generator=g).float() / 256)
            layer.batch_norm.running_mean.copy_(0.1 * rn(32))
            layer.batch_norm.running_var.copy_(torch.rand(50, generator=g) + 0.5)
            layer.batch_norm.weight.copy_(1 + 0.1 * rn(50))
            layer.batch_norm.bias.copy_(0.1 * rn(32))
        N, sizes = 20, [30, 50, 40, 25][:groups]
        E = sum(sizes)
        node = rn(N, 74)
        ei = torch.randint(0, N, (2, E), generator=g)
        ei[0, :5] = 3  # several edges into one node; node N-1 left without edges where possible
        ei[ei == N - 1] = 0
        ea = rn(E, 96)
        sh = sh_l1(rn(E, 3))
        offs = np.cumsum([0] + sizes)
        ea_groups = [ea[offs[k]:offs[k + 1]] for k in range(groups)]
        with torch.no_grad():
            out = layer(node, ei, ea_groups, sh, edge_weight=1.0)
        arrs.update({f"{tag}_node": node, f"{tag}_edge_index": ei, f"{tag}_edge_attr": ea, f"{tag}_sh": sh,
                     f"{tag}_sizes": np.array(sizes), f"{tag}_out": out})
        for k, v in layer.state_dict().items():
            arrs[f"{tag}_sd.{k}"] = v
    npz("g2_conv_layer.npz", **arrs)

    # ------------------------------------------------------------------ G3
    arrs = {}
    d = torch.rand(64, generator=g) * 40
    for name, (lo, hi) in {"lig": (0.0, 5.0), "rec": (0.0, 30.0), "cross": (0.0, 80.0)}.items():
        arrs[f"gs_{name}"] = GaussianSmearing(lo, hi, 32)(d)
    arrs["gs_d"] = d
    sched20 = get_t_schedule("expbeta", 20)
    sched40 = get_t_schedule("expbeta", 40)
    arrs["sched20"], arrs["sched40"] = sched20, sched40
    _, margs = make_score_model()
    arrs["sigma20"] = np.stack([np.asarray(t_to_sigma(t, t, t, margs)) for t in sched20])
    tt = torch.tensor(sched20, dtype=torch.float32)
    arrs["sin_emb20"] = sinusoidal_embedding(1000 * tt, 32)
    arrs["sin_emb20_f32_t"] = tt
    sig_t = t_to_sigma(tt, tt, tt, margs)
    arrs["so3_norm20"] = ref_so3.score_norm(sig_t[1])
    arrs["torus_norm20"] = ref_torus.score_norm(sig_t[2].numpy())
    torch.manual_seed(5)
    from datasets.process_mols import lig_feature_dims, rec_residue_feature_dims
    arrs["lig_feature_dims"] = np.array(lig_feature_dims[0])
    arrs["rec_feature_dims"] = np.array(rec_residue_feature_dims[0])
    enc = AtomEncoder(32, lig_feature_dims, 32)
    xcat = torch.stack([torch.randint(0, dd, (10,), generator=g) for dd in lig_feature_dims[0]], 1).float()
    xin = torch.cat([xcat, rn(10, 32)], 1)
    with torch.no_grad():
        arrs["enc_out"] = enc(xin)
    arrs["enc_in"] = xin
    for k, v in enc.state_dict().items():
        arrs[f"enc_sd.{k}"] = v
    npz("g3_small_ops.npz", **arrs)

    # ------------------------------------------------------------------ G4
    arrs = {}
    aa = rn(8, 3)
    aa[0] = 0.0
    aa[1] = torch.tensor([3e-7, -2e-7, 1e-7])
    aa[2] = aa[2] / aa[2].norm() * 3.1
    arrs["aa"], arrs["aa_mat"] = aa, axis_angle_to_matrix(aa)
    A = rn(5, 9, 3)
    Rt = axis_angle_to_matrix(rn(5, 3))
    Bm = torch.bmm(A, Rt.transpose(1, 2)) + rn(5, 1, 3) + 0.05 * rn(5, 9, 3)
    Bm[4] = A[4] * torch.tensor([1.0, 1.0, -1.0]) + 0.01 * rn(9, 3)  # forces the reflection branch
    R_, t_ = rigid_transform_Kabsch_3D_torch_batch(A, Bm)
    arrs["kab_A"], arrs["kab_B"], arrs["kab_R"], arrs["kab_t"] = A, Bm, R_, t_
    for wl, Rn in (("tiny", 2), ("c2_dockgen_median", 6), ("c4_large_pocket", 16)):
        cplx = make_workload(wl)
        b = 3
        batch = hetero.Batch.from_data_list([copy.deepcopy(cplx) for _ in range(b)])
        pos = batch["ligand"].pos + 0.3 * rn(*batch["ligand"].pos.shape)
        mask_rotate = torch.from_numpy(cplx["ligand"].mask_rotate)
        tr, rot, tor = rn(b, 3), 0.5 * rn(b, 3), rn(b * Rn)
        M = cplx["ligand", "ligand"].num_edges
        ei = batch["ligand", "ligand"].edge_index[:, :M]
        em = batch["ligand"].edge_mask[:M]
        flex = modify_conformer_torsion_angles_batch(pos.reshape(b, -1, 3), ei.T[em], mask_rotate, tor.reshape(b, -1))
        new = modify_conformer_batch(pos, batch, tr, rot, tor, mask_rotate)
        rigid_only = modify_conformer_batch(pos, batch, tr, rot, None, mask_rotate)
        arrs.update({f"{wl}_pos": pos, f"{wl}_tr": tr, f"{wl}_rot": rot, f"{wl}_tor": tor, f"{wl}_flex": flex,
                     f"{wl}_new": new, f"{wl}_rigid": rigid_only})
    npz("g4_pose.npz", **arrs)

    # ------------------------------------------------------------------ G6 forward + sampling on 'tiny'
    mine, margs = make_score_model(seed=0)
    sd = {k: v.clone() for k, v in mine.state_dict().items()}
    ref_model, _ = ref_import.reference_score_model(sd)
    cplx = make_workload("tiny")
    B = 3
    torch.manual_seed(123)
    np.random.seed(123)
    # like inference.py:409-424 the list elements are 1-graph batches (mask_rotate list-wrapped)
    data_list = [hetero.Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(B)]
    ref_sampling.randomize_position(data_list, False, False, margs.tr_sigma_max)
    pos0 = torch.stack([d["ligand"].pos for d in data_list])
    arrs = {"pos0": pos0}
    for t in (1.0, 0.5, 0.05):
        batch = hetero.Batch.from_data_list([copy.deepcopy(d) for d in data_list])
        set_time(batch, None, t, t, t, B, False, False, torch.device("cpu"))
        with torch.no_grad():
            tr, rot, tor, _ = ref_model(batch)
        tag = f"t{t}"
        arrs[f"{tag}_tr"], arrs[f"{tag}_rot"], arrs[f"{tag}_tor"] = tr, rot, tor
    npz("g6_forward.npz", **arrs)

    # sampling(): record the noise the reference draws (torch.normal on the global CPU generator)
    drawn = []
    real_normal = torch.normal

    def rec_normal(*a, **k):
        z = real_normal(*a, **k)
        drawn.append(z.clone())
        return z

    ref_sampling.DataLoader = hetero.DataLoader
    ref_sampling.Batch = hetero.Batch
    S = 20
    sched = get_t_schedule("expbeta", S)
    step_scores = []
    orig_forward = ref_model.forward

    def spy(batch):
        out = orig_forward(batch)
        step_scores.append([o.clone() for o in out[:3]])
        return out

    torch.manual_seed(42)
    torch.normal = rec_normal
    try:
        dl = [copy.deepcopy(d) for d in data_list]
        from functools import partial
        out_list, conf = ref_sampling.sampling(dl, spy, S, sched, sched, sched, torch.device("cpu"),
                                               partial(t_to_sigma, args=margs), margs, batch_size=B)
    finally:
        torch.normal = real_normal
    assert conf is None and len(drawn) == 3 * S
    arrs = {"pos0": pos0, "schedule": sched,
            "noise_tr": torch.stack(drawn[0::3]), "noise_rot": torch.stack(drawn[1::3]), "noise_tor": torch.stack(drawn[2::3]),
            "final_pos": torch.stack([d["ligand"].pos for d in out_list]),
            "step_tr": torch.stack([s[0] for s in step_scores]), "step_rot": torch.stack([s[1] for s in step_scores]),
            "step_tor": torch.stack([s[2] for s in step_scores])}
    npz("g6_sampling.npz", **arrs)

    # ------------------------------------------------------------------ G7 randomize_position
    arrs = {}
    for wl in ("tiny", "c2_dockgen_median"):
        cplx = make_workload(wl)
        dl = [hetero.Batch.from_data_list([copy.deepcopy(cplx)]) for _ in range(4)]
        np.random.seed(7)
        torch.manual_seed(7)
        ref_sampling.randomize_position(dl, False, False, margs.tr_sigma_max)
        arrs[f"{wl}_pos"] = torch.stack([d["ligand"].pos for d in dl])
    npz("g7_randomize.npz", **arrs)


if __name__ == "__main__":
    main()
