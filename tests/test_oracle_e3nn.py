"""Consistency pins of the restated e3nn semantics (oracle/e3nn_ref.py) -- the boundary where parity cannot be pinned by
importing e3nn (not installable here).  What CAN be checked without e3nn:

  * O(3) equivariance (random rotations and the inversion) of the spherical harmonics l <= 2, of
    FullyConnectedTensorProduct at lmax = 1 and lmax = 2 and of FullTensorProduct -- a wrong sign / basis / normalisation
    in any Wigner-3j block breaks this;
  * the published values: w3j(1,1,0) = delta/sqrt3, w3j(1,1,1) = epsilon/sqrt6, component-normalised Y2 in e3nn's basis
    (y polar axis), unit Frobenius norm, the (l1 <-> l2) symmetry and the orthogonality relation of every triple used;
  * the reference's OWN hand-written drop-in for e3nn's FullyConnectedTensorProduct (FasterTensorProduct,
    models/tensor_layers.py:66-117, pinned by golden g1 through oracle.score_ref.faster_tensor_product) equals the restated
    FullyConnectedTensorProduct once the per-edge weights are re-laid from output-block-major to instruction-major:
    signs and normalisations of the l <= 1 paths therefore agree with what the reference's authors matched to real e3nn;
  * the w3j(1,2,1) contraction the torsion head hard-codes (csrc/tp_conv.hip::bond_conv_kernel):
    FullTensorProduct(sh(v), Y2(b))[1o] = +(3/sqrt2) sqrt3 (b b^T - I/3) v;
  * the closed-form constants baked into the product package (e3nn_constants.py) equal the restated algorithm's tensors, and the
    checkpoint loader rejects `_w3j_*` buffers that disagree with them.
"""
import math

import numpy as np
import pytest
import torch

from oracle import e3nn_ref as e3
from oracle import score_ref as sr

torch.set_default_dtype(torch.float32)
DT = torch.float64


def rand_rotation(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def d2_of(Rm, rng):
    """Representation matrix of a rotation on the l = 2 real harmonics: Y2(R v) = D2 Y2(v), solved from samples."""
    v = torch.from_numpy(rng.normal(size=(64, 3)))
    a = e3.sh_l2(v).numpy()
    b = e3.sh_l2(v @ torch.from_numpy(Rm).T).numpy()
    D, res, *_ = np.linalg.lstsq(a, b, rcond=None)
    assert np.abs(a @ D - b).max() < 1e-12          # Y2 spans an invariant subspace: the fit is exact
    return D.T


def rep(irreps, Rm, inv, D2):
    """Block-diagonal representation of (rotation Rm, optional inversion) on `irreps`."""
    irreps = e3.Irreps(irreps)
    M = np.zeros((irreps.dim, irreps.dim))
    o = 0
    for mul, ir in irreps:
        D = {0: np.eye(1), 1: Rm, 2: D2}[ir.l] * ((ir.p if inv else 1))
        for _ in range(mul):
            M[o:o + ir.dim, o:o + ir.dim] = D
            o += ir.dim
    return torch.from_numpy(M)


@pytest.fixture(scope="module")
def group_elems():
    rng = np.random.default_rng(5)
    out = []
    for k in range(4):
        Rm = rand_rotation(rng)
        out.append((Rm, bool(k % 2), d2_of(Rm, rng)))
    return out


def test_spherical_harmonics_values_and_equivariance(group_elems):
    s3, s5, s15 = math.sqrt(3), math.sqrt(5), math.sqrt(15)
    v = torch.tensor([[0.0, 1, 0], [1, 0, 0], [0, 0, 1], [1, 1, 0], [0, 2, 2], [3, 0, -3]], dtype=DT)
    y = e3.sh_l2(v).numpy()
    # e3nn _spherical_harmonics.py, normalize=True, normalization='component': sqrt5 * (sqrt3 xz, sqrt3 xy, y^2 - (x^2+z^2)/2, sqrt3 yz, sqrt3/2 (z^2-x^2))
    want = np.array([[0, 0, s5, 0, 0], [0, 0, -s5 / 2, 0, -s15 / 2], [0, 0, -s5 / 2, 0, s15 / 2],
                     [0, s15 / 2, s5 / 4, 0, -s15 / 4], [0, 0, s5 / 4, s15 / 2, s15 / 4], [-s15 / 2, 0, -s5 / 2, 0, 0]])
    np.testing.assert_allclose(y, want, atol=1e-12)
    y1 = e3.sh_l1(v).numpy()
    np.testing.assert_allclose(y1[:, 0], 1.0)
    np.testing.assert_allclose(y1[:, 1:], s3 * (v / v.norm(dim=1, keepdim=True)).numpy(), atol=1e-12)
    rng = np.random.default_rng(0)
    u = torch.from_numpy(rng.normal(size=(50, 3)))
    np.testing.assert_allclose((e3.sh_l2(u) ** 2).sum(1).numpy(), 5.0, atol=1e-12)      # component normalisation: |Y_l|^2 = 2l + 1
    np.testing.assert_allclose((e3.sh_l1(u)[:, 1:] ** 2).sum(1).numpy(), 3.0, atol=1e-12)
    for Rm, inv, D2 in group_elems:
        np.testing.assert_allclose(D2.T @ D2, np.eye(5), atol=1e-12)
        assert abs(np.linalg.det(D2) - 1.0) < 1e-10
        g = torch.from_numpy(Rm * (-1.0 if inv else 1.0))
        full = e3.spherical_harmonics("1x0e+1x1o+1x2e", u @ g.T)
        np.testing.assert_allclose(full.numpy(), (e3.spherical_harmonics("1x0e+1x1o+1x2e", u) @ rep("1x0e+1x1o+1x2e", Rm, inv, D2).T).numpy(),
                                   atol=1e-12)


@pytest.mark.parametrize("in1,in2,out", [
    ("4x0e+2x1o+2x1e+3x0o", "1x0e+1x1o", "3x0e+2x1o+2x1e+2x0o"),                 # score-model layers (lmax = 1)
    ("3x0e+2x1o+2x1e+3x0o", "1x0e+1x1o+1x2e", "3x0e+2x1o+2x1e+3x0o"),            # confidence-model layers (sh_lmax = 2)
    ("4x0e+2x1o+2x1e+2x0o", "1x0e+1x1o", "2x1o+2x1e"),                           # final_conv
    ("4x0e+2x1o+2x1e+2x0o", "1x1o+1x2o+1x2e", "3x0o+3x0e"),                      # tor_bond_conv (3o omitted: it feeds no l = 0 output)
])
def test_fctp_is_o3_equivariant(group_elems, in1, in2, out):
    tp = e3.FullyConnectedTensorProduct(in1, in2, out)
    rng = np.random.default_rng(1)
    E = 7
    x1 = torch.from_numpy(rng.normal(size=(E, tp.irreps_in1.dim)))
    x2 = torch.from_numpy(rng.normal(size=(E, tp.irreps_in2.dim)))
    w = torch.from_numpy(rng.normal(size=(E, tp.weight_numel)))
    y = tp(x1, x2, w)
    assert y.abs().max() > 0.1
    for Rm, inv, D2 in group_elems:
        yg = tp(x1 @ rep(in1, Rm, inv, D2).T, x2 @ rep(in2, Rm, inv, D2).T, w)
        np.testing.assert_allclose(yg.numpy(), (y @ rep(out, Rm, inv, D2).T).numpy(), atol=1e-12)


def test_full_tensor_product_is_o3_equivariant_on_l_le_2(group_elems):
    ftp = e3.FullTensorProduct("1x0e+1x1o", "1x2e")
    assert str(ftp.irreps_out[0][1]) == "1o"                                     # the block the torsion head reads comes first
    rng = np.random.default_rng(2)
    x1, x2 = torch.from_numpy(rng.normal(size=(9, 4))), torch.from_numpy(rng.normal(size=(9, 5)))
    y = ftp(x1, x2)
    sl = ftp.irreps_out.slices()
    for Rm, inv, D2 in group_elems:
        yg = ftp(x1 @ rep("1x0e+1x1o", Rm, inv, D2).T, x2 @ rep("1x2e", Rm, inv, D2).T)
        for (mul, ir), s in zip(ftp.irreps_out, sl):
            if ir.l > 2:
                continue
            np.testing.assert_allclose(yg[:, s].numpy(), (y[:, s] @ rep([(mul, ir)], Rm, inv, D2).T).numpy(), atol=1e-12)


def test_wigner_3j_published_values_and_symmetries():
    eps = np.zeros((3, 3, 3))
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[j, i, k] = 1.0, -1.0
    np.testing.assert_allclose(e3.wigner_3j(1, 1, 0)[:, :, 0], np.eye(3) / math.sqrt(3), atol=1e-14)
    np.testing.assert_allclose(e3.wigner_3j(1, 1, 1), eps / math.sqrt(6), atol=1e-14)
    np.testing.assert_allclose(e3.wigner_3j(0, 0, 0), np.ones((1, 1, 1)), atol=1e-14)
    for ls in ((0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1), (0, 2, 2), (2, 2, 0), (1, 2, 1), (1, 1, 2), (2, 1, 1), (1, 2, 2), (2, 2, 2), (1, 2, 3)):
        l1, l2, l3 = ls
        C = e3.wigner_3j(*ls)
        assert abs(np.linalg.norm(C) - 1.0) < 1e-12
        # sum_ij C_ijk C_ijk' = delta_kk' / (2 l3 + 1)
        np.testing.assert_allclose(np.einsum("ijk,ijl->kl", C, C), np.eye(2 * l3 + 1) / (2 * l3 + 1), atol=1e-12)
        # exchanging the first two arguments: (-1)^(l1+l2+l3)
        np.testing.assert_allclose(e3.wigner_3j(l2, l1, l3), (-1) ** (l1 + l2 + l3) * np.transpose(C, (1, 0, 2)), atol=1e-12)
        # cyclic permutation of the arguments leaves the tensor unchanged
        np.testing.assert_allclose(e3.wigner_3j(l3, l1, l2), np.transpose(C, (2, 0, 1)), atol=1e-12)


def test_baked_constants_match_restated_algorithm_and_loader_checks_them():
    from confidence_bootstrapping_amd.e3nn_constants import w3j_closed_form, check_w3j_buffers
    for ls in ((0, 0, 0), (0, 1, 1), (1, 0, 1), (1, 1, 0), (1, 1, 1), (0, 2, 2), (2, 0, 2), (2, 2, 0), (1, 2, 1), (2, 1, 1), (1, 1, 2)):
        np.testing.assert_allclose(w3j_closed_form(*ls), e3.wigner_3j(*ls), atol=1e-12, err_msg=str(ls))
    assert w3j_closed_form(2, 2, 2) is None
    good = {"final_conv.tp._compiled_main_left_right._w3j_1_1_1": torch.from_numpy(e3.wigner_3j(1, 1, 1)).float(),
            "final_tp_tor._compiled_main_left_right._w3j_1_2_1": torch.from_numpy(e3.wigner_3j(1, 2, 1)).float(),
            "final_tp_tor._compiled_main_left_right._w3j_1_2_3": torch.zeros(3, 5, 7),        # not used by the engine: not checked
            "tr_final_layer.0.weight": torch.zeros(32, 33)}
    assert len(check_w3j_buffers(good)) == 2
    bad = dict(good)
    bad["final_conv.tp._compiled_main_left_right._w3j_1_1_1"] = -good["final_conv.tp._compiled_main_left_right._w3j_1_1_1"]
    with pytest.raises(RuntimeError, match="Wigner-3j"):
        check_w3j_buffers(bad)


def test_model_load_state_dict_rejects_foreign_w3j(score_model):
    model, _ = score_model
    sd = dict(model.state_dict())
    sd["final_conv.tp._compiled_main_left_right._w3j_1_1_0"] = torch.from_numpy(e3.wigner_3j(1, 1, 0)).float()
    model.load_state_dict(sd, strict=True)                                       # matching constants: accepted, then dropped
    sd["tor_bond_conv.tp._compiled_main_left_right._w3j_1_1_0"] = 2.0 * sd["final_conv.tp._compiled_main_left_right._w3j_1_1_0"]
    with pytest.raises(RuntimeError, match="Wigner-3j"):
        model.load_state_dict(sd, strict=True)


def test_torsion_head_contraction():
    """1o block of FullTensorProduct(sh(v), Y2(b)) = +(3/sqrt2) sqrt3 (b b^T - I/3) v  (models/score_model.py:436-437; the
    constant 3.6742346 of bond_conv_kernel)."""
    rng = np.random.default_rng(3)
    v = torch.from_numpy(rng.normal(size=(20, 3)))
    b = torch.from_numpy(rng.normal(size=(20, 3)))
    ftp = e3.FullTensorProduct("1x0e+1x1o", "1x2e")
    y = ftp(e3.sh_l1(v), e3.sh_l2(b))[:, ftp.irreps_out.slices()[0]]
    vh, bh = torch.nn.functional.normalize(v, dim=1), torch.nn.functional.normalize(b, dim=1)
    want = (3.0 / math.sqrt(2.0)) * math.sqrt(3.0) * (bh * (bh * vh).sum(1, keepdim=True) - vh / 3.0)
    np.testing.assert_allclose(y.numpy(), want.numpy(), atol=1e-12)


@pytest.mark.parametrize("lvl_in,lvl_out", [(0, 1), (1, 2), (2, 3), (3, 3)])
def test_reference_faster_tp_equals_restated_fctp(lvl_in, lvl_out):
    """The reference's FasterTensorProduct (golden g1 pins the oracle's copy) against the restated e3nn FullyConnectedTensorProduct
    with the same per-edge weights re-laid from output-block-major [fan_in, mul_out] to instruction-major [mul_in, 1, mul_out]."""
    in_irreps, out_irreps = sr.IRREP_SEQ[lvl_in], sr.IRREP_SEQ[lvl_out]
    tp = e3.FullyConnectedTensorProduct(in_irreps, "1x0e+1x1o", out_irreps)
    n = sr.faster_tp_weight_numel(in_irreps, out_irreps)
    assert tp.weight_numel == n
    rng = np.random.default_rng(4)
    E = 11
    x = torch.from_numpy(rng.normal(size=(E, tp.irreps_in1.dim)))
    sh = e3.sh_l1(torch.from_numpy(rng.normal(size=(E, 3))))
    w_fast = torch.from_numpy(rng.normal(size=(E, n)))
    # FasterTensorProduct: per output irrep (0e, 1o, 1e, 0o) a block [fan_in rows, mul_out]; the rows are the contributing
    # (in1 irrep, sh irrep) pairs in in1 order, mul_in rows each -- the instruction order restricted to that output
    shapes = sr.faster_tp_weight_shapes(in_irreps, out_irreps)
    out_keys = [str(ir) for _, ir in tp.irreps_out]
    block_off, o = {}, 0
    for key in ("0e", "1o", "1e", "0o"):
        fin, mo = shapes[key]
        block_off[key] = o
        o += fin * mo
    row_next = {k: 0 for k in block_off}
    w_fctp = torch.zeros(E, n, dtype=DT)
    off = 0
    for (i1, i2, io) in tp.instructions:
        m1, mo = tp.irreps_in1[i1][0], tp.irreps_out[io][0]
        key = out_keys[io]
        r0 = row_next[key]
        blk = w_fast[:, block_off[key] + r0 * mo: block_off[key] + (r0 + m1) * mo]          # [E, m1 * mo], row-major (row, out)
        w_fctp[:, off:off + m1 * mo] = blk
        row_next[key] += m1
        off += m1 * mo
    got = tp(x, sh, w_fctp)
    want = sr.faster_tensor_product(x, sh, w_fast, in_irreps, out_irreps)
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=1e-12)


def test_batchnorm_eval_field_semantics():
    """e3nn.nn.BatchNorm in eval mode: 0e channels (x - mean) * rsqrt(var + eps) * w + b, every other irrep x * rsqrt(var + eps) * w with
    one statistic per multiplicity channel (shared by the 2l+1 components) -- hence equivariant."""
    irreps = "3x0e+2x1o+2x0o"
    bn = e3.BatchNorm(irreps).double().eval()
    rng = np.random.default_rng(6)
    with torch.no_grad():
        bn.weight.copy_(torch.from_numpy(rng.uniform(0.5, 2, 7)))
        bn.bias.copy_(torch.from_numpy(rng.normal(size=3)))
        bn.running_mean.copy_(torch.from_numpy(rng.normal(size=3)))
        bn.running_var.copy_(torch.from_numpy(rng.uniform(0.5, 2, 7)))
    x = torch.from_numpy(rng.normal(size=(5, 11)))
    torch.set_grad_enabled(False)
    y = bn(x)
    sc = (bn.running_var + 1e-5).rsqrt() * bn.weight
    np.testing.assert_allclose(y[:, :3].numpy(), ((x[:, :3] - bn.running_mean) * sc[:3] + bn.bias).numpy(), atol=1e-12)
    np.testing.assert_allclose(y[:, 3:9].numpy(), (x[:, 3:9].reshape(5, 2, 3) * sc[3:5, None]).reshape(5, 6).numpy(), atol=1e-12)
    np.testing.assert_allclose(y[:, 9:].numpy(), (x[:, 9:] * sc[5:]).numpy(), atol=1e-12)
    Rm = rand_rotation(rng)
    M = rep(irreps, Rm, True, np.eye(5))
    np.testing.assert_allclose(bn(x @ M.T).numpy(), (y @ M.T).numpy(), atol=1e-12)
    torch.set_grad_enabled(True)


def test_wigner3j_matches_the_sympy_derivation():
    """a second pin of sign and normalisation that shares no code with the build's own CG / change-of-basis
    functions (oracle/pin_wigner_sympy.py: sympy's exact Clebsch-Gordan coefficients, a change of basis FITTED from sympy's Ynm against
    the real polynomials e3nn documents, e3nn's published einsum recipe).  All 15 triangle-admissible triples with l <= 2 -- everything
    the score model's heads and the confidence model's lmax = 2 tensor products contract with: the oracle's wigner_3j, the closed forms
    hard-wired in the kernels (e3nn_constants.py) and the committed fixture g20 agree to 1e-12, signs included."""
    import os
    from oracle import pin_wigner_sympy as pw
    from oracle import e3nn_ref
    from confidence_bootstrapping_amd import e3nn_constants
    tab = pw.derive(seed=3)                       # other random fitting points than the fixture's
    assert len(tab) == 15
    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g20_wigner3j_sympy.npz"))
    n_closed = 0
    for ls, c in tab.items():
        assert abs(np.linalg.norm(c) - 1.0) < 1e-12
        assert np.abs(fix["w3j_%d_%d_%d" % ls] - c).max() < 1e-12, ls
        assert np.abs(e3nn_ref.wigner_3j(*ls) - c).max() < 1e-12, ls
        cf = e3nn_constants.w3j_closed_form(*ls)
        if cf is not None:
            n_closed += 1
            assert np.abs(cf - c).max() < 1e-12, ls
    assert n_closed == 11
    # anchors that do not depend on e3nn's recipe: the two tensors the reference writes out by hand (models/tensor_layers.py:76-82)
    eps = np.zeros((3, 3, 3))
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[j, i, k] = 1.0, -1.0
    assert np.abs(tab[(1, 1, 1)] - eps / np.sqrt(6.0)).max() < 1e-12
    assert np.abs(tab[(1, 1, 0)][:, :, 0] - np.eye(3) / np.sqrt(3.0)).max() < 1e-12
    # and the even-sum triples against the real Gaunt integrals of the documented basis (quadrature on the sphere): same tensor up
    # to ONE positive or negative factor per triple -- the structure is basis-convention free
    rng = np.random.default_rng(0)
    v = rng.normal(size=(200000, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    for ls, c in tab.items():
        if sum(ls) % 2:
            continue
        R = [pw.real_basis(l, v) for l in ls]
        g = np.einsum("na,nb,nc->abc", R[0], R[1], R[2]) / len(v)
        g /= np.linalg.norm(g)
        assert min(np.abs(g - c).max(), np.abs(g + c).max()) < 2e-2, ls          # Monte-Carlo quadrature: 1 / sqrt(n)
