"""CPU prototypes of the closed-form arithmetic hard-wired in csrc/kernels.hip, checked against the oracle's
generic e3nn restatement (Wigner-3j einsums) and SVD-based Kabsch.  The HIP kernels are transliterations of
these functions; the -m gpu tests then check the kernels themselves."""
import math

import numpy as np
import torch

from oracle import e3nn_ref as e3, pose_ref as pr

S3, S6 = math.sqrt(3.0), math.sqrt(6.0)


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1], a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], -1)


def final_conv_tp(x, v, w):
    """x [E,74], v [E,3] unit edge vector (sh = [1, sqrt3 v]), w [E,124] -> [E,12] = 2x1o | 2x1e (kernels.hip center_head)."""
    E = x.shape[0]
    x0e, x1o, x1e, x0o = x[:, :32], x[:, 32:50].reshape(E, 6, 3), x[:, 50:68].reshape(E, 6, 3), x[:, 68:74]
    sh1 = S3 * v
    w0 = w[:, 0:64].reshape(E, 32, 2)
    w1, w2, w3, w4, w5 = [w[:, 64 + 12 * k: 76 + 12 * k].reshape(E, 6, 2) for k in range(5)]
    pw_o, pw_e = math.sqrt(3 / 44), math.sqrt(3 / 18)
    o = np.zeros((E, 2, 3))
    e = np.zeros((E, 2, 3))
    o += pw_o * np.einsum("euw,eu,ek->ewk", w0, x0e, sh1) / S3
    o += pw_o * np.einsum("euw,euk->ewk", w1, x1o) / S3
    e += pw_e * np.einsum("euw,euk->ewk", w2, cross(x1o, sh1[:, None, :])) / S6
    e += pw_e * np.einsum("euw,euk->ewk", w3, x1e) / S3
    o += pw_o * np.einsum("euw,euk->ewk", w4, cross(x1e, sh1[:, None, :])) / S6
    e += pw_e * np.einsum("euw,eu,ek->ewk", w5, x0o, sh1) / S3
    return np.concatenate([o.reshape(E, 6), e.reshape(E, 6)], 1)


def tor_t1(v, b):
    """1o block of FullTensorProduct(sh(v), Y2(b)):  (3/sqrt2) * (b b^T - I/3) (sqrt3 v)  for unit v, b."""
    bv = (b * v).sum(-1, keepdims=True)
    return (3.0 / math.sqrt(2.0)) * S3 * (b * bv - v / 3.0)


def tor_conv_tp(x, t1, w):
    """x [E,74], t1 [E,3], w [E,384] -> [E,64] = 32x0o | 32x0e (kernels.hip bond_head)."""
    E = x.shape[0]
    x1o, x1e = x[:, 32:50].reshape(E, 6, 3), x[:, 50:68].reshape(E, 6, 3)
    wa, wb = w[:, :192].reshape(E, 6, 32), w[:, 192:].reshape(E, 6, 32)
    pw = math.sqrt(1 / 6)
    da = (x1o * t1[:, None, :]).sum(-1) / S3
    db = (x1e * t1[:, None, :]).sum(-1) / S3
    return np.concatenate([pw * np.einsum("euw,eu->ew", wb, db), pw * np.einsum("euw,eu->ew", wa, da)], 1)


def horn_rotation(A, B):
    """Proper rotation R (and t) minimising |R A + t - B| via Horn's quaternion method with Jacobi sweeps (fp64)."""
    ca, cb = A.mean(0), B.mean(0)
    S = (A - ca).T @ (B - cb)
    Sxx, Sxy, Sxz, Syx, Syy, Syz, Szx, Szy, Szz = S.reshape(-1)
    N = np.array([[Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx],
                  [Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz],
                  [Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy],
                  [Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz]])
    V = np.eye(4)
    for _ in range(12):
        for p in range(3):
            for q in range(p + 1, 4):
                if abs(N[p, q]) < 1e-300:
                    continue
                th = (N[q, q] - N[p, p]) / (2 * N[p, q])
                t = (1.0 if th >= 0 else -1.0) / (abs(th) + math.sqrt(th * th + 1))
                c = 1 / math.sqrt(t * t + 1)
                s = t * c
                J = np.eye(4)
                J[p, p] = J[q, q] = c
                J[p, q], J[q, p] = s, -s
                N = J.T @ N @ J
                V = V @ J
    k = int(np.argmax(np.diag(N)))
    w, x, y, z = V[:, k]
    R = np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])
    return R, cb - R @ ca


def test_final_conv_formula():
    g = torch.Generator().manual_seed(0)
    E = 7
    x, vec, w = torch.randn(E, 74, generator=g), torch.randn(E, 3, generator=g), torch.randn(E, 124, generator=g)
    tp = e3.FullyConnectedTensorProduct("32x0e+6x1o+6x1e+6x0o", "1x0e+1x1o", "2x1o+2x1e")
    assert tp.weight_numel == 124
    ref = tp(x.double(), e3.sh_l1(vec.double()), w.double()).numpy()
    v = torch.nn.functional.normalize(vec.double(), dim=-1).numpy()
    np.testing.assert_allclose(final_conv_tp(x.double().numpy(), v, w.double().numpy()), ref, rtol=1e-10, atol=1e-12)


def test_tor_conv_formula():
    g = torch.Generator().manual_seed(1)
    E = 9
    x, vec, bvec, w = (torch.randn(E, 74, generator=g).double(), torch.randn(E, 3, generator=g).double(),
                       torch.randn(E, 3, generator=g).double(), torch.randn(E, 384, generator=g).double())
    ftp = e3.FullTensorProduct("1x0e+1x1o", "2e")
    sh_full = ftp(e3.sh_l1(vec), e3.sh_l2(bvec))
    assert str(ftp.irreps_out[0][1]) == "1o" and sh_full.shape[1] == 20
    tp = e3.FullyConnectedTensorProduct("32x0e+6x1o+6x1e+6x0o", ftp.irreps_out, "32x0o+32x0e")
    assert tp.weight_numel == 384 and len(tp.instructions) == 2
    ref = tp(x, sh_full, w).numpy()
    v = torch.nn.functional.normalize(vec, dim=-1).numpy()
    b = torch.nn.functional.normalize(bvec, dim=-1).numpy()
    t1 = tor_t1(v, b)
    np.testing.assert_allclose(t1, sh_full[:, :3].numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(tor_conv_tp(x.numpy(), t1, w.numpy()), ref, rtol=1e-10, atol=1e-12)


def test_horn_matches_svd_kabsch():
    g = torch.Generator().manual_seed(2)
    A = torch.randn(4, 11, 3, generator=g)
    Rt = pr.axis_angle_to_matrix(torch.randn(4, 3, generator=g))
    B = torch.bmm(A, Rt.transpose(1, 2)) + torch.randn(4, 1, 3, generator=g) + 0.1 * torch.randn(4, 11, 3, generator=g)
    B[3] = A[3] * torch.tensor([1.0, 1.0, -1.0]) + 0.01 * torch.randn(11, 3, generator=g)   # reflection case
    R, t = pr.kabsch_batch(A.double(), B.double())
    for k in range(4):
        Rh, th = horn_rotation(A[k].double().numpy(), B[k].double().numpy())
        np.testing.assert_allclose(Rh, R[k].numpy(), atol=1e-9)
        np.testing.assert_allclose(th, t[k, :, 0].numpy(), atol=1e-9)


def test_f32_split_arithmetic_is_fp32_grade():
    """The OpsBf16x3 operand policy (csrc/tp_conv_dev.h): x = hi + mid + lo with three bf16 planes is EXACT for fp32 inputs, and the
    six plane products kept (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) with fp32 accumulation reproduce a K = 96 dot product as
    accurately as plain fp32 accumulation does -- both measured against float64."""
    import numpy as np

    def bf16(x):   # round-to-nearest-even to 8 significand bits, returned as float32
        u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)

    def split(x):
        h = bf16(x)
        m = bf16(x - h)
        lo = bf16(x - h - m)
        return h, m, lo

    rng = np.random.default_rng(0)
    a = (rng.normal(size=(512, 96)) * np.exp(rng.normal(size=(512, 96)))).astype(np.float32)   # wide dynamic range
    b = rng.normal(size=(96, 64)).astype(np.float32)
    ah, am, al = split(a)
    bh, bm, bl = split(b)
    assert np.array_equal((ah.astype(np.float64) + am + al).astype(np.float32), a)      # the split loses nothing
    assert np.array_equal((bh.astype(np.float64) + bm + bl).astype(np.float32), b)
    truth = a.astype(np.float64) @ b.astype(np.float64)
    # fp32 accumulation over k (what v_mfma_f32_32x32x2_f32 does, up to its internal order)
    acc32 = np.zeros((512, 64), dtype=np.float32)
    for k in range(96):
        acc32 += a[:, k:k + 1] * b[k:k + 1, :]
    # split policy: per 16-wide k-step the six plane products, smallest first, exact products, fp32 accumulation
    accx3 = np.zeros((512, 64), dtype=np.float32)
    for k0 in range(0, 96, 16):
        sl = slice(k0, k0 + 16)
        for pa, pb in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):
            accx3 += (pa[:, sl].astype(np.float64) @ pb[sl, :].astype(np.float64)).astype(np.float32)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    e32 = np.abs(acc32 - truth) / scale
    ex3 = np.abs(accx3 - truth) / scale
    assert ex3.max() < 4e-7 and e32.max() < 8e-7            # a few ulp of fp32 relative to sum |a||b| (measured 2.9e-7 / 3.9e-7)
    assert ex3.mean() <= 1.5 * e32.mean() + 1e-9            # no worse on average than plain fp32 accumulation
