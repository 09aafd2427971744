"""Second, build-independent pin of the e3nn constants the oracle restates.

TEST INFRASTRUCTURE ONLY (oracle).  e3nn==0.5.0 (reference environment.yml:129) is not installable here, so oracle/e3nn_ref.py restates
its real Wigner-3j algorithm with its OWN Clebsch-Gordan code (Racah's formula in exact fractions) and its OWN real<->complex change of
basis, and confidence_bootstrapping_amd/e3nn_constants.py hard-wires closed forms of the same tensors.  This script re-derives every
real-basis Wigner 3j with l1, l2, l3 <= 2 (all the triples reachable by the score model's heads -- models/score_model.py:245-274 --
and the confidence model's lmax = 2 FullyConnectedTensorProducts -- models/all_atom_score_model.py with sh_lmax = 2) WITHOUT using any
of that code:

  1. SU(2) Clebsch-Gordan coefficients <l1 m1 l2 m2 | l3 m3> from sympy.physics.wigner.clebsch_gordan (exact arithmetic);
  2. the complex -> real change of basis FITTED, per l, from sympy's complex spherical harmonics Ynm (Condon-Shortley phase) and the
     real polynomials e3nn DOCUMENTS for o3.spherical_harmonics (l = 1: (x, y, z); l = 2: (xz, xy, y^2 - (x^2 + z^2)/2, yz, (z^2 - x^2)/2),
     polar axis y) by solving  Y^c_mu(r) = sum_m U[mu, m] R_m(r)  at random points -- no hand-written sign table;
  3. e3nn's published recipe for the real 3j (o3/_wigner.py): Q_l = (-i)^l U_l;  C = Re einsum("ij,kl,mn,ikn->jlm", Q1, Q2, conj(Q3^T), CG),
     normalised to unit Frobenius norm.

What this pins: magnitudes, the relative signs inside every tensor AND the overall sign e3nn's recipe produces, independent of the
build's CG / basis code.  What it cannot pin: that recipe itself (taken from e3nn's published source: the einsum and the (-i)^l factor).
The two triples the reference implements by hand -- (1,1,0) = delta/sqrt3 and (1,1,1) = epsilon/sqrt6, models/tensor_layers.py:76-82,
golden g1 -- anchor the recipe's sign independently.

  python oracle/pin_wigner_sympy.py            prints the table and writes tests/golden/g20_wigner3j_sympy.npz
tests/test_oracle_e3nn.py::test_wigner3j_matches_the_sympy_derivation runs derive() again and compares all three sources to 1e-12.
"""
import itertools
import os

import numpy as np

LMAX = 2


def real_basis(l, xyz):
    """e3nn's documented real spherical harmonics (unit-sphere polynomials, 'norm'-free: only the span and the signs matter; each row is
    L2-normalised over the sphere numerically by the caller).  xyz: [n, 3] unit vectors -> [n, 2l+1]."""
    x, y, z = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    if l == 0:
        return np.ones((len(x), 1))
    if l == 1:
        return np.stack([x, y, z], 1)
    if l == 2:
        s3 = np.sqrt(3.0)
        return np.stack([s3 * x * z, s3 * x * y, y * y - 0.5 * (x * x + z * z), s3 * y * z, 0.5 * s3 * (z * z - x * x)], 1)
    raise NotImplementedError(l)


def complex_sh(l, xyz):
    """sympy's Ynm(l, m, theta, phi), m = -l..l, with the POLAR AXIS y (e3nn's convention): the frame (x', y', z') = (z, x, y) is a cyclic
    -- proper -- rotation of (x, y, z).  -> [n, 2l+1] complex."""
    import sympy as sp
    th, ph = sp.symbols("theta phi", real=True)
    xp, yp, zp = xyz[:, 2], xyz[:, 0], xyz[:, 1]
    theta, phi = np.arccos(np.clip(zp, -1, 1)), np.arctan2(yp, xp)
    cols = []
    for m in range(-l, l + 1):
        f = sp.lambdify((th, ph), sp.Ynm(l, m, th, ph).expand(func=True), "numpy")
        cols.append(np.asarray(f(theta, phi), dtype=np.complex128) * np.ones_like(theta))
    return np.stack(cols, 1)


def fitted_change_of_basis(l, rng):
    """U[mu, m] with Y^c_mu = sum_m U[mu, m] R_m, R = the documented real basis scaled so that U is unitary.  Least squares over random
    points; the residual is asserted to vanish (the real polynomials span the same space)."""
    v = rng.normal(size=(400, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    R, Y = real_basis(l, v), complex_sh(l, v)
    # orthonormalise the scale of R against Y: sphere average of |Y_mu|^2 is 1 / (4 pi); scale each R_m to the same mean square
    # (analytic: <R_m^2> = 1 / (2l + 1) for these polynomials up to their common factor; done numerically on a Lebedev-free way: exact
    # ratios from the fit itself -- U is made unitary by scaling its columns)
    U, res, rank, _ = np.linalg.lstsq(R.astype(np.complex128), Y, rcond=None)       # R @ U = Y  ->  U[m, mu]
    assert rank == 2 * l + 1 and np.abs(R @ U - Y).max() < 1e-12
    U = U.T                                                                          # [mu, m]
    U = U / np.linalg.norm(U, axis=0, keepdims=True)                                 # unit columns: the common scale of the R_m drops out
    assert np.abs(U.conj().T @ U - np.eye(2 * l + 1)).max() < 1e-12, "fitted change of basis is not unitary"
    return U


def su2_cg(l1, l2, l3):
    from sympy.physics.wigner import clebsch_gordan
    out = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            m3 = m1 + m2
            if abs(m3) <= l3:
                out[l1 + m1, l2 + m2, l3 + m3] = float(clebsch_gordan(l1, l2, l3, m1, m2, m3))
    return out


def derive(seed=0):
    """{(l1, l2, l3): real Wigner 3j [2l1+1, 2l2+1, 2l3+1], unit Frobenius norm} for every triangle-admissible triple with l <= LMAX"""
    rng = np.random.default_rng(seed)
    Q = {l: (-1j) ** l * fitted_change_of_basis(l, rng) for l in range(LMAX + 1)}
    out = {}
    for l1, l2, l3 in itertools.product(range(LMAX + 1), repeat=3):
        if not abs(l1 - l2) <= l3 <= l1 + l2:
            continue
        c = np.einsum("ij,kl,mn,ikn->jlm", Q[l1], Q[l2], np.conj(Q[l3].T), su2_cg(l1, l2, l3).astype(np.complex128))
        assert np.abs(c.imag).max() < 1e-12, (l1, l2, l3)
        c = c.real
        out[(l1, l2, l3)] = c / np.linalg.norm(c)
    return out


def main():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import e3nn_ref
    from confidence_bootstrapping_amd import e3nn_constants
    tab = derive()
    worst = 0.0
    print("l1 l2 l3   |oracle - sympy|   |closed form - sympy|   nonzeros")
    for ls, c in sorted(tab.items()):
        d_or = np.abs(e3nn_ref.wigner_3j(*ls) - c).max()
        cf = e3nn_constants.w3j_closed_form(*ls)
        d_cf = np.abs(cf - c).max() if cf is not None else float("nan")
        worst = max(worst, d_or, 0.0 if cf is None else d_cf)
        print(f"{ls[0]:2d} {ls[1]:2d} {ls[2]:2d}   {d_or:.2e}            {d_cf:.2e}               {int((np.abs(c) > 1e-12).sum())}")
    # the two hand-written triples of the reference (models/tensor_layers.py:76-82): the recipe's sign agrees with them
    eps = np.zeros((3, 3, 3))
    for i, j, k in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):
        eps[i, j, k], eps[j, i, k] = 1.0, -1.0
    assert np.abs(tab[(1, 1, 1)] - eps / np.sqrt(6)).max() < 1e-12 and np.abs(tab[(1, 1, 0)][:, :, 0] - np.eye(3) / np.sqrt(3)).max() < 1e-12
    assert worst < 1e-12, worst
    path = os.path.join(root, "tests", "golden", "g20_wigner3j_sympy.npz")
    np.savez(path, **{"w3j_%d_%d_%d" % ls: c for ls, c in tab.items()})
    print("max deviation", worst, "->", path)


if __name__ == "__main__":
    main()
