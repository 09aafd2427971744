"""CPU restatement of the e3nn==0.5.0 semantics the reference's hot path relies on.

TEST INFRASTRUCTURE ONLY (oracle).  Nothing under oracle/ may be imported by the product
package `confidence_bootstrapping_amd`; only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py use it, and only as the checker.

e3nn is an un-vendored dependency of the reference (environment.yml:129 `e3nn==0.5.0`) and is
not installable here, so its published algorithm is restated.  PARITY UNPINNED at this
boundary: the reference has no tests/golden vectors and e3nn cannot be imported to pin these
functions.  Second pin since round 6 (oracle/pin_wigner_sympy.py, tests/golden/g20_wigner3j_sympy.npz): every real Wigner 3j
with l <= 2 re-derived from sympy's exact Clebsch-Gordan coefficients and a change of basis fitted from sympy's Ynm against the
real polynomials e3nn documents -- no code shared with this file -- agrees with wigner_3j() below to 1e-15, signs included.
Internal consistency is checked as well (tests/test_oracle_e3nn.py):
  * O(3) equivariance of SH / tensor products under random rotations + inversion,
  * w3j(1,1,0) = delta/sqrt(3) and w3j(1,1,1) = eps/sqrt(6) with the SAME sign the reference's
    own hand-written FasterTensorProduct uses (models/tensor_layers.py:76-82), which was
    written as a drop-in for e3nn's FullyConnectedTensorProduct.

Call sites in the reference that this file stands in for:
  o3.spherical_harmonics      models/score_model.py:436,519,536,581-582,647,661
  o3.FullyConnectedTensorProduct  models/tensor_layers.py:185  (final_conv, tor_bond_conv)
  o3.FullTensorProduct        models/score_model.py:265,437
  e3nn.nn.BatchNorm           models/tensor_layers.py:193,211-212
  o3.Irreps                   models/tensor_layers.py:50-56, models/score_model.py:72
"""
from __future__ import annotations

import math
import re
from fractions import Fraction
from functools import lru_cache
from math import factorial

import numpy as np
import torch


# ----------------------------------------------------------------------------- Irreps (mini)
class Irrep(tuple):
    """(l, p) with p = +1 (even 'e') / -1 (odd 'o').  Sort order follows e3nn: by l, then odd < even
    ... e3nn orders irreps of equal l with p=-(-1)**l first; only equality/str are relied on here."""

    def __new__(cls, l, p=None):
        if isinstance(l, Irrep):
            return l
        if isinstance(l, str):
            m = re.fullmatch(r"\s*(\d+)([eo])\s*", l)
            return tuple.__new__(cls, (int(m.group(1)), 1 if m.group(2) == "e" else -1))
        return tuple.__new__(cls, (int(l), int(p)))

    l = property(lambda s: s[0])
    p = property(lambda s: s[1])
    dim = property(lambda s: 2 * s[0] + 1)

    def __str__(self):
        return f"{self[0]}{'e' if self[1] == 1 else 'o'}"

    __repr__ = __str__

    def __mul__(self, other):
        other = Irrep(other)
        return [Irrep(l, self.p * other.p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]


class Irreps(tuple):
    """Tuple of (mul, Irrep).  Supports what the reference touches: parse, iterate, slices(), dim, ==."""

    def __new__(cls, spec=None):
        if isinstance(spec, Irreps):
            return spec
        out = []
        if isinstance(spec, str):
            for term in spec.split("+"):
                term = term.strip()
                if not term:
                    continue
                if "x" in term:
                    mul, ir = term.split("x")
                    out.append((int(mul), Irrep(ir)))
                else:
                    out.append((1, Irrep(term)))
        elif spec is not None:
            for item in spec:
                if isinstance(item, (str, Irrep)):
                    out.append((1, Irrep(item)))
                else:
                    mul, ir = item
                    out.append((int(mul), Irrep(*ir) if not isinstance(ir, (Irrep, str)) else Irrep(ir)))
        return tuple.__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, Irrep(l, p ** l)) for l in range(lmax + 1)])

    @property
    def dim(self):
        return sum(m * ir.dim for m, ir in self)

    @property
    def num_irreps(self):
        return sum(m for m, _ in self)

    def slices(self):
        s, i = [], 0
        for m, ir in self:
            s.append(slice(i, i + m * ir.dim))
            i += m * ir.dim
        return s

    def __str__(self):
        return "+".join(f"{m}x{ir}" for m, ir in self)

    __repr__ = __str__


# ----------------------------------------------------------------------------- spherical harmonics
def sh_l1(vec: torch.Tensor) -> torch.Tensor:
    """o3.spherical_harmonics('1x0e+1x1o', vec, normalize=True, normalization='component')
    -> [1, sqrt3*x, sqrt3*y, sqrt3*z] of the unit vector (zero vector -> NaN like e3nn's normalize? e3nn
    uses F.normalize (eps=1e-12) so a zero vector maps to zeros, which is restated here)."""
    n = torch.nn.functional.normalize(vec, dim=-1)
    return torch.cat([torch.ones_like(n[..., :1]), math.sqrt(3.0) * n], dim=-1)


def sh_l2(vec: torch.Tensor) -> torch.Tensor:
    """o3.spherical_harmonics('2e', vec, normalize=True, normalization='component') : 5 comps."""
    n = torch.nn.functional.normalize(vec, dim=-1)
    x, y, z = n[..., 0], n[..., 1], n[..., 2]
    s3 = math.sqrt(3.0)
    sh = torch.stack([s3 * x * z, s3 * x * y, y * y - 0.5 * (x * x + z * z), s3 * y * z,
                      (s3 / 2.0) * (z * z - x * x)], dim=-1)
    return math.sqrt(5.0) * sh


def spherical_harmonics(irreps, vec, normalize=True, normalization="component"):
    assert normalize and normalization == "component"
    irreps = Irreps(irreps) if not isinstance(irreps, Irreps) else irreps
    ls = [ir.l for _, ir in irreps]
    if ls == [0, 1]:
        return sh_l1(vec)
    if ls == [2]:
        return sh_l2(vec)
    if ls == [0, 1, 2]:
        return torch.cat([sh_l1(vec), sh_l2(vec)], -1)
    raise NotImplementedError(str(irreps))


# ----------------------------------------------------------------------------- Wigner 3j (real basis)
def _su2_cg_coeff(j1, m1, j2, m2, j3, m3):
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))

    def f(n):
        return factorial(round(n))

    c = ((2.0 * j3 + 1.0) * Fraction(
        f(j3 + j1 - j2) * f(j3 - j1 + j2) * f(j1 + j2 - j3) * f(j3 + m3) * f(j3 - m3),
        f(j1 + j2 + j3 + 1) * f(j1 - m1) * f(j1 + m1) * f(j2 - m2) * f(j2 + m2))) ** 0.5
    s = 0
    for v in range(vmin, vmax + 1):
        s += (-1) ** int(v + j2 + m2) * Fraction(
            f(j2 + j3 + m1 - v) * f(j1 - m1 + v),
            f(v) * f(j3 - j1 + j2 - v) * f(j3 + m3 - v) * f(v + j1 - j2 - m3))
    return float(c * s)


def _su2_cg(j1, j2, j3):
    mat = np.zeros((2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1))
    for m1 in range(-j1, j1 + 1):
        for m2 in range(-j2, j2 + 1):
            if abs(m1 + m2) <= j3:
                mat[j1 + m1, j2 + m2, j3 + m1 + m2] = _su2_cg_coeff(j1, m1, j2, m2, j3, m1 + m2)
    return mat


def _real_to_complex(l):
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = 1 / 2 ** 0.5
        q[l + m, l - abs(m)] = -1j / 2 ** 0.5
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m / 2 ** 0.5
        q[l + m, l - abs(m)] = 1j * (-1) ** m / 2 ** 0.5
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1, l2, l3) -> np.ndarray:
    """Real-basis Wigner 3j with unit Frobenius norm (e3nn o3.wigner_3j algorithm: SU(2) CG
    conjugated by the real<->complex change of basis, real part, normalised)."""
    q1, q2, q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    c = _su2_cg(l1, l2, l3).astype(np.complex128)
    c = np.einsum("ij,kl,mn,ikn->jlm", q1, q2, np.conj(q3.T), c)
    assert np.abs(c.imag).max() < 1e-9
    c = c.real
    return c / np.linalg.norm(c)


# ----------------------------------------------------------------------------- tensor products
class FullyConnectedTensorProduct(torch.nn.Module):
    """o3.FullyConnectedTensorProduct(in1, in2, out, shared_weights=False): mode 'uvw',
    irrep_normalization='component', path_normalization='element', weights supplied per edge.
    Instruction order: for i1 in in1, for i2 in in2, for io in out if ir_out in ir1*ir2.
    Per-edge weight layout: instructions concatenated, each [mul1, mul2, mul_out] row-major."""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, shared_weights=False):
        super().__init__()
        assert not shared_weights
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        self.instructions = []
        for i1, (m1, ir1) in enumerate(self.irreps_in1):
            for i2, (m2, ir2) in enumerate(self.irreps_in2):
                for io, (mo, iro) in enumerate(self.irreps_out):
                    if iro in ir1 * ir2:
                        self.instructions.append((i1, i2, io))
        fan = {}
        for (i1, i2, io) in self.instructions:
            fan[io] = fan.get(io, 0) + self.irreps_in1[i1][0] * self.irreps_in2[i2][0]
        self.path_weight = [math.sqrt(self.irreps_out[io][1].dim / fan[io]) for (_, _, io) in self.instructions]
        self.weight_numel = sum(self.irreps_in1[i1][0] * self.irreps_in2[i2][0] * self.irreps_out[io][0]
                                for (i1, i2, io) in self.instructions)

    def forward(self, x1, x2, weight):
        s1, s2, so = self.irreps_in1.slices(), self.irreps_in2.slices(), self.irreps_out.slices()
        E = x1.shape[0]
        out = [torch.zeros(E, mo, iro.dim, dtype=x1.dtype) for mo, iro in self.irreps_out]
        off = 0
        for (i1, i2, io), pw in zip(self.instructions, self.path_weight):
            m1, ir1 = self.irreps_in1[i1]
            m2, ir2 = self.irreps_in2[i2]
            mo, iro = self.irreps_out[io]
            n = m1 * m2 * mo
            w = weight[:, off:off + n].reshape(E, m1, m2, mo)
            off += n
            a = x1[:, s1[i1]].reshape(E, m1, ir1.dim)
            b = x2[:, s2[i2]].reshape(E, m2, ir2.dim)
            c = torch.from_numpy(wigner_3j(ir1.l, ir2.l, iro.l)).to(x1.dtype)
            out[io] = out[io] + pw * torch.einsum("euvw,eui,evj,ijk->ewk", w, a, b, c)
        return torch.cat([o.reshape(E, -1) for o in out], dim=-1)


class FullTensorProduct(torch.nn.Module):
    """o3.FullTensorProduct(in1, in2): mode 'uvuv', no weights, path weight sqrt(2 l_out + 1),
    output irreps sorted by (l, p) with e3nn's ordering (for equal l: p = -(-1)^l ... only the order of the
    l=2 pair could differ; the hot path consumes the 1o block only, which is first either way)."""

    def __init__(self, irreps_in1, irreps_in2):
        super().__init__()
        self.irreps_in1, self.irreps_in2 = Irreps(irreps_in1), Irreps(irreps_in2)
        outs = []
        for i1, (m1, ir1) in enumerate(self.irreps_in1):
            for i2, (m2, ir2) in enumerate(self.irreps_in2):
                for iro in ir1 * ir2:
                    outs.append((m1 * m2, iro, i1, i2))
        # e3nn Irrep sort key: (l, -p * (-1)**l)  => for l even: e before o ; l odd: o before e
        order = sorted(range(len(outs)), key=lambda k: (outs[k][1].l, -outs[k][1].p * (-1) ** outs[k][1].l))
        self.paths = [outs[k] for k in order]
        self.irreps_out = Irreps([(m, ir) for (m, ir, _, _) in self.paths])

    def forward(self, x1, x2):
        s1, s2 = self.irreps_in1.slices(), self.irreps_in2.slices()
        E = x1.shape[0]
        res = []
        for (m, iro, i1, i2) in self.paths:
            m1, ir1 = self.irreps_in1[i1]
            m2, ir2 = self.irreps_in2[i2]
            a = x1[:, s1[i1]].reshape(E, m1, ir1.dim)
            b = x2[:, s2[i2]].reshape(E, m2, ir2.dim)
            c = torch.from_numpy(wigner_3j(ir1.l, ir2.l, iro.l)).to(x1.dtype)
            res.append(math.sqrt(iro.dim) * torch.einsum("eui,evj,ijk->euvk", a, b, c).reshape(E, -1))
        return torch.cat(res, dim=-1)


class BatchNorm(torch.nn.Module):
    """e3nn.nn.BatchNorm(irreps) (eps=1e-5, affine, normalization='component', reduce='mean').
    Eval: scalars (l=0,p=+1) (x-mean)*rsqrt(var+eps)*w + b ; every other irrep x*rsqrt(var+eps)*w,
    one statistic per multiplicity channel.  state_dict: weight[F], bias[Fs], running_mean[Fs], running_var[F]
    with F = sum of muls, Fs = sum of muls of 0e irreps."""

    def __init__(self, irreps, eps=1e-5):
        super().__init__()
        self.irreps = Irreps(irreps)
        nf = sum(m for m, _ in self.irreps)
        ns = sum(m for m, ir in self.irreps if ir.l == 0 and ir.p == 1)
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(nf))
        self.bias = torch.nn.Parameter(torch.zeros(ns))
        self.register_buffer("running_mean", torch.zeros(ns))
        self.register_buffer("running_var", torch.ones(nf))

    momentum = 0.1

    def forward(self, x):
        """Training mode restated from e3nn 0.5.0 e3nn/nn/_batchnorm.py: 0e fields are centred on their batch mean, every
        field is scaled by (batch mean of its squared components + eps)^-1/2, gradients flow through the statistics, and the
        running buffers move by `momentum` towards the (detached) batch values."""
        out, ix, iw, ib = [], 0, 0, 0
        new_mean, new_var = [], []
        for m, ir in self.irreps:
            d = ir.dim
            f = x[:, ix:ix + m * d].reshape(-1, m, d)
            ix += m * d
            scalar = ir.l == 0 and ir.p == 1
            if scalar:
                if self.training:
                    mean = f.mean(dim=(0, 2))
                    new_mean.append(mean.detach())
                else:
                    mean = self.running_mean[ib:ib + m]
                f = f - mean.reshape(1, m, 1)
            if self.training:
                var = f.pow(2).mean(dim=2).mean(dim=0)
                new_var.append(var.detach())
            else:
                var = self.running_var[iw:iw + m]
            scale = (var + self.eps).pow(-0.5) * self.weight[iw:iw + m]
            f = f * scale.reshape(1, m, 1)
            if scalar:
                f = f + self.bias[ib:ib + m].reshape(1, m, 1)
                ib += m
            iw += m
            out.append(f.reshape(-1, m * d))
        if self.training:
            with torch.no_grad():
                if new_mean:
                    self.running_mean.copy_((1 - self.momentum) * self.running_mean + self.momentum * torch.cat(new_mean))
                self.running_var.copy_((1 - self.momentum) * self.running_var + self.momentum * torch.cat(new_var))
        return torch.cat(out, dim=-1)
