"""CPU emulation of fctp_conv_kernel's tile algorithm (csrc/fctp_conv.hip) on the REAL packed weight stream produced by
the C library (cbd_conf_pack_stream, host-only), compared with the oracle's FCBlock + e3nn FullyConnectedTensorProduct
(lmax = 2).  Pins, without a GPU: the e3nn instruction/weight layout, the folded path weights and Wigner-3j constants,
the k-permutation between the two GEMMs (K = 72) and the tile row layouts.  MFMA operand layout as in
tests/test_pack_emulation.py."""
import numpy as np
import pytest
import torch

from oracle import e3nn_ref as e3
from oracle.confidence_ref import ConfConfig, SH_IRREPS

KS, TILE_W = 36, 36 * 64
NS, C1O, C1E, C0O, STRIDE = 24, 24, 42, 60, 84


def row_of(reg, hf):
    return (reg & 3) + 8 * (reg >> 2) + 4 * hf


def gemm_tile(tile, B):
    A = tile[0].reshape(9, 64, 4)
    D = np.zeros((32, 32))
    for s in range(KS):
        a = A[s >> 2, :, s & 3]
        for hf in range(2):
            D += np.outer(a[32 * hf:32 * hf + 32], B[s][32 * hf:32 * hf + 32])
    D += tile[1][:, None]
    acc = np.zeros((16, 64))
    for lane in range(64):
        for reg in range(16):
            acc[reg, lane] = D[row_of(reg, lane >> 5), lane & 31]
    return acc


def shape(IN, OUT):
    n1o, n1e, n0o = (6 if IN >= 1 else 0), (6 if IN >= 2 else 0), (24 if IN >= 3 else 0)
    fan0e, fan1o = 24 + n1o, 24 + 2 * n1o + n1e
    fan1e = n1o + 2 * n1e + n0o if OUT >= 2 else 0
    fan0o = n1e + n0o if OUT >= 3 else 0
    return dict(n1o=n1o, n1e=n1e, n0o=n0o, fan0e=fan0e, fan1o=fan1o, fan1e=fan1e, fan0o=fan0o,
                g0e=(fan0e + 3) // 4, t1o=(fan1o + 4) // 5, t1e=(fan1e + 4) // 5, g0o=(fan0o + 3) // 4)


def cross(a, n):
    return np.array([a[1] * n[2] - a[2] * n[1], a[2] * n[0] - a[0] * n[2], a[0] * n[1] - a[1] * n[0]])


def quad(a, n):
    return n * float(a @ n) - a / 3.0


def emulate(stream, IN, OUT, xin, xrow, n):
    S = shape(IN, OUT)
    dense = lambda fan, g: 1 <= fan - 4 * g <= 2          # tail group of <= 2 mids: two denser tiles (conf_common.h::sc_tail_dense)
    sc_tiles = lambda fan: 0 if fan == 0 else 3 * ((fan + 3) // 4 - 1) + (2 if dense(fan, (fan + 3) // 4 - 1) else 3)
    # merged tails (conf_common.h::FctpShape::merged): block 0e has no tile B of its own, its octet 2 rides in slots 2, 3 of block 0o's
    merged = OUT >= 3 and dense(S["fan0e"], S["g0e"] - 1) and dense(S["fan0o"], S["g0o"] - 1)
    # merged vector tails (FctpShape::vmerged): block 1o's tail mids sit behind block 1e's in 1e's last tile
    r1o, r1e = S["fan1o"] % 5, S["fan1e"] % 5
    vmerged = OUT >= 2 and r1o > 0 and r1e > 0 and r1o + r1e <= 5
    nt = 3 + sc_tiles(S["fan0e"]) + S["t1o"] + S["t1e"] + sc_tiles(S["fan0o"]) - int(merged) - int(vmerged)
    assert stream.size == (nt + 1) * TILE_W + nt * 32
    wts, bias = stream[:nt * TILE_W].reshape(nt, TILE_W), stream[(nt + 1) * TILE_W:].reshape(nt, 32)
    tiles = [(wts[k], bias[k]) for k in range(nt)]
    lanes = np.arange(64)
    j, hf = lanes & 31, lanes >> 5
    Bx = np.zeros((KS, 64))
    for s in range(KS):
        Bx[s] = xin[j, 24 * (s // 12) + 12 * hf + (s % 12)]
    T = 0
    h1 = np.zeros((KS, 64))
    for m in range(2):
        acc = gemm_tile(tiles[T], Bx); T += 1
        h1[16 * m:16 * m + 16] = np.maximum(acc, 0)
    acc = gemm_tile(tiles[T], Bx); T += 1
    h1[32:36] = np.maximum(acc[:4], 0)
    assert np.all(acc[4:] == 0)              # hidden rows >= 72 are zero rows of the stream

    v3 = lambda e, c0: xrow[e, c0:c0 + 3]

    def mid0e(e, i):
        if i < 24: return xrow[e, i]
        if i < S["fan0e"]: return float(v3(e, C1O + 3 * (i - 24)) @ n[e])
        return 0.0

    def mid1o(e, i):
        if i < 24: return xrow[e, i] * n[e]
        if i < 24 + S["n1o"]: return v3(e, C1O + 3 * (i - 24))
        if i < 24 + 2 * S["n1o"]: return quad(v3(e, C1O + 3 * (i - 24 - S["n1o"])), n[e])
        if i < S["fan1o"]: return cross(v3(e, C1E + 3 * (i - 24 - 2 * S["n1o"])), n[e])
        return np.zeros(3)

    def mid1e(e, i):
        if i < S["n1o"]: return cross(v3(e, C1O + 3 * i), n[e])
        if i < S["n1o"] + S["n1e"]: return v3(e, C1E + 3 * (i - S["n1o"]))
        if i < S["n1o"] + 2 * S["n1e"]: return quad(v3(e, C1E + 3 * (i - S["n1o"] - S["n1e"])), n[e])
        if i < S["fan1e"]: return xrow[e, C0O + (i - S["n1o"] - 2 * S["n1e"])] * n[e]
        return np.zeros(3)

    def mid0o(e, i):
        if i < S["n1e"]: return float(v3(e, C1E + 3 * i) @ n[e])
        if i < S["fan0o"]: return xrow[e, C0O + (i - S["n1e"])]
        return 0.0

    out = np.zeros((32, STRIDE))

    keeps = {}
    tail0e = {}

    def scalar_block(ngroups, mid, col0, fan):
        nonlocal T
        keep = keeps[col0] = np.zeros((12, 64))
        is0e = col0 == 0
        for g in range(ngroups):
            if dense(fan, g):         # tile A: slot i = (mid i & 1, output octet i >> 1); tile B: slots 0, 1 = octet 2
                acc = gemm_tile(tiles[T], h1); T += 1
                for lane in range(64):
                    for i in range(4):
                        m = mid(lane & 31, 4 * g + (i & 1))
                        for c in range(4):
                            keep[4 * (i >> 1) + c, lane] += m * acc[4 * i + c, lane]
                if merged and is0e:
                    tail0e["g"] = g
                    continue
                acc = gemm_tile(tiles[T], h1); T += 1
                for lane in range(64):
                    if not merged:
                        assert np.all(acc[8:, lane] == 0)
                    for i in range(2):
                        m = mid(lane & 31, 4 * g + i)
                        for c in range(4):
                            keep[8 + c, lane] += m * acc[4 * i + c, lane]
                        if merged:
                            m0 = mid0e(lane & 31, 4 * tail0e["g"] + i)
                            for c in range(4):
                                keeps[0][8 + c, lane] += m0 * acc[4 * (2 + i) + c, lane]
                continue
            for q in range(3):
                acc = gemm_tile(tiles[T], h1); T += 1
                for lane in range(64):
                    for i in range(4):
                        m = mid(lane & 31, 4 * g + i)
                        for c in range(4):
                            keep[4 * q + c, lane] += m * acc[4 * i + c, lane]
    def write_scalar(col0):
        keep = keeps[col0]
        for lane in range(64):
            for q in range(3):
                for c in range(4):
                    out[lane & 31, col0 + 8 * q + c + 4 * (lane >> 5)] = keep[4 * q + c, lane]

    vkeeps = {}

    def vec_block(ntile, mid, col0, fan, guest=None):
        nonlocal T
        keep = vkeeps[col0] = np.zeros((3, 3, 64))
        for t in range(ntile):
            acc = gemm_tile(tiles[T], h1); T += 1
            for lane in range(64):
                for q in range(5):
                    if 5 * t + q >= fan:
                        continue
                    m = mid(lane & 31, 5 * t + q)
                    for o in range(3):
                        keep[o, :, lane] += m * acc[3 * q + o, lane]
                if guest is not None and t == ntile - 1:          # block 1o's tail mids in the slots behind this block's
                    gn, gbase, gslot = guest
                    for g in range(gn):
                        m = mid1o(lane & 31, gbase + g)
                        for o in range(3):
                            vkeeps[C1O][o, :, lane] += m * acc[3 * (gslot + g) + o, lane]

    def write_vec(col0):
        keep = vkeeps[col0]
        for lane in range(64):
            for o in range(3):
                oo = 3 * (lane >> 5) + o
                out[lane & 31, col0 + 3 * oo:col0 + 3 * oo + 3] = keep[o, :, lane]

    scalar_block(S["g0e"], mid0e, 0, S["fan0e"])
    vec_block(S["t1o"] - int(vmerged), mid1o, C1O, 5 * (S["t1o"] - 1) if vmerged else S["fan1o"])
    if OUT >= 2:
        vec_block(S["t1e"], mid1e, C1E, S["fan1e"], (r1o, 5 * (S["t1o"] - 1), r1e) if vmerged else None)
        write_vec(C1E)
    write_vec(C1O)
    if OUT >= 3:
        scalar_block(S["g0o"], mid0o, C0O, S["fan0o"])
        write_scalar(C0O)
    write_scalar(0)
    assert T == len(tiles)
    return out


@pytest.mark.parametrize("IN,OUT", [(0, 1), (1, 2), (2, 3), (3, 3)])
def test_packed_stream_reproduces_fcblock_and_e3nn_tensor_product(IN, OUT):
    from confidence_bootstrapping_amd.engine import pack_fctp_stream, load_library
    from confidence_bootstrapping_amd.all_atom_score_model import fctp_weight_numel
    lib = load_library()
    g = torch.Generator().manual_seed(100 + 10 * IN + OUT)
    seq = ConfConfig().irrep_seq
    in_irr, out_irr = seq[IN], seq[OUT]
    tp = e3.FullyConnectedTensorProduct(in_irr, SH_IRREPS, out_irr, shared_weights=False)
    W = tp.weight_numel
    assert W == {(0, 1): 720, (1, 2): 972, (2, 3): 1224, (3, 3): 1944}[(IN, OUT)] == fctp_weight_numel(in_irr, "1x0e+1x1o+1x2e", out_irr)
    w1, b1 = torch.randn(72, 72, generator=g) / 8, torch.randn(72, generator=g) / 4
    w2, b2 = torch.randn(W, 72, generator=g) / 8, torch.randn(W, generator=g) / 4
    stream = pack_fctp_stream(IN, OUT, w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy())
    assert stream.size == lib.cbd_conf_stream_floats(IN, OUT)
    E = 32
    in_dim, out_dim = e3.Irreps(in_irr).dim, e3.Irreps(out_irr).dim
    xin = torch.randn(E, 72, generator=g)
    xd = torch.randn(E, in_dim, generator=g)
    vec = torch.randn(E, 3, generator=g)
    hid = torch.relu(xin.double() @ w1.double().T + b1.double())
    tpw = hid @ w2.double().T + b2.double()
    sh = e3.spherical_harmonics(SH_IRREPS, vec.double())
    ref = tp(xd.double(), sh, tpw).numpy()
    xrow = np.zeros((E, STRIDE))
    xrow[:, :in_dim] = xd.double().numpy()
    n = torch.nn.functional.normalize(vec.double(), dim=-1).numpy()
    got = emulate(stream.astype(np.float64), IN, OUT, xin.double().numpy(), xrow, n)
    np.testing.assert_allclose(got[:, :out_dim], ref, rtol=2e-5, atol=2e-5)
    assert np.all(got[:, out_dim:] == 0)
