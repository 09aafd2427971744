"""CPU emulation of the tp_conv kernel's tile algorithm on the REAL packed weight stream produced by the C
library (cbd_pack_conv_stream, host-only), compared with the oracle's FCBlock + FasterTensorProduct.

This pins, without a GPU: the weight re-packing (k-permutation between the two GEMMs, row layout of the vector
blocks, folded normalisation factors) and the lane/register index arithmetic the HIP kernel uses
(csrc/tp_conv.hip).  The MFMA itself is emulated as an exact matrix product with its documented operand layout:
  v_mfma_f32_32x32x2_f32: A[i = lane&31][k = lane>>5], B[k = lane>>5][j = lane&31],
                          D[row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)][col = lane&31].
"""
import numpy as np
import pytest
import torch

from oracle import score_ref as sr
from oracle.e3nn_ref import sh_l1

KSTEPS, TILE_W, TILE = 48, 48 * 64, 48 * 64 + 32
COL_1O, COL_1E, COL_0O = 32, 50, 68


def row_of(reg, hf):
    return (reg & 3) + 8 * (reg >> 2) + 4 * hf


def gemm_tile(tile, B):
    """tile: (weights [TILE_W], bias [32]); B[s][lane] -> acc[reg][lane] exactly as gemm_tile() in tp_conv.hip."""
    A = tile[0].reshape(12, 64, 4)                  # [sg][lane][4]
    bias = tile[1]
    D = np.zeros((32, 32), dtype=np.float64)        # [row][edge]
    for s in range(KSTEPS):
        a = A[s >> 2, :, s & 3]                     # per lane
        for hf in range(2):
            D += np.outer(a[32 * hf:32 * hf + 32], B[s][32 * hf:32 * hf + 32])   # A[i][k=hf] * B[k=hf][j]
    D += bias[:, None]
    acc = np.zeros((16, 64))
    for lane in range(64):
        for reg in range(16):
            acc[reg, lane] = D[row_of(reg, lane >> 5), lane & 31]
    return acc


def shape(IN, OUT):
    n1o, n1e, n0o = (6 if IN >= 1 else 0), (6 if IN >= 2 else 0), (6 if IN >= 3 else 0)
    fan0e, fan1o = 32 + n1o, 32 + n1o + n1e
    fan1e = n1o + n1e + n0o if OUT >= 2 else 0
    fan0o = n1e + n0o if OUT >= 3 else 0
    return dict(n1o=n1o, n1e=n1e, n0o=n0o, fan0e=fan0e, fan1o=fan1o, fan1e=fan1e, fan0o=fan0o,
                t0e=fan0e, t1o=(fan1o + 4) // 5, t1e=(fan1e + 4) // 5, t0o=(fan0o + 4) // 5)


def cross(a, v):
    return np.array([a[1] * v[2] - a[2] * v[1], a[2] * v[0] - a[0] * v[2], a[0] * v[1] - a[1] * v[0]])


def emulate(stream, IN, OUT, xin, xrow, v, merged=False):
    """One 32-edge wave tile.  xin [32,96] = [edge_attr | x_src[:32] | x_dst[:32]], xrow [32,80], v [32,3] unit.
    `merged`: the inference kernel's layout (common.h::ConvShape::vmerged) -- block 1e's partly filled last tile rides in the free
    slots of block 0o's last tile."""
    S = shape(IN, OUT)
    r1e, r0o = (S["fan1e"] % 5, S["fan0o"] % 5) if OUT >= 3 else (0, 0)
    vm = bool(merged and OUT >= 3 and r1e > 0 and r0o > 0 and r1e + r0o <= 5)
    own1e = 5 * (S["t1e"] - 1) if vm else S["fan1e"]
    nt = 3 + S["t0e"] + S["t1o"] + (S["t1e"] if OUT >= 2 else 0) + (S["t0o"] if OUT >= 3 else 0) - int(vm)
    assert stream.size == (nt + 1) * TILE_W + nt * 32
    assert np.all(stream[nt * TILE_W:(nt + 1) * TILE_W] == 0)      # prefetch target after the last tile
    wts, bias = stream[:nt * TILE_W].reshape(nt, TILE_W), stream[(nt + 1) * TILE_W:].reshape(nt, 32)
    tiles = [(wts[k], bias[k]) for k in range(nt)]
    lanes = np.arange(64)
    j, hf = lanes & 31, lanes >> 5
    # first-Linear B operand: lane half hf holds columns 16hf..16hf+15 of each 32-wide source
    Bx = np.zeros((KSTEPS, 64))
    for s in range(KSTEPS):
        Bx[s] = xin[j, 32 * (s // 16) + 16 * hf + (s % 16)]
    T = 0
    h1 = np.zeros((KSTEPS, 64))
    for m in range(3):
        acc = gemm_tile(tiles[T], Bx); T += 1
        h1[16 * m:16 * m + 16] = np.maximum(acc, 0)

    def mid0e(e, i):
        return xrow[e, i] if i < 32 else float(xrow[e, COL_1O + 3 * (i - 32):COL_1O + 3 * (i - 32) + 3] @ v[e])

    def mid1o(e, i):
        if i < 32: return xrow[e, i] * v[e]
        if i < 32 + S["n1o"]: return xrow[e, COL_1O + 3 * (i - 32):COL_1O + 3 * (i - 32) + 3]
        if i < S["fan1o"]: return cross(xrow[e, COL_1E + 3 * (i - 32 - S["n1o"]):][:3], v[e])
        return np.zeros(3)

    def mid1e(e, i):
        if i < S["n1o"]: return cross(xrow[e, COL_1O + 3 * i:][:3], v[e])
        if i < S["n1o"] + S["n1e"]: return xrow[e, COL_1E + 3 * (i - S["n1o"]):][:3]
        if i < S["fan1e"]: return xrow[e, COL_0O + (i - S["n1o"] - S["n1e"])] * v[e]
        return np.zeros(3)

    def mid0o(e, i):
        if i < S["n1e"]: return float(xrow[e, COL_1E + 3 * i:][:3] @ v[e])
        if i < S["fan0o"]: return xrow[e, COL_0O + (i - S["n1e"])]
        return 0.0

    out = np.zeros((32, 80))
    o0e = np.zeros((16, 64))
    for i in range(S["t0e"]):
        acc = gemm_tile(tiles[T], h1); T += 1
        for lane in range(64):
            o0e[:, lane] += mid0e(lane & 31, i) * acc[:, lane]
    for lane in range(64):
        for reg in range(16):
            out[lane & 31, row_of(reg, lane >> 5)] = o0e[reg, lane]

    keeps = {}

    def vec_block(ntile, mid, col0, n_mids):
        keep = keeps[col0] = np.zeros((3, 3, 64))          # [o local][c][lane]; lane half hf owns outputs 3hf..3hf+2
        nonlocal T
        for t in range(ntile):
            acc = gemm_tile(tiles[T], h1); T += 1
            for lane in range(64):
                for q in range(5):
                    if 5 * t + q >= n_mids:
                        continue
                    m = mid(lane & 31, 5 * t + q)
                    for o in range(3):
                        keep[o, :, lane] += m * acc[3 * q + o, lane]

    def write_vec(col0):
        for lane in range(64):
            for o in range(3):
                oo = 3 * (lane >> 5) + o
                out[lane & 31, col0 + 3 * oo:col0 + 3 * oo + 3] = keeps[col0][o, :, lane]

    vec_block(S["t1o"], mid1o, COL_1O, S["fan1o"])
    write_vec(COL_1O)
    if OUT >= 2:
        vec_block(S["t1e"] - int(vm), mid1e, COL_1E, own1e)
    if OUT >= 3:
        k0 = np.zeros((3, 64))
        for t in range(S["t0o"]):
            acc = gemm_tile(tiles[T], h1); T += 1
            for lane in range(64):
                for q in range(5):
                    i = 5 * t + q
                    if i < S["fan0o"]:
                        k0[:, lane] += mid0o(lane & 31, i) * acc[3 * q:3 * q + 3, lane]
                    elif vm and i - S["fan0o"] < S["fan1e"] - own1e:      # block 1e's tail mids behind block 0o's own
                        m = mid1e(lane & 31, own1e + (i - S["fan0o"]))
                        for o in range(3):
                            keeps[COL_1E][o, :, lane] += m * acc[3 * q + o, lane]
        for lane in range(64):
            out[lane & 31, COL_0O + 3 * (lane >> 5):COL_0O + 3 * (lane >> 5) + 3] = k0[:, lane]
    if OUT >= 2:
        write_vec(COL_1E)
    assert T == len(tiles)
    return out


@pytest.mark.parametrize("merged", [False, True])
@pytest.mark.parametrize("IN,OUT", [(0, 1), (1, 2), (2, 3), (3, 3)])
def test_packed_stream_reproduces_fcblock_and_tensor_product(IN, OUT, merged):
    """`merged` = the layout the inference kernel reads (cbd_pack_conv_stream_infer); plain = the training kernels' and the bf16 kernel's"""
    from confidence_bootstrapping_amd.engine import pack_conv_stream, load_library
    lib = load_library()
    g = torch.Generator().manual_seed(10 * IN + OUT)
    in_irr, out_irr = sr.IRREP_SEQ[IN], sr.IRREP_SEQ[OUT]
    W = sr.faster_tp_weight_numel(in_irr, out_irr)
    w1, b1 = torch.randn(96, 96, generator=g) / 8, torch.randn(96, generator=g) / 4
    w2, b2 = torch.randn(W, 96, generator=g) / 8, torch.randn(W, generator=g) / 4
    stream = pack_conv_stream(IN, OUT, w1.numpy(), b1.numpy(), w2.numpy(), b2.numpy(), merged=merged)
    ntiles = {(0, 1): 3 + 32 + 7, (1, 2): 3 + 38 + 8 + 2, (2, 3): 3 + 38 + 9 + 3 + 2, (3, 3): 3 + 38 + 9 + 4 + 3}[(IN, OUT)]
    ntiles -= int(merged and OUT >= 3)           # 2 -> 3 and 3 -> 3: the tails of blocks 1e and 0o share a tile
    assert stream.size == (ntiles + 1) * TILE_W + ntiles * 32 == (lib.cbd_conv_stream_floats_infer if merged else lib.cbd_conv_stream_floats)(IN, OUT)
    E = 32
    in_dim, out_dim = sr.e3.Irreps(in_irr).dim, sr.e3.Irreps(out_irr).dim
    xin = torch.randn(E, 96, generator=g)
    xd = torch.randn(E, in_dim, generator=g)
    vec = torch.randn(E, 3, generator=g)
    # oracle: FCBlock then FasterTensorProduct (reference arithmetic), in float64
    hid = torch.relu(xin.double() @ w1.double().T + b1.double())
    tpw = hid @ w2.double().T + b2.double()
    ref = sr.faster_tensor_product(xd.double(), sh_l1(vec.double()), tpw, in_irr, out_irr).numpy()
    xrow = np.zeros((E, 80))
    xrow[:, :in_dim] = xd.double().numpy()
    v = torch.nn.functional.normalize(vec.double(), dim=-1).numpy()
    got = emulate(stream.astype(np.float64), IN, OUT, xin.double().numpy(), xrow, v, merged=merged)
    np.testing.assert_allclose(got[:, :out_dim], ref, rtol=2e-5, atol=2e-5)   # fp32-rounded folded factors in the stream
    assert np.all(got[:, out_dim:] == 0)
