"""-m gpu: the pre-split bf16-plane GEMM (csrc/planes_gemm.hip, gemm_precision 2's f32-equivalent arithmetic: three planes, six
products) through the C-ABI against float64 on the same inputs: single segments, taps as row-shifted segments, ragged tiles,
K not a multiple of 16, the three init modes and ReLU.  Tolerance: f32-equivalent (1e-6 of the result's norm; an exact-f32 GEMM of
these sizes sits at ~1e-7)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.gpu_util import dev, host, rel_l2

pytestmark = pytest.mark.gpu


def planes_of(pkg, x, lead, tail):
    lib, abi = pkg.hipabi.load(), pkg.hipabi
    nbytes = lib.tdnnf_planes_bytes(x.shape[0], x.shape[1], lead, tail)
    buf = torch.full((nbytes // 2,), float("nan"), dtype=torch.bfloat16, device="cuda")  # NaN-filled: pads must be written
    abi.check(lib.tdnnf_planes_split(abi.pmat(x), lead, tail, abi.ptr(buf), abi.stream()))
    return buf


def run(pkg, M, N, Di, offs, init_mode=2, relu=0, seed=0):
    """C[m] = sum_i X[base + m + offs[i]] . W[:, i Di : (i + 1) Di]^T"""
    lib, abi = pkg.hipabi.load(), pkg.hipabi
    rng = np.random.default_rng(seed)
    K = len(offs)
    lo, hi = min(0, min(offs)), max(0, max(offs))
    rows_in = M + hi - lo
    X = rng.standard_normal((rows_in, Di)).astype(np.float32)
    W = (rng.standard_normal((N, K * Di)) / np.sqrt(K * Di)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    BN = 160 if ((N + 159) // 160) * 160 - N < ((N + 127) // 128) * 128 - N else (256 if N % 256 == 0 else 128)
    Xd, Wd = dev(X), dev(W)
    lead, tail = 3, 256 + 5  # any lead; the tail covers the 256-row tile
    ap = planes_of(pkg, Xd, lead, tail)
    bp = planes_of(pkg, Wd, 0, ((N + BN - 1) // BN) * BN - N)
    Cd = dev(C0.copy())
    a_row = (C.c_longlong * K)(*[lead + (o - lo) for o in offs])
    a_col = (C.c_int * K)(*([0] * K))
    b_col = (C.c_int * K)(*[i * Di for i in range(K)])
    cols = (C.c_int * K)(*([Di] * K))
    abi.check(lib.tdnnf_planes_gemm(abi.ptr(ap), lead + rows_in + tail, abi.ptr(bp), ((N + BN - 1) // BN) * BN, K, a_row, a_col, b_col, cols,
                                    abi.ptr(dev(bias)), init_mode, relu, abi.pmat(Cd), abi.stream()))
    ref = np.zeros((M, N))
    for i, o in enumerate(offs):
        ref += X[o - lo:o - lo + M].astype(np.float64) @ W[:, i * Di:(i + 1) * Di].astype(np.float64).T
    if init_mode == 0:
        ref += C0
    elif init_mode == 1:
        ref += bias
    if relu:
        ref = np.maximum(ref, 0)
    return host(Cd), ref


@pytest.mark.parametrize("M,N,Di,offs", [
    (256, 160, 64, [0]),              # one tile, one segment
    (700, 160, 1536, [-3, 0]),        # the TDNN-F .linear shape: two taps, ragged rows
    (513, 1536, 160, [0, 1]),         # the .affine shape: 256-wide tiles, N = 6 tiles
    (400, 384, 96, [0, 2]),           # 128-wide tiles
    (300, 96, 48, [-1, 0, 1]),        # narrow output, three taps
    (260, 200, 40, [0]),              # K not a multiple of 16 (zero-padded K block), ragged columns
])
def test_planes_gemm_matches_float64(pkg, M, N, Di, offs):
    got, ref = run(pkg, M, N, Di, offs)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) < 1e-6, rel_l2(got, ref)


def test_planes_gemm_init_modes_and_relu(pkg):
    for mode in (0, 1, 2):
        for relu in (0, 1):
            got, ref = run(pkg, 300, 160, 320, [0, 2], init_mode=mode, relu=relu, seed=mode * 2 + relu)
            assert rel_l2(got, ref) < 1e-6, (mode, relu, rel_l2(got, ref))


def test_planes_split_is_three_bf16_planes(pkg):
    """x = p0 + p1 + p2 to 2^-24 relative, lead / tail rows are zeros, the layout is [K block][plane][row][16] with the halves of a
    32-byte row record swapped when bit 3 of the row is set."""
    rng = np.random.default_rng(1)
    X = (rng.standard_normal((37, 40)) * np.exp(rng.uniform(-8, 8, (37, 40)))).astype(np.float32)
    lead, tail = 5, 9
    buf = host(planes_of(pkg, dev(X), lead, tail).float())
    R, nkb = lead + 37 + tail, 3
    P = buf.reshape(nkb, 3, R, 2, 8)
    rows = np.arange(R)
    sw = (rows >> 3) & 1
    P = np.where(sw[None, None, :, None, None] == 1, P[:, :, :, ::-1, :], P).reshape(nkb, 3, R, 16)
    assert not P[:, :, :lead].any() and not P[:, :, lead + 37:].any()
    rec = P[:, :, lead:lead + 37].sum(1).transpose(1, 0, 2).reshape(37, nkb * 16)
    assert not rec[:, 40:].any()
    assert np.abs(rec[:, :40] - X).max() <= 2.0 ** -23 * np.abs(X).max() and rel_l2(rec[:, :40], X) < 2.0 ** -24
