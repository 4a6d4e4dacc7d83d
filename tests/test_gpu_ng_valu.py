"""The natural-gradient statistics pass H = X~ W_t^T on the vector ALUs (csrc/ng_valu.hip) against float64 and against the MFMA rows
GEMM it replaces (tdnnf_ng_stats_pass: OnlineNaturalGradient::PreconditionDirections' first product; call sites
/root/reference/src/nnet3/nnet-tdnn-component.cc:598-599, nnet-simple-component.cc:3001-3002)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.gpu_util import F, Hip, dev, host, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(pkg):
    return Hip(pkg)


# rank, rows N, taps (row offsets), row_stride, Di, column of ones, tap coefficients
CASES = [
    (20, 700, (0,), 1, 161, False, False),       # one tap, a ragged width (161 = 10 K steps + 1 column), rows not a multiple of the tile
    (20, 1000, (0, 6), 1, 96, True, False),      # two taps of one matrix + the appended ones (a .affine's input side)
    (40, 515, (0, 3, 9), 1, 52, False, True),    # three taps with coefficients (TdnnDARTSV3Component), 52 = 1 K step of 32 + 20
    (80, 300, (2,), 3, 198, False, False),       # rank 80 (output side): K steps of 64, 198 = 3 * 64 + 6; every third row of X
    (80, 130, (0, 1), 1, 64, True, True),        # whole K steps only
    (20, 4, (0,), 1, 8, False, False),           # fewer rows than a wave
]


@pytest.mark.parametrize("R,N,offs,rs,Di,ones,use_eff", CASES)
def test_stats_pass_on_the_vector_alus(hip, pkg, R, N, offs, rs, Di, ones, use_eff):
    rng = np.random.default_rng(R + N)
    K = len(offs)
    rows_x = (N - 1) * rs + max(offs) + 1
    ldx = ((Di + 3) // 4) * 4 + 8
    xbuf = torch.full((rows_x, ldx), float("nan"), device="cuda")  # (what lies beside the matrix must never be read into a product)
    X = rng.standard_normal((rows_x, Di)).astype(F)
    xbuf[:, :Di] = dev(X)
    D = K * Di + (1 if ones else 0)
    W = (rng.standard_normal((R, D)) / np.sqrt(D)).astype(F)
    eff = rng.uniform(0.2, 1.0, K).astype(F) if use_eff else None
    if use_eff:
        eff[0] = 0.0  # a tap switched off
    wt = torch.zeros((K * Di + 64, R), device="cuda")  # W^T, k-major, 64 zero rows behind it
    wt[:K * Di] = dev(np.ascontiguousarray(W[:, :K * Di].T))
    ldw = ((D + 3) // 4) * 4
    wd = torch.zeros((R, ldw), device="cuda")
    wd[:, :D] = dev(W)
    bias = dev(np.ascontiguousarray(W[:, D - 1])) if ones else None
    effd = dev(eff) if use_eff else None
    cap = max(1024, (N + 127) // 128)
    ix = pkg.hipabi.indexes(rs, offs)
    ref = np.zeros((N, R))
    sq = 0.0
    for i, o in enumerate(offs):
        xi = X[o:o + (N - 1) * rs + 1:rs].astype(np.float64) * (eff[i] if use_eff else 1.0)
        ref += xi @ W[:, i * Di:(i + 1) * Di].astype(np.float64).T
        sq += (xi * xi).sum()
    if ones:
        ref += W[:, D - 1].astype(np.float64)
    out = {}
    for valu in (1, 0):
        H = torch.full((N, R), float("nan"), device="cuda")
        part = torch.full((cap,), float("nan"), dtype=torch.float64, device="cuda")
        hip.ng_stats_pass(C.byref(ix), xbuf[:, :Di], Di, hip.vec(effd) if use_eff else None, hip.vec(wt), hip.vec(wd), ldw,
                          hip.vec(bias) if ones else None, H, hip.vec(part), cap, valu, None, 0, hip.stream())
        out[valu] = (host(H), float(host(part).sum()))
        assert np.isfinite(out[valu][0]).all()
        assert rel_l2(out[valu][0], ref) < 2e-6, valu
        assert abs(out[valu][1] - sq) < 1e-5 * sq, valu
    assert rel_l2(out[1][0], out[0][0]) < 2e-6


def test_stats_pass_rejects_other_ranks(hip, pkg):
    ix = pkg.hipabi.indexes(1, (0,))
    X = torch.zeros((64, 32), device="cuda")
    wt = torch.zeros((32 + 64, 12), device="cuda")
    H = torch.zeros((64, 12), device="cuda")
    with pytest.raises(pkg.hipabi.HipAbiError, match="rank 20 / 40 / 80"):
        hip.ng_stats_pass(C.byref(ix), X, 32, None, hip.vec(wt), None, 0, None, H, None, 0, 1, None, 0, hip.stream())


@pytest.mark.parametrize("offs,ones", [((0, 256), False), ((128, 0, 384), True)])
def test_stats_pass_one_pass_over_the_taps(hip, pkg, offs, ones):
    """K taps that are row shifts of one matrix (a .linear's spliced input): P = X [W_0^T | W_1^T ..] in one pass, H[m] = sum_i P[m + o_i][block i],
    the taps' ||x||^2 from the pass's per-tile sums -- against float64 and the tap-by-tap MFMA pass."""
    rng = np.random.default_rng(len(offs))
    R, N, Di, K = 20, 32768 + 128, 1024, len(offs)
    rows_x = N + max(offs)
    ldx = Di + 32
    xbuf = torch.full((rows_x, ldx), float("nan"), device="cuda")
    X = rng.standard_normal((rows_x, Di)).astype(F)
    xbuf[:, :Di] = dev(X)
    D = K * Di + (1 if ones else 0)
    W = (rng.standard_normal((R, D)) / np.sqrt(D)).astype(F)
    ldw = ((D + 3) // 4) * 4
    wd = torch.zeros((R, ldw), device="cuda")
    wd[:, :D] = dev(W)
    bias = dev(np.ascontiguousarray(W[:, D - 1])) if ones else None
    cap = max(1024, (N + 127) // 128)
    ix = pkg.hipabi.indexes(1, offs)
    ref = np.zeros((N, R))
    sq = 0.0
    for i, o in enumerate(offs):
        xi = X[o:o + N].astype(np.float64)
        ref += xi @ W[:, i * Di:(i + 1) * Di].astype(np.float64).T
        sq += (xi * xi).sum()
    if ones:
        ref += W[:, D - 1].astype(np.float64)
    nb = hip.lib.tdnnf_ng_stats_pass_workspace_bytes(R, Di, K, N)
    ws = hip.ws(nb)
    out = {}
    for form in (2, 0):
        H = torch.full((N, R), float("nan"), device="cuda")
        part = torch.full((cap,), float("nan"), dtype=torch.float64, device="cuda")
        hip.ng_stats_pass(C.byref(ix), xbuf[:, :Di], Di, None, None, hip.vec(wd), ldw, hip.vec(bias) if ones else None, H, hip.vec(part), cap, form,
                          hip.vec(ws), nb, hip.stream())
        out[form] = (host(H), float(host(part).sum()))
        assert rel_l2(out[form][0], ref) < 2e-6, form
        assert abs(out[form][1] - sq) < 1e-6 * sq, form
    assert rel_l2(out[2][0], out[0][0]) < 2e-6
