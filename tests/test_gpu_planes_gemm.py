"""-m gpu: the pre-split plane GEMMs (csrc/planes_gemm.hip) through the C-ABI against float64 on the same inputs, for both
arithmetics -- three bf16 planes / six products ("bf16x6", gemm_precision 2) and two scaled f16 planes / three products ("f16x3",
gemm_precision 3): single segments, taps as row-shifted segments, ragged tiles, K not a multiple of 16, the three init modes and
ReLU, the transposed planes, and data chosen to break a scaled 16-bit split (huge dynamic range, one spike, tiny values).
Tolerance: f32-equivalent (1e-6 of the result's norm; an exact-f32 GEMM of these sizes sits at ~1e-7 to 4e-7)."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.gpu_util import dev, host, rel_l2

pytestmark = pytest.mark.gpu
ELEM = {3: torch.bfloat16, 2: torch.float16}


def tile_cols(N):
    return 160 if ((N + 159) // 160) * 160 - N < ((N + 127) // 128) * 128 - N else (256 if N % 256 == 0 else 128)


def planes_of(pkg, np_, x, lead=0, tail=0, t_tail=None):
    """(P, R, PT, Rt, scale) of a device matrix; NaN-filled buffers: pads must be written."""
    lib, abi = pkg.hipabi.load(), pkg.hipabi
    rows, cols = x.shape
    R = lead + rows + tail
    P = torch.full((lib.tdnnf_planes_bytes(np_, R, (cols + 15) // 16) // 2,), float("nan"), dtype=ELEM[np_], device="cuda")
    PT, Rt = None, 0
    if t_tail is not None:
        Rt = cols + t_tail
        PT = torch.full((lib.tdnnf_planes_bytes(np_, Rt, ((rows + 63) // 64) * 4) // 2,), float("nan"), dtype=ELEM[np_], device="cuda")
    scale = torch.zeros(4, device="cuda")
    ws = torch.zeros(lib.tdnnf_planes_split_workspace_bytes() // 4 + 4, device="cuda")
    abi.check(lib.tdnnf_planes_split(np_, abi.pmat(x), lead, R, abi.ptr(P), Rt, abi.ptr(PT) if PT is not None else None, abi.ptr(scale), abi.ptr(ws), abi.stream()))
    return P, R, PT, Rt, scale


def run(pkg, np_, M, N, Di, offs, init_mode=2, relu=0, seed=0, make=None):
    """C[m] = sum_i X[base + m + offs[i]] . W[:, i Di : (i + 1) Di]^T"""
    lib, abi = pkg.hipabi.load(), pkg.hipabi
    rng = np.random.default_rng(seed)
    K = len(offs)
    lo, hi = min(0, min(offs)), max(0, max(offs))
    rows_in = M + hi - lo
    X = rng.standard_normal((rows_in, Di)).astype(np.float32)
    W = (rng.standard_normal((N, K * Di)) / np.sqrt(K * Di)).astype(np.float32)
    if make is not None:
        X, W = make(rng, X, W)
    bias = rng.standard_normal(N).astype(np.float32)
    C0 = rng.standard_normal((M, N)).astype(np.float32)
    BN = tile_cols(N)
    Xd, Wd = dev(X), dev(W)
    lead, tail = 3, 256 + 5  # any lead; the tail covers the 256-row tile
    ap, RA, _, _, sa = planes_of(pkg, np_, Xd, lead, tail)
    bp, RB, _, _, sb = planes_of(pkg, np_, Wd, 0, ((N + BN - 1) // BN) * BN - N)
    Cd = dev(C0.copy())
    a_row = (C.c_longlong * K)(*[lead + (o - lo) for o in offs])
    a_col = (C.c_int * K)(*([0] * K))
    b_col = (C.c_int * K)(*[i * Di for i in range(K)])
    cols = (C.c_int * K)(*([Di] * K))
    bias_d = dev(bias)  # (kept in a variable: a temporary's memory goes back to the allocator before the launch)
    abi.check(lib.tdnnf_planes_gemm(np_, abi.ptr(ap), RA, abi.ptr(sa), abi.ptr(bp), RB, abi.ptr(sb), K, a_row, None, a_col, b_col, cols,
                                    abi.ptr(bias_d), init_mode, relu, abi.pmat(Cd), abi.stream()))
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


SHAPES = [
    (256, 160, 64, [0]),              # one tile, one segment
    (700, 160, 1536, [-3, 0]),        # the TDNN-F .linear shape: two taps, ragged rows
    (513, 1536, 160, [0, 1]),         # the .affine shape: 256-wide tiles, N = 6 tiles
    (400, 384, 96, [0, 2]),           # 128-wide tiles
    (300, 96, 48, [-1, 0, 1]),        # narrow output, three taps
    (260, 200, 40, [0]),              # K not a multiple of 16 (zero-padded K block), ragged columns
]


@pytest.mark.parametrize("np_", [3, 2], ids=["bf16x6", "f16x3"])
@pytest.mark.parametrize("M,N,Di,offs", SHAPES)
def test_planes_gemm_matches_float64(pkg, np_, M, N, Di, offs):
    got, ref = run(pkg, np_, M, N, Di, offs)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) < 1e-6, rel_l2(got, ref)


@pytest.mark.parametrize("np_", [3, 2], ids=["bf16x6", "f16x3"])
def test_planes_gemm_init_modes_and_relu(pkg, np_):
    for mode in (0, 1, 2):
        for relu in (0, 1):
            got, ref = run(pkg, np_, 300, 160, 320, [0, 2], init_mode=mode, relu=relu, seed=mode * 2 + relu)
            assert rel_l2(got, ref) < 1e-6, (mode, relu, rel_l2(got, ref))


def _wide(rng, X, W):  # log-normal magnitudes over ~10 decades
    return (X * np.exp(2.5 * rng.standard_normal(X.shape))).astype(np.float32), (W * np.exp(2.0 * rng.standard_normal(W.shape))).astype(np.float32)


def _spike(rng, X, W):  # one element carries almost all of the matrix's norm: the scale must still not overflow it
    X = (X * 1e-3).astype(np.float32)
    X[5, 7] = 3.0e4
    return X, W


def _tiny(rng, X, W):  # derivative-like magnitudes far below f16's range before scaling
    return (X * 1e-9).astype(np.float32), (W * 1e-4).astype(np.float32)


def _huge(rng, X, W):  # and far above it
    return (X * 1e12).astype(np.float32), (W * 1e6).astype(np.float32)


@pytest.mark.parametrize("make", [_wide, _spike, _tiny, _huge], ids=["ten-decades", "one-spike", "tiny", "huge"])
def test_scaled_f16_planes_on_hostile_data(pkg, make):
    """f16x3's scale comes from the Frobenius norm (s ||X||_F <= 65504), so no element can overflow whatever the data; the result stays
    f32-equivalent in norm.  The same data through exact f32 (torch) for comparison: f16x3 must not be more than 4x worse."""
    got, ref = run(pkg, 2, 700, 160, 1536, [-3, 0], seed=11, make=make)
    assert np.isfinite(got).all()
    e = rel_l2(got, ref)
    assert e < 1e-6, e


def test_planes_split_layout_and_transposed_planes(pkg):
    """x s = sum of the planes to 2^-22 relative (f16 pairs) / x = sum to 2^-24 (bf16 triples); lead / tail rows are zeros; the layout is
    [K block][plane][row][16] with the halves of a 32-byte row record swapped when bit 3 of the row is set; the transposed planes
    hold the same values with k = row index."""
    rng = np.random.default_rng(1)
    rows, cols = 137, 40
    X = (rng.standard_normal((rows, cols)) * np.exp(rng.uniform(-3, 3, (rows, cols)))).astype(np.float32)
    lead, tail, t_tail = 5, 9, 24

    def unswizzle(buf, nkb, np_, R):
        P = buf.reshape(nkb, np_, R, 2, 8)
        sw = (np.arange(R) >> 3) & 1
        return np.where(sw[None, None, :, None, None] == 1, P[:, :, :, ::-1, :], P).reshape(nkb, np_, R, 16)

    for np_, tol in ((3, 2.0 ** -23), (2, 2.0 ** -21)):
        P, R, PT, Rt, scale = planes_of(pkg, np_, dev(X), lead, tail, t_tail)
        s, inv, fro_rec = host(scale)[:3]
        if np_ == 2:
            fro = np.sqrt((X.astype(np.float64) ** 2).sum())
            assert s == 2.0 ** np.floor(np.log2(min(65504.0 / fro, 64.0 / (fro / np.sqrt(X.size))))) and inv == 1.0 / s and fro <= fro_rec <= fro * 1.00001
        else:
            s = 1.0
        nkb = (cols + 15) // 16
        Pn = unswizzle(host(P.float()), nkb, np_, R)
        assert not Pn[:, :, :lead].any() and not Pn[:, :, lead + rows:].any()
        rec = Pn[:, :, lead:lead + rows].astype(np.float64).sum(1).transpose(1, 0, 2).reshape(rows, nkb * 16)
        assert not rec[:, cols:].any()
        assert np.abs(rec[:, :cols] / s - X).max() <= tol * np.abs(X).max() and rel_l2(rec[:, :cols] / s, X) < tol
        nkbt = ((rows + 63) // 64) * 4
        Tn = unswizzle(host(PT.float()), nkbt, np_, Rt)
        assert not Tn[:, :, cols:].any()  # rows of the transposed planes beyond the matrix's columns
        rect = Tn[:, :, :cols].astype(np.float64).sum(1).transpose(1, 0, 2).reshape(cols, nkbt * 16)  # [column][row index]
        assert not rect[:, rows:].any()  # the K padding behind the last row
        assert np.array_equal(rect[:, :rows].T, rec[:, :cols])  # the same values, transposed


EPI_SHAPES = [
    # M, N, Di, offs, add rows [first, count), C column offset inside a wider buffer (0: 16-byte aligned rows; 1: not -> element-wise path)
    (700, 1536, 160, [0, 2], (0, 700), 0),      # .linear backward-data with the bypass addend over all rows; 128-row tiles (K <= 640)
    (700, 1536, 160, [0, 2], (130, 301), 0),    # the addend covers rows 130 .. 430 only: tile-interior and tile-crossing edges
    (513, 160, 1536, [-3, 0], (5, 500), 0),     # 160-wide tile: chunks of 64, 64, 32 columns
    (300, 200, 48, [0], (0, 300), 0),           # ragged right edge: the last chunk takes the element-wise path, the others the row path
    (300, 256, 48, [0], (7, 100), 1),           # output rows not 16-byte aligned: everything element-wise
    (1000, 256, 1536, [0], (0, 1000), 0),       # 256-row tiles, long K
]


@pytest.mark.parametrize("np_", [3, 2], ids=["bf16x6", "f16x3"])
@pytest.mark.parametrize("M,N,Di,offs,addr,coff", EPI_SHAPES)
def test_planes_gemm_epilogue_addend_and_column_statistics(pkg, np_, M, N, Di, offs, addr, coff):
    """tdnnf_planes_gemm_epilogue against float64: bias, ReLU, the row-windowed addend, and the column sums / sums of squares of the STORED
    output (one partial row per row tile), on the row-contiguous and the element-wise epilogue paths alike."""
    lib, abi = pkg.hipabi.load(), pkg.hipabi
    rng = np.random.default_rng(5)
    K = len(offs)
    lo, hi = min(0, min(offs)), max(0, max(offs))
    X = rng.standard_normal((M + hi - lo, Di)).astype(np.float32)
    W = (rng.standard_normal((N, K * Di)) / np.sqrt(K * Di)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((addr[1], N)).astype(np.float32)
    BN = tile_cols(N)
    lead, tail = 3, 256 + 5
    ap, RA, _, _, sa = planes_of(pkg, np_, dev(X), lead, tail)
    bp, RB, _, _, sb = planes_of(pkg, np_, dev(W), 0, ((N + BN - 1) // BN) * BN - N)
    ldc = ((N + coff + 3) // 4) * 4 + 4
    Cbuf = torch.full((M, ldc), float("nan"), device="cuda")
    Cd = Cbuf[:, coff:coff + N]
    stats = torch.full((2 * ((M + 127) // 128) * N,), float("nan"), device="cuda")
    nrows = C.c_int()
    a_row = (C.c_longlong * K)(*[lead + (o - lo) for o in offs])
    a_col = (C.c_int * K)(*([0] * K))
    b_col = (C.c_int * K)(*[i * Di for i in range(K)])
    cols = (C.c_int * K)(*([Di] * K))
    bias_d, add_d = dev(bias), dev(add)  # (kept in variables: a temporary's memory goes back to the allocator, and to the next upload, before the launch)
    abi.check(lib.tdnnf_planes_gemm_epilogue(np_, abi.ptr(ap), RA, abi.ptr(sa), abi.ptr(bp), RB, abi.ptr(sb), K, a_row, None, a_col, b_col, cols, abi.ptr(bias_d), 1, 1,
                                             abi.pmat(add_d), 0.66, addr[0], abi.ptr(stats), C.byref(nrows), abi.pmat(Cd), abi.stream()))
    ref = np.zeros((M, N))
    for i, o in enumerate(offs):
        ref += X[o - lo:o - lo + M].astype(np.float64) @ W[:, i * Di:(i + 1) * Di].astype(np.float64).T
    ref += bias
    ref[addr[0]:addr[0] + addr[1]] += 0.66 * add.astype(np.float64)
    ref = np.maximum(ref, 0)
    got = host(Cd)
    assert np.isfinite(got).all()
    assert rel_l2(got, ref) < 1e-6, rel_l2(got, ref)
    assert np.isnan(host(Cbuf[:, :coff])).all() and np.isnan(host(Cbuf[:, coff + N:])).all()  # nothing outside the output's columns
    t = nrows.value
    assert t in ((M + 255) // 256, (M + 127) // 128)
    st = host(stats)[:2 * t * N].reshape(2, t, N).astype(np.float64)
    assert np.isfinite(st).all()
    g64 = got.astype(np.float64)
    assert np.allclose(st[0].sum(0), g64.sum(0), rtol=1e-5, atol=1e-3)
    assert np.allclose(st[1].sum(0), (g64 * g64).sum(0), rtol=1e-5, atol=1e-3)
    th = 128 if t != (M + 255) // 256 else 256  # the launch's tile height (one partial row per row tile)
    for k in range(t):
        assert np.allclose(st[0, k], g64[k * th:(k + 1) * th].sum(0), rtol=1e-5, atol=1e-3)
