"""-m gpu parity tests proper: every C-ABI entry of include/tdnnf_hip.h against the CPU oracle on
identical seeded inputs (sizes the oracle finishes in seconds).  Tolerances: the reference path is
fp32 (BaseFloat); our kernels are exact-f32 MFMA, so differences are summation-order only.
BASELINE.json's bars are objf 1e-4 relative and param-grad L2 1e-3; we hold 2e-5 here."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

from tests.gpu_util import F, Hip, dev, host, padded, rel_l2

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def hip(pkg):
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return Hip(pkg)


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(F)


TDNN_CASES = [
    # name, offsets, num_t_out, B, Di, Do, t_step_out
    ("config1-linear", [-1, 0, 1], 150, 128, 40, 160, 1),      # BASELINE.json configs[0]
    ("config1-affine-k3", [-1, 0, 1], 20, 16, 160, 1536, 1),
    ("tdnnf-linear", [-1, 0], 24, 32, 1536, 160, 1),
    ("tdnnf-affine-stride3", [0, 3], 9, 16, 160, 1536, 3),
    ("tdnnf-linear-stride3", [-3, 0], 9, 16, 1536, 160, 3),
    ("supernet-k7", [-6, -5, -4, -3, -2, -1, 0], 12, 8, 96, 160, 1),
    ("single-tap", [0], 7, 5, 220, 1536, 1),
    ("ragged", [-2, 0, 1], 5, 3, 36, 50, 1),                   # dims not multiples of the tile
    ("output-layer", [0], 6, 7, 256, 6034, 1),
]


@pytest.mark.parametrize("case", TDNN_CASES, ids=[c[0] for c in TDNN_CASES])
def test_tdnn_propagate_backprop_update(hip, ora, pkg, case):
    name, offs, nt, B, Di, Do, step = case
    L = ora.lib()
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B, t_step_out=step)
    K = len(offs)
    x = _rand(rng, rows_in, Di)
    W = (_rand(rng, Do, K * Di) / np.sqrt(K * Di)).astype(F)
    b = _rand(rng, Do)
    c = (rng.random(K) + 0.25).astype(F)
    if K >= 3:
        c[1] = 0.0  # a zero coefficient must skip the tap (uniform-sample mode)
    dy = _rand(rng, N, Do)
    ix = pkg.hipabi.indexes(rho, ro)
    # ---- forward (init_mode 1: bias)
    y_ref = np.zeros((N, Do), F)
    L.oracle_tdnn_propagate(ora.omat(x), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), ora.fptr(b), ora.fptr(c), 1,
                            ora.omat(y_ref))
    xd, _ = padded(x)
    yd, ybuf = padded(np.zeros((N, Do), F))
    Wd, bd, cd = dev(W), dev(b), dev(c)
    hip.tdnn_propagate(C.byref(ix), xd, hip.vec(Wd), K * Di, Do, Di, hip.vec(bd), hip.vec(cd), 1, yd, hip.stream())
    assert rel_l2(host(yd), y_ref) < TOL
    assert (host(ybuf)[:, Do:] == 7.0).all(), "wrote outside the view"
    # init_mode 0 adds into out, init_mode 2 overwrites with no bias
    hip.tdnn_propagate(C.byref(ix), xd, hip.vec(Wd), K * Di, Do, Di, None, hip.vec(cd), 0, yd, hip.stream())
    assert rel_l2(host(yd), 2 * y_ref - b) < TOL
    hip.tdnn_propagate(C.byref(ix), xd, hip.vec(Wd), K * Di, Do, Di, None, None, 2, yd, hip.stream())
    y1 = np.zeros((N, Do), F)
    L.oracle_tdnn_propagate(ora.omat(x), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), None, None, 2, ora.omat(y1))
    assert rel_l2(host(yd), y1) < TOL
    # ---- backward data (adds into in_deriv)
    dx0 = _rand(rng, rows_in, Di)
    dx_ref = dx0.copy()
    L.oracle_tdnn_backprop_data(ora.omat(dy), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), ora.fptr(c), ora.omat(dx_ref))
    dyd, _ = padded(dy)
    dxd, dxbuf = padded(dx0)
    hip.tdnn_backprop_data(C.byref(ix), dyd, hip.vec(Wd), K * Di, Do, Di, hip.vec(cd), dxd, hip.stream())
    assert rel_l2(host(dxd) - dx0, dx_ref - dx0) < TOL
    assert (host(dxbuf)[:, Di:] == 7.0).all()
    # ---- raw parameter gradient (UpdateSimple) accumulated on top of existing content
    W0, b0 = _rand(rng, Do, K * Di), _rand(rng, Do)
    W_ref, b_ref = W0.copy(), b0.copy()
    L.oracle_tdnn_update_simple(ora.omat(x), ora.omat(dy), Do, Di, K, rho, ora.iptr(ro), ora.fptr(c), 0.5, ora.fptr(W_ref),
                                K * Di, ora.fptr(b_ref))
    Wacc, bacc = dev(W0), dev(b0)
    nbytes = hip.tdnn_update_workspace_bytes(Do, Di, K, N)
    ws = hip.ws(nbytes)
    hip.tdnn_update_simple(C.byref(ix), xd, dyd, Do, Di, hip.vec(cd), 0.5, hip.vec(Wacc), K * Di, hip.vec(bacc), hip.vec(ws),
                           nbytes, hip.stream())
    assert rel_l2(host(Wacc) - W0, W_ref - W0) < TOL
    assert rel_l2(host(bacc) - b0, b_ref - b0) < TOL


def test_tdnn_argument_errors(hip, pkg):
    """KALDI_ASSERTs of GetInputPart (nnet-tdnn-component.cc:811-813) become TDNNF_EINVAL, nothing is launched."""
    x = torch.zeros(10, 8, device="cuda")
    y = torch.zeros(10, 4, device="cuda")
    W = torch.zeros(4, 16, device="cuda")
    ix = pkg.hipabi.indexes(1, [0, 1])  # needs 11 input rows
    rc = hip.lib.tdnnf_tdnn_propagate(C.byref(ix), pkg.hipabi.pmat(x), pkg.hipabi.ptr(W), 16, 4, 8, None, None, 2,
                                      pkg.hipabi.pmat(y), None)
    assert rc == 1 and b"too few rows" in hip.lib.tdnnf_last_error()
    rc = hip.lib.tdnnf_tdnn_propagate(C.byref(ix), pkg.hipabi.pmat(x), pkg.hipabi.ptr(W), 16, 4, 8, None, None, 1,
                                      pkg.hipabi.pmat(y), None)
    assert rc == 1


@pytest.mark.parametrize("flags", [0, 1, 2, 4, 1 | 4, 8 | 16, 1 | 8 | 16])
def test_darts_coef_and_alpha_update(hip, ora, pkg, flags):
    L = ora.lib()
    rng = np.random.default_rng(flags)
    offs = [-2, -1, 0]
    K, B, Di, Do, nt = 3, 4, 24, 20, 9
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B)
    share = L.oracle_tdnn_share_index(ora.iptr(np.asarray(offs, np.int32)), K)
    la, u = _rand(rng, K), rng.random(K).astype(F)
    su = np.asarray([0.4], F)
    coef, eff = np.zeros(K, F), np.zeros(K, F)
    L.oracle_tdnn_darts_coef(ora.fptr(la), K, flags, 0.7, ora.fptr(u), float(su[0]), ora.fptr(coef))
    L.oracle_tdnn_darts_effective_coef(ora.fptr(coef), K, flags, share, ora.fptr(eff))
    lad, ud, sud = dev(la), dev(u), dev(su)
    cm, ce = torch.zeros(K, device="cuda"), torch.zeros(K, device="cuda")
    hip.tdnn_darts_coef(hip.vec(lad), K, flags, 0.7, hip.vec(ud), hip.vec(sud), share, hip.vec(cm), hip.vec(ce), hip.stream())
    np.testing.assert_allclose(host(cm), coef, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(host(ce), eff, rtol=1e-5, atol=1e-7)
    # alpha update from the unscaled tap gradients
    x, W, dy = _rand(rng, rows_in, Di), _rand(rng, Do, K * Di), _rand(rng, N, Do)
    s = np.zeros(K)
    L.oracle_tdnn_darts_tap_dots(ora.omat(x), ora.omat(dy), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), ora.dptr(s))
    acc0 = _rand(rng, K)
    acc = acc0.copy()
    L.oracle_tdnn_darts_alpha_update(ora.dptr(s), ora.fptr(coef), K, flags, share, 0.7, 0.01, ora.fptr(acc))
    ix = pkg.hipabi.indexes(rho, ro)
    G = torch.zeros(Do, K * Di, device="cuda")
    nbytes = hip.tdnn_update_workspace_bytes(Do, Di, K, N)
    ws = hip.ws(nbytes)
    hip.tdnn_update_simple(C.byref(ix), dev(x), dev(dy), Do, Di, None, 1.0, hip.vec(G), K * Di, None, hip.vec(ws), nbytes, hip.stream())
    accd = dev(acc0)
    dots = torch.zeros(K * 65, dtype=torch.float64, device="cuda")  # TDNNF_TAP_DOTS_DOUBLES(K)
    hip.tdnn_darts_alpha_update(hip.vec(G), K * Di, hip.vec(dev(W)), K * Di, Do, Di, K, hip.vec(cm), flags, share, 0.7, 0.01,
                                hip.vec(accd), hip.vec(dots), hip.stream())
    np.testing.assert_allclose(host(dots)[:K], s, rtol=1e-4)
    np.testing.assert_allclose(host(accd), acc, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("N,D", [(19200, 1536), (777, 256), (53, 37)])
def test_batchnorm(hip, ora, N, D):
    L = ora.lib()
    rng = np.random.default_rng(N)
    x = (_rand(rng, N, D) * 1.7 + 0.3).astype(F)
    dz = _rand(rng, N, D)
    z_ref, memo_ref = np.zeros_like(x), np.zeros((5, D), F)
    L.oracle_batchnorm_propagate(ora.omat(x), 1e-3, 1.0, ora.omat(z_ref), ora.fptr(memo_ref))
    xd, _ = padded(x)
    zd, _ = padded(np.zeros_like(x))
    memo = torch.zeros(5, D, device="cuda")
    nbytes = hip.colreduce_workspace_bytes(N, D)
    ws = hip.ws(nbytes)
    hip.batchnorm_propagate(xd, 1e-3, 1.0, zd, hip.vec(memo), hip.vec(ws), nbytes, hip.stream())
    np.testing.assert_allclose(host(memo)[:3], memo_ref[:3], rtol=2e-5, atol=1e-6)
    assert rel_l2(host(zd), z_ref) < TOL
    dx_ref = np.zeros_like(x)
    L.oracle_batchnorm_backprop(ora.omat(z_ref), ora.omat(dz), 1.0, ora.fptr(memo_ref), ora.omat(dx_ref))
    dxd, _ = padded(np.zeros_like(x))
    hip.batchnorm_backprop(zd, dev(dz), 1.0, hip.vec(memo), dxd, hip.vec(ws), nbytes, hip.stream())
    assert rel_l2(host(dxd), dx_ref) < 5e-5
    # in-place forward (the component is in-place capable, nnet-normalize-component.h:187-188)
    hip.batchnorm_propagate(xd, 1e-3, 1.0, xd, hip.vec(memo), hip.vec(ws), nbytes, hip.stream())
    assert rel_l2(host(xd), z_ref) < TOL
    # stats / derived / test mode
    stats = torch.zeros(1 + 2 * D, dtype=torch.float64, device="cuda")
    hip.batchnorm_store_stats(hip.vec(memo), D, N, hip.vec(stats), hip.stream())
    hip.batchnorm_store_stats(hip.vec(memo), D, N, hip.vec(stats), hip.stream())
    cnt = C.c_double(0)
    ssum, ssq = np.zeros(D), np.zeros(D)
    for _ in range(2):
        L.oracle_batchnorm_store_stats(ora.fptr(memo_ref), D, N, C.byref(cnt), ora.dptr(ssum), ora.dptr(ssq))
    st = host(stats)
    assert st[0] == cnt.value
    np.testing.assert_allclose(st[1:1 + D], ssum, rtol=1e-5, atol=1e-3)
    sc_ref, of_ref = np.zeros(D, F), np.zeros(D, F)
    L.oracle_batchnorm_compute_derived(cnt.value, ora.dptr(ssum), ora.dptr(ssq), D, 1e-3, 1.0, ora.fptr(sc_ref), ora.fptr(of_ref))
    sc, of = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    hip.batchnorm_compute_derived(hip.vec(stats), D, 1e-3, 1.0, hip.vec(sc), hip.vec(of), hip.stream())
    np.testing.assert_allclose(host(sc), sc_ref, rtol=1e-4)
    np.testing.assert_allclose(host(of), of_ref, rtol=1e-4, atol=1e-6)
    out = torch.zeros(N, D, device="cuda")
    hip.batchnorm_test_propagate(dev(x), hip.vec(sc), hip.vec(of), out, hip.stream())
    ref = np.zeros_like(x)
    L.oracle_batchnorm_test_propagate(ora.omat(x), ora.fptr(sc_ref), ora.fptr(of_ref), ora.omat(ref))
    assert rel_l2(host(out), ref) < 1e-4
    hip.batchnorm_test_backprop(dev(dz), hip.vec(sc), out, hip.stream())
    assert rel_l2(host(out), dz * sc_ref) < 1e-4


@pytest.mark.parametrize("gumbel", [False, True])
def test_softmax_flops(hip, ora, gumbel):
    L = ora.lib()
    rng = np.random.default_rng(9)
    N, Cc, eta, tau = 1000, 8, 0.3, (0.6 if gumbel else 1.0)
    x = np.repeat(_rand(rng, 1, Cc), N, 0)
    x[3] += 0.5  # rows need not be identical for the kernel to be right
    u = rng.random(Cc).astype(F) if gumbel else None
    p_ref = np.zeros_like(x)
    L.oracle_softmax_flops_propagate(ora.omat(x), ora.fptr(u), tau, ora.omat(p_ref))
    p = torch.zeros(N, Cc, device="cuda")
    hip.softmax_flops_propagate(dev(x), hip.vec(dev(u)) if gumbel else None, tau, p, hip.stream())
    np.testing.assert_allclose(host(p), p_ref, rtol=2e-5)
    flops = -np.asarray([25, 50, 80, 100, 120, 160, 200, 240], F)
    dp = _rand(rng, N, Cc)
    dp_ref, dx_ref = dp.copy(), np.zeros_like(x)
    L.oracle_softmax_flops_backprop(ora.omat(p_ref), ora.omat(dp_ref), eta, ora.fptr(flops), Cc, tau, ora.omat(dx_ref))
    dpd, dxd = dev(dp), torch.zeros(N, Cc, device="cuda")
    hip.softmax_flops_backprop(p, dpd, eta, hip.vec(dev(flops)), Cc, tau, dxd, hip.stream())
    np.testing.assert_allclose(host(dpd), dp_ref, rtol=1e-6)  # in-place mutation reproduced
    np.testing.assert_allclose(host(dxd), dx_ref, rtol=2e-4, atol=1e-7)


@pytest.mark.parametrize("N,Cc", [(700, 5), (257, 13), (64, 3)])
def test_plain_gumbel_softmax_without_flops_vector(hip, ora, N, Cc):
    """GumbelSoftmaxComponent (nnet-simple-component.cc:9774-9831): any width, no FLOPs penalty (flops_dev NULL),
    out_deriv is left untouched, in_deriv = DiffSoftmax / temp."""
    L = ora.lib()
    rng = np.random.default_rng(90 + Cc)
    tau = 0.45
    x = _rand(rng, N, Cc)
    u = rng.random(Cc).astype(F)
    p_ref = np.zeros_like(x)
    L.oracle_softmax_flops_propagate(ora.omat(x), ora.fptr(u), tau, ora.omat(p_ref))
    p = torch.zeros(N, Cc, device="cuda")
    hip.softmax_flops_propagate(dev(x), hip.vec(dev(u)), tau, p, hip.stream())
    np.testing.assert_allclose(host(p), p_ref, rtol=2e-5, atol=1e-20)
    np.testing.assert_allclose(host(p).sum(1), 1.0, rtol=1e-5)
    dp = _rand(rng, N, Cc)
    dp_ref, dx_ref = dp.copy(), np.zeros_like(x)
    L.oracle_softmax_flops_backprop(ora.omat(p_ref), ora.omat(dp_ref), 0.3, None, 0, tau, ora.omat(dx_ref))
    dpd, dxd = dev(dp), torch.zeros(N, Cc, device="cuda")
    hip.softmax_flops_backprop(p, dpd, 0.3, None, 0, tau, dxd, hip.stream())
    assert (host(dpd) == dp).all()  # no penalty: the output derivative is not mutated
    # (e_c - <p, e> cancels for some elements: float rounding of the row dot product shows there, so the bound is absolute)
    np.testing.assert_allclose(host(dxd), dx_ref, rtol=2e-4, atol=2e-6)
    assert rel_l2(host(dxd), dx_ref) < 1e-5
    # independent check: float64 Jacobian (diag(p) - p p^T) / temp
    pd = p_ref.astype(np.float64)
    want = pd * (dp - (pd * dp).sum(1, keepdims=True)) / tau
    np.testing.assert_allclose(host(dxd), want, rtol=2e-4, atol=2e-6)
    assert rel_l2(host(dxd), want) < 1e-5


def test_small_ops(hip, ora):
    L = ora.lib()
    rng = np.random.default_rng(10)
    N = 300
    for u in [0.0, 0.124, 0.125, 0.5, 0.999]:
        out = torch.zeros(N, 8, device="cuda")
        hip.onehot_propagate(hip.vec(dev(np.asarray([u], F))), out, hip.stream())
        ref = np.zeros((N, 8), F)
        L.oracle_onehot_propagate(u, ora.omat(ref))
        assert (host(out) == ref).all()
    a = _rand(rng, N, 1)
    o0 = _rand(rng, N, 40)
    ref = o0.copy()
    L.oracle_copyn_propagate(ora.omat(a), 1.5, ora.omat(ref))
    od = dev(o0)
    hip.copyn_propagate(dev(a), 1.5, od, hip.stream())
    np.testing.assert_allclose(host(od), ref, rtol=1e-6, atol=1e-6)  # fma contraction on the GPU
    do = _rand(rng, N, 40)
    da0 = _rand(rng, N, 1)
    ref = da0.copy()
    L.oracle_copyn_backprop(ora.omat(do), 1.5, ora.omat(ref))
    dad = dev(da0)
    hip.copyn_backprop(dev(do), 1.5, dad, hip.stream())
    np.testing.assert_allclose(host(dad), ref, rtol=1e-5, atol=1e-6)
    x = _rand(rng, N, 480)
    ref = np.zeros((N, 240), F)
    L.oracle_elementwise_product_propagate(ora.omat(x), 240, ora.omat(ref))
    y = torch.zeros(N, 240, device="cuda")
    hip.elementwise_product_propagate(dev(x), 240, y, hip.stream())
    np.testing.assert_allclose(host(y), ref, rtol=1e-6)
    dyy = _rand(rng, N, 240)
    ref = np.zeros_like(x)
    L.oracle_elementwise_product_backprop(ora.omat(x), ora.omat(dyy), 240, ora.omat(ref))
    dxx = torch.zeros(N, 480, device="cuda")
    hip.elementwise_product_backprop(dev(x), dev(dyy), 240, dxx, hip.stream())
    np.testing.assert_allclose(host(dxx), ref, rtol=1e-6)
    alpha = _rand(rng, 8)
    out = torch.zeros(N, 8, device="cuda")
    hip.constant_function_propagate(hip.vec(dev(alpha)), out, hip.stream())
    assert (host(out) == alpha).all()
    d = _rand(rng, N, 8)
    acc0 = _rand(rng, 8)
    ref = acc0.copy()
    L.oracle_constant_function_backprop(ora.omat(d), 0.1, ora.fptr(ref))
    accd = dev(acc0)
    nb = hip.colreduce_workspace_bytes(N, 8)
    ws = hip.ws(nb)
    hip.constant_function_backprop(dev(d), 0.1, hip.vec(accd), hip.vec(ws), nb, hip.stream())
    np.testing.assert_allclose(host(accd), ref, rtol=1e-4, atol=1e-5)
    fl = -np.asarray([25, 50, 80, 100, 120, 160, 200, 240], F)
    ref = np.zeros((N, 8), F)
    L.oracle_flops_constraint_backprop(ora.fptr(fl), 0.2, N, 8, ora.omat(ref))
    out = torch.zeros(N, 8, device="cuda")
    hip.flops_constraint_backprop(hip.vec(dev(fl)), 0.2, N, 8, out, hip.stream())
    np.testing.assert_allclose(host(out), ref, rtol=1e-6)
    mask = (1 + 0.2 * _rand(rng, 4, 64)).astype(F)
    x = _rand(rng, 40, 64)
    ref = np.zeros_like(x)
    L.oracle_general_dropout_propagate(ora.omat(x), ora.fptr(mask), 4, ora.omat(ref))
    out = torch.zeros(40, 64, device="cuda")
    hip.general_dropout(dev(x), hip.vec(dev(mask)), 4, out, hip.stream())
    np.testing.assert_allclose(host(out), ref, rtol=1e-6)


def test_relu_sum_logsoftmax(hip, ora):
    L = ora.lib()
    rng = np.random.default_rng(11)
    N, D = 999, 1536
    x, dy = _rand(rng, N, D), _rand(rng, N, D)
    xd, _ = padded(x)
    r = torch.zeros(N, D, device="cuda")
    hip.relu_propagate(xd, r, hip.stream())
    assert (host(r) == np.maximum(x, 0)).all()
    dxd = torch.zeros(N, D, device="cuda")
    hip.relu_backprop(r, dev(dy), dxd, hip.stream())
    assert (host(dxd) == (x > 0) * dy).all()
    stats = torch.zeros(1 + 2 * D, dtype=torch.float64, device="cuda")
    nb = hip.colreduce_workspace_bytes(N, D)
    ws = hip.ws(nb)
    hip.relu_store_stats(r, hip.vec(stats), hip.vec(ws), nb, hip.stream())
    vs, ds = np.zeros(D), np.zeros(D)
    cnt = C.c_double(0)
    rx = np.maximum(x, 0)  # keep alive: omat() only borrows the buffer
    L.oracle_relu_store_stats(ora.omat(rx), ora.dptr(vs), ora.dptr(ds), C.byref(cnt))
    st = host(stats)
    assert st[0] == N and (st[1 + D:] == ds).all()
    np.testing.assert_allclose(st[1:1 + D], vs, rtol=1e-5)
    # self-repair with stats that trip both thresholds
    st2 = np.concatenate([[100.0], np.zeros(D), rng.choice([0.0, 50.0, 100.0], D)])
    ref = dy.copy()
    L.oracle_relu_repair(ora.dptr(np.ascontiguousarray(st2[1 + D:])), 100.0, D, 1e-5, 0.05, 0.95, ora.omat(ref))
    dd = dev(dy)
    hip.relu_repair(hip.vec(dev(st2)), D, 1e-5, 0.05, 0.95, dd, hip.stream())
    np.testing.assert_allclose(host(dd), ref, rtol=1e-6, atol=1e-9)
    # Sum(Scale(0.66, a), b), in place on b
    a, b = _rand(rng, N, D), _rand(rng, N, D)
    bd = dev(b)
    hip.sum_scaled(dev(a), 0.66, bd, 1.0, bd, hip.stream())
    np.testing.assert_allclose(host(bd), F(0.66) * a + b, rtol=1e-6, atol=1e-6)
    hip.add_scaled(dev(a), 2.0, bd, hip.stream())
    np.testing.assert_allclose(host(bd), F(0.66) * a + b + 2 * a, rtol=1e-5, atol=1e-6)
    # log-softmax over 6034 pdfs
    z = _rand(rng, 64, 6034) * 3
    ref = np.zeros_like(z)
    L.oracle_log_softmax_propagate(ora.omat(z), ora.omat(ref))
    zd, _ = padded(z)
    out = torch.zeros(64, 6034, device="cuda")
    hip.log_softmax_propagate(zd, out, hip.stream())
    np.testing.assert_allclose(host(out), ref, rtol=1e-5, atol=1e-5)
    e = _rand(rng, 64, 6034)
    dref = np.zeros_like(z)
    L.oracle_log_softmax_backprop(ora.omat(ref), ora.omat(e), ora.omat(dref))
    dd = torch.zeros(64, 6034, device="cuda")
    hip.log_softmax_backprop(out, dev(e), dd, hip.stream())
    np.testing.assert_allclose(host(dd), dref, rtol=1e-4, atol=1e-5)


def test_affine(hip, ora):
    L = ora.lib()
    rng = np.random.default_rng(12)
    N, Di, Do = 700, 256, 1536
    x, W, b, dy = _rand(rng, N, Di), _rand(rng, Do, Di) / 16, _rand(rng, Do), _rand(rng, N, Do)
    W = W.astype(F)
    ref = np.zeros((N, Do), F)
    L.oracle_affine_propagate(ora.omat(x), ora.fptr(W), Di, ora.fptr(b), Do, ora.omat(ref))
    y = torch.zeros(N, Do, device="cuda")
    hip.affine_propagate(dev(x), hip.vec(dev(W)), Di, hip.vec(dev(b)), Do, y, hip.stream())
    assert rel_l2(host(y), ref) < TOL
    dref = np.zeros((N, Di), F)
    L.oracle_affine_backprop(ora.omat(dy), ora.fptr(W), Di, Di, ora.omat(dref))
    dx = torch.full((N, Di), 3.0, device="cuda")  # must be overwritten
    hip.affine_backprop(dev(dy), hip.vec(dev(W)), Di, Di, dx, hip.stream())
    assert rel_l2(host(dx), dref) < TOL
    Wr, br = np.zeros_like(W), np.zeros_like(b)
    L.oracle_affine_update_simple(ora.omat(x), ora.omat(dy), 1.0, ora.fptr(Wr), Di, ora.fptr(br))
    Wa, ba = torch.zeros(Do, Di, device="cuda"), torch.zeros(Do, device="cuda")
    nb = hip.tdnn_update_workspace_bytes(Do, Di, 1, N)
    ws = hip.ws(nb)
    hip.affine_update_simple(dev(x), dev(dy), 1.0, hip.vec(Wa), Di, hip.vec(ba), hip.vec(ws), nb, hip.stream())
    assert rel_l2(host(Wa), Wr) < TOL and rel_l2(host(ba), br) < TOL


CHAIN_CASES = [(50, 40, 3, 8, 0.1, 0.0), (300, 200, 6, 30, 0.1, 5e-5), (2000, 600, 4, 25, 0.1, 0.0), (64, 6034, 2, 10, 0.0, 0.0)]


# both forms of the denominator recursion: one persistent workgroup per sequence (state vectors in LDS), one launch per frame
# over all sequences on sequence-minor arrays (the form large graphs get); B = 70 and 130 put sequences in a second chunk of lanes
@pytest.mark.parametrize("mode", [1, 2], ids=["persistent", "wide"])
@pytest.mark.parametrize("H,P,B,T,leaky,l2", CHAIN_CASES + [(500, 300, 70, 6, 0.1, 0.0), (120, 90, 130, 4, 0.05, 1e-5)])
def test_chain_objf_and_deriv(hip, ora, pkg, H, P, B, T, leaky, l2, mode):
    pkg.hipabi.check(pkg.hipabi.load().tdnnf_chain_set_denominator_mode(mode))
    try:
        _chain_case(hip, ora, pkg, H, P, B, T, leaky, l2)
    finally:
        pkg.hipabi.load().tdnnf_chain_set_denominator_mode(0)


def _chain_case(hip, ora, pkg, H, P, B, T, leaky, l2):
    L = ora.lib()
    g = pkg.synth.make_den_graph(H, P, mean_out_degree=6.0, seed=H)
    sup = pkg.synth.make_supervision(B, T, P, seed=T, weight=1.0)
    rng = np.random.default_rng(H + T)
    y = (_rand(rng, T * B, P) * 1.5).astype(F)
    xo = _rand(rng, T * B, P)
    gs, ss = ora.den_graph_struct(g), ora.supervision_struct(sup)
    objf, l2t, w = C.c_double(), C.c_double(), C.c_double()
    d_ref, xd_ref = np.zeros_like(y), np.zeros_like(y)
    ok = L.oracle_chain_objf_and_deriv(C.byref(gs), C.byref(ss), ora.omat(y), leaky, l2, 0.1, C.byref(objf), C.byref(l2t),
                                       C.byref(w), ora.omat(d_ref), ora.omat(xd_ref))
    assert ok == 1
    dg, ds = pkg.hipabi.DenGraph(g), pkg.hipabi.Supervision(sup)
    nb = hip.chain_workspace_bytes(dg.h, B, T)
    ws = hip.ws(nb)
    ws.fill_(float("nan"))  # nothing may depend on what the workspace held
    res = torch.zeros(8, dtype=torch.float64, device="cuda")
    yd, _ = padded(y)
    dd, _ = padded(np.full_like(y, 5.0))
    xdd = torch.full((T * B, P), 5.0, device="cuda")
    hip.chain_objf_and_deriv(dg.h, ds.h, yd, dev(xo), leaky, l2, 0.1, hip.vec(res), dd, xdd, hip.vec(ws), nb, hip.stream())
    r = host(res)
    assert r[5] == 1.0 and r[2] == w.value
    assert abs(r[0] - objf.value) < 1e-4 * abs(objf.value), (r[0], objf.value)   # BASELINE bar: 1e-4 relative
    assert abs(r[1] - l2t.value) <= 1e-5 * abs(l2t.value) + 1e-12
    assert rel_l2(host(dd), d_ref) < 1e-4
    assert rel_l2(host(xdd), 0.1 * xd_ref) < 1e-4
    assert abs(r[6] - float((xo.astype(np.float64) * xd_ref).sum())) < 1e-4 * max(1.0, abs(r[6]))
    # size-independent property: occupation probabilities of num and den each sum to 1 per frame
    np.testing.assert_allclose(host(xdd).sum(1) / 0.1, 1.0, rtol=1e-4)
    # bitwise reproducible
    dd2 = torch.zeros_like(dd)
    hip.chain_objf_and_deriv(dg.h, ds.h, yd, None, leaky, l2, 0.1, hip.vec(res), dd2, None, hip.vec(ws), nb, hip.stream())
    assert torch.equal(dd2, dd) and host(res)[0] == r[0]


@pytest.mark.parametrize("mode", [1, 2], ids=["persistent", "wide"])
def test_chain_failure_path(hip, ora, pkg, mode):
    pkg.hipabi.check(pkg.hipabi.load().tdnnf_chain_set_denominator_mode(mode))
    try:
        _chain_failure(hip, ora, pkg)
    finally:
        pkg.hipabi.load().tdnnf_chain_set_denominator_mode(0)


def _chain_failure(hip, ora, pkg):
    g = pkg.synth.make_den_graph(30, 20, seed=1)
    sup = pkg.synth.make_supervision(2, 5, 20, seed=1)
    y = np.zeros((10, 20), F)
    y[3, 4] = np.nan
    dg, ds = pkg.hipabi.DenGraph(g), pkg.hipabi.Supervision(sup)
    nb = hip.chain_workspace_bytes(dg.h, 2, 5)
    ws = hip.ws(nb)
    res = torch.zeros(8, dtype=torch.float64, device="cuda")
    d, xd = torch.ones(10, 20, device="cuda"), torch.ones(10, 20, device="cuda")
    hip.chain_objf_and_deriv(dg.h, ds.h, dev(y), None, 0.1, 0.0, 0.1, hip.vec(res), d, xd, hip.vec(ws), nb, hip.stream())
    r = host(res)
    assert r[5] == 0.0 and r[0] == -10.0 * 10 and not host(d).any() and not host(xd).any()


def test_constrain_orthonormal(hip, ora):
    L = ora.lib()
    rng = np.random.default_rng(13)
    for rows, cols, scale in [(160, 3072, -1.0), (256, 1536, -1.0), (48, 200, 1.0)]:
        M = (_rand(rng, rows, cols) / np.sqrt(cols)).astype(F)
        ref = M.copy()
        nb = hip.constrain_orthonormal_workspace_bytes(rows, cols)
        ws = hip.ws(nb)
        Md = dev(M)
        for _ in range(3):
            L.oracle_constrain_orthonormal(scale, ora.fptr(ref), rows, cols, cols)
            hip.constrain_orthonormal(scale, hip.vec(Md), rows, cols, cols, hip.vec(ws), nb, hip.stream())
        assert rel_l2(host(Md), ref) < 1e-5


def test_update_with_max_change(hip, ora):
    L = ora.lib()
    rng = np.random.default_rng(14)
    sizes = [491520, 1536, 3, 160 * 3072, 0, 7]
    begin = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(begin[-1])
    params, delta = _rand(rng, n), (_rand(rng, n) * 2e-3).astype(F)
    delta[begin[1]:begin[2]] *= 50
    mc = np.asarray([0.75, 0.75, 0.0, 0.75, 0.75, 1.5], F)
    dots = np.asarray([float((delta[b:e].astype(np.float64) ** 2).sum()) for b, e in zip(begin[:-1], begin[1:])])
    sf = np.zeros(len(sizes), F)
    ok = C.c_int()
    L.oracle_max_change_scales(ora.dptr(dots), ora.fptr(mc), len(sizes), 2.0, 1.0, 1.0, ora.fptr(sf), C.byref(ok))
    ref = params.copy()
    for i, (b, e) in enumerate(zip(begin[:-1], begin[1:])):
        ref[b:e] += sf[i] * delta[b:e]
    pd, dd = dev(params), dev(delta)
    nb = hip.max_change_workspace_bytes(len(sizes))
    ws = hip.ws(nb)
    info = torch.zeros(len(sizes) + 1, device="cuda")
    hip.update_with_max_change(hip.vec(pd), hip.vec(dd), len(sizes), begin.ctypes.data_as(C.c_void_p),
                               mc.ctypes.data_as(C.c_void_p), 2.0, 1.0, 1.0, 1, hip.vec(ws), nb, hip.vec(info), hip.stream())
    np.testing.assert_allclose(host(info)[:-1], sf, rtol=1e-5)
    assert host(info)[-1] == 1.0 and not host(dd).any()
    np.testing.assert_allclose(host(pd), ref, rtol=1e-6, atol=1e-7)
    x, yv = _rand(rng, 1000), _rand(rng, 1000)
    yd = dev(yv)
    hip.axpy(hip.vec(dev(x)), -0.5, hip.vec(yd), 1000, hip.stream())
    np.testing.assert_allclose(host(yd), yv - 0.5 * x, rtol=1e-6, atol=1e-7)


def test_natural_gradient(hip, ora):
    """Same minibatch sequence through the oracle and the device implementation: X_hat, scale and the
    low-rank state track each other (the R x R eig is in double on both sides)."""
    L = ora.lib()
    rng = np.random.default_rng(15)
    N, D, R = 512, 161, 20
    basis = _rand(rng, 5, D)
    ng_ref = L.oracle_ng_create(R, 4, 2000.0, 4.0)
    ng = C.c_void_p()
    hip.ng_create(R, 4, 2000.0, 4.0, C.byref(ng))
    for it in range(16):
        X = (_rand(rng, N, 5) @ basis * 2 + _rand(rng, N, D) * 0.5).astype(F)
        Xr = X.copy()
        sr = C.c_float()
        L.oracle_ng_precondition(ng_ref, ora.omat(Xr), C.byref(sr))
        Xd, _ = padded(X)
        sd = C.c_float()
        hip.ng_precondition(ng, Xd, C.byref(sd), hip.stream())
        assert rel_l2(host(Xd), Xr) < 2e-3, it
        assert abs(sd.value - sr.value) < 2e-3 * sr.value, it
    hip.lib.tdnnf_ng_destroy(ng)
    L.oracle_ng_destroy(ng_ref)


# the recursions with several workgroups per sequence (few sequences: chain.hip, den_mw_kernel), against the oracle like the other forms
# and against the one-workgroup kernels; four workgroups per sequence (B = 16 / 8: a sequence's workgroups on one XCD; B = 12: the plain block order)
@pytest.mark.parametrize("H,P,B,T,leaky", [(4000, 6034, 16, 40, 0.1), (1500, 700, 8, 60, 1e-5), (900, 400, 12, 30, 0.05), (2100, 900, 24, 25, 0.1)])
def test_chain_denominator_several_workgroups_per_sequence(hip, ora, pkg, H, P, B, T, leaky):
    pkg.hipabi.check(pkg.hipabi.load().tdnnf_chain_set_denominator_mode(1))
    try:
        _chain_case(hip, ora, pkg, H, P, B, T, leaky, 0.0)  # (the entry point takes the split form, whose recursions these are)
        got = _chain_deriv(hip, pkg, H, P, B, T, leaky)
        pkg.hipabi.check(pkg.hipabi.load().tdnnf_chain_set_denominator_mode(3))  # one workgroup per sequence
        one = _chain_deriv(hip, pkg, H, P, B, T, leaky)
    finally:
        pkg.hipabi.load().tdnnf_chain_set_denominator_mode(0)
    assert not np.array_equal(got[1], one[1])  # the switch did switch: other summation orders
    assert abs(got[0] - one[0]) < 1e-6 * abs(one[0])
    assert rel_l2(got[1], one[1]) < 1e-5


def test_multi_workgroup_denominator_gives_up_cleanly(hip, ora, pkg):
    """a multi-workgroup launch whose workgroups are not co-resident must not cost the minibatch.  With the test hook
    den_mw_test_abort the recursions see a raised abort word at their first poll; the one-workgroup kernels launched behind them then redo
    both recursions (same kernels as mode 3: the result is bit-identical to it and matches the oracle), the fallback is counted, and the
    next call reports it and leaves the multi-workgroup form alone for the rest of the process."""
    lib = pkg.hipabi.load()
    H, P, B, T, leaky = 1500, 700, 8, 60, 0.05
    fb, off = C.c_int(), C.c_int()
    pkg.hipabi.check(lib.tdnnf_chain_den_mw_status(C.byref(fb), C.byref(off), 1))
    try:
        pkg.hipabi.check(lib.tdnnf_chain_set_denominator_mode(3))
        one = _chain_deriv(hip, pkg, H, P, B, T, leaky)
        pkg.hipabi.check(lib.tdnnf_chain_set_denominator_mode(1))
        with pkg.hipabi.option("den_mw_test_abort", 1):
            got = _chain_deriv(hip, pkg, H, P, B, T, leaky)
        assert np.isfinite(got[1]).all() and got[0] == one[0] and np.array_equal(got[1], one[1])
        pkg.hipabi.check(lib.tdnnf_chain_den_mw_status(C.byref(fb), C.byref(off), 0))
        assert fb.value == 1 and off.value == 0  # counted; reported and switched off by the NEXT call
        again = _chain_deriv(hip, pkg, H, P, B, T, leaky)
        pkg.hipabi.check(lib.tdnnf_chain_den_mw_status(C.byref(fb), C.byref(off), 0))
        assert off.value == 1 and np.array_equal(again[1], one[1])
        _chain_case(hip, ora, pkg, H, P, B, T, leaky, 0.0)  # and against the oracle, like every other form
    finally:
        pkg.hipabi.check(lib.tdnnf_chain_den_mw_status(C.byref(fb), C.byref(off), 1))
        lib.tdnnf_chain_set_denominator_mode(0)


def _chain_deriv(hip, pkg, H, P, B, T, leaky):
    g = pkg.synth.make_den_graph(H, P, mean_out_degree=6.0, seed=H)
    sup = pkg.synth.make_supervision(B, T, P, seed=T, weight=1.0)
    rng = np.random.default_rng(H + T)
    y = (_rand(rng, T * B, P) * 1.5).astype(F)
    xo = _rand(rng, T * B, P)
    dg, ds = pkg.hipabi.DenGraph(g), pkg.hipabi.Supervision(sup)
    nb = hip.chain_workspace_bytes(dg.h, B, T)
    ws = hip.ws(nb)
    ws.fill_(float("nan"))
    res = torch.zeros(8, dtype=torch.float64, device="cuda")
    yd, _ = padded(y)
    dd, _ = padded(np.full_like(y, 5.0))
    xdd = torch.full((T * B, P), 5.0, device="cuda")
    hip.chain_objf_and_deriv(dg.h, ds.h, yd, dev(xo), leaky, 0.0, 0.1, hip.vec(res), dd, xdd, hip.vec(ws), nb, hip.stream())
    return host(res)[0], host(dd)


# the persistent LDS-DMA-ring form of the rows GEMM (csrc/gemm_ring.hip) takes launches without tap coefficients whose taps are whole K
# steps of 16 and whose output gets the 128-wide tile: every init mode, ragged row and column tiles, row strides, a block that walks
# several tiles (more tiles than the chip has block slots), against a float64 product of the spliced input
RING_CASES = [
    # name, offsets, num_t_out, B, Di, Do, t_step_out
    ("affine-k2", [0, 1], 37, 16, 160, 1536, 1),          # 592 rows: ragged last row tile
    ("affine-k2-stride3", [0, 3], 11, 24, 160, 1536, 3),
    ("ragged-columns", [-1, 0, 1], 9, 13, 48, 200, 1),    # 117 rows, 200 columns: one tile row, a 72-column tile
    ("output-layer", [0], 5, 31, 256, 6034, 1),
    ("many-tiles", [0], 400, 128, 32, 4096, 1),           # 51 200 x 4 096: 12 800 tiles on 768 slots
]


@pytest.mark.parametrize("case", RING_CASES, ids=[c[0] for c in RING_CASES])
def test_rows_gemm_persistent_ring_shapes(hip, pkg, case):
    name, offs, nt, B, Di, Do, step = case
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B, t_step_out=step)
    K = len(offs)
    x = _rand(rng, rows_in, Di)
    W = (_rand(rng, Do, K * Di) / np.sqrt(K * Di)).astype(F)
    b = _rand(rng, Do)
    y0 = _rand(rng, N, Do)
    xt, Wt = torch.from_numpy(x).cuda().double(), torch.from_numpy(W).cuda().double()
    ref = torch.zeros(N, Do, dtype=torch.float64, device="cuda")
    out_rows = torch.arange(N, device="cuda")
    for k in range(K):
        src = int(ro[k]) + out_rows * rho  # (PrecomputeIndexes: input row = row_offset + row_stride * output row)
        ref += xt[src] @ Wt[:, k * Di:(k + 1) * Di].T
    ix = pkg.hipabi.indexes(rho, ro)
    xd, _ = padded(x)
    Wd, bd = dev(W), dev(b)
    for mode in (1, 0, 2):
        yd, ybuf = padded(y0)
        hip.tdnn_propagate(C.byref(ix), xd, hip.vec(Wd), K * Di, Do, Di, hip.vec(bd) if mode == 1 else None, None, mode, yd, hip.stream())
        want = ref + (torch.from_numpy(b).cuda().double() if mode == 1 else 0) + (torch.from_numpy(y0).cuda().double() if mode == 0 else 0)
        got = yd.double()
        assert float((got - want).norm() / want.norm()) < TOL, (name, mode)
        assert (host(ybuf)[:, Do:] == 7.0).all(), "wrote outside the view"
