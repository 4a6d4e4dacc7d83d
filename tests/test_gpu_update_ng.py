"""-m gpu: the component-level natural-gradient updates behind the C-ABI -- the branch Backprop takes in every recipe
(nnet-tdnn-component.cc:427-430) -- against the oracle's LITERAL restatement of
TdnnDARTSV3Component::UpdateNaturalGradient (nnet-tdnn-component.cc:457-626) and
NaturalGradientAffineComponent::Update (nnet-simple-component.cc:2980-3024): per-tap dots -> logit update, splice x
coefficients, PreconditionDirections on both copies, scaled AddMatMat.  Several calls in a row, so the preconditioners'
state (refreshed on each of the first ten calls) evolves on both sides."""
import ctypes as C
import zlib

import numpy as np
import pytest
import torch

from tests.gpu_util import F, Hip, dev, host, padded, rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(pkg):
    return Hip(pkg)


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(F)


def _oracle_update_ng(L, ora, x, dy, W, rho, ro, K, Di, Do, coef, eff, flags, share, temp, ng_in, ng_out, lr, Wacc, bacc, aacc):
    """the literal order of the reference, from the oracle's pieces"""
    N = dy.shape[0]
    if coef is not None:
        s = np.zeros(K)
        if not (flags & 4):
            L.oracle_tdnn_darts_tap_dots(ora.omat(x), ora.omat(dy), ora.fptr(W), W.shape[1], Do, Di, K, rho, ora.iptr(ro), ora.dptr(s))
        L.oracle_tdnn_darts_alpha_update(ora.dptr(s), ora.fptr(coef), K, flags, share, temp, lr, ora.fptr(aacc))
    ones = 1 if bacc is not None else 0
    X = np.zeros((N, K * Di + ones), F)
    L.oracle_tdnn_splice(ora.omat(x), N, Di, K, rho, ora.iptr(ro), ora.fptr(eff) if eff is not None else None, ones, ora.omat(X))
    Y = dy.copy()
    a, b = C.c_float(1.0), C.c_float(1.0)
    L.oracle_ng_precondition(ng_in, ora.omat(X), C.byref(a))
    L.oracle_ng_precondition(ng_out, ora.omat(Y), C.byref(b))
    sc = F(a.value * b.value) * F(lr)
    Xw = np.ascontiguousarray(X[:, :K * Di])
    L.oracle_affine_update_simple(ora.omat(Xw), ora.omat(Y), float(sc), ora.fptr(Wacc), Wacc.shape[1], None)
    if ones:
        bacc += (sc * (Y.astype(np.float64) * X[:, -1:].astype(np.float64)).sum(0)).astype(F)


CASES = [
    # name, offsets, B, num_t_out, Di, Do, flags (None = plain TdnnComponent), bias
    ("plain-linear-nobias", [-1, 0], 16, 12, 96, 40, None, False),
    ("plain-affine-bias", [0, 3], 8, 30, 40, 192, None, True),
    ("darts-softmax", [-2, -1, 0], 8, 24, 64, 32, 0, True),
    ("darts-gumbel-entropy-updatealpha", [0, 1, 2, 3], 8, 24, 32, 64, 1 | 8 | 16, True),
    ("darts-freeselect", [-2, -1, 0], 8, 24, 64, 32, 2, True),
    ("darts-uniform-pretrain-k7", [-6, -5, -4, -3, -2, -1, 0], 8, 24, 48, 32, 4, True),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_tdnn_update_natural_gradient(hip, ora, pkg, case):
    name, offs, B, nt, Di, Do, flags, bias = case
    L = ora.lib()
    rng = np.random.default_rng(zlib.crc32(name.encode()) % 1000)
    K = len(offs)
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B)
    ix = pkg.hipabi.indexes(rho, ro)
    darts = flags is not None
    fl = flags or 0
    share = L.oracle_tdnn_share_index(ora.iptr(np.asarray(offs, np.int32)), K)
    W = (_rand(rng, Do, K * Di) / np.sqrt(K * Di)).astype(F)
    Dx = K * Di + (1 if bias else 0)
    rin, rout = min(20, (Dx + 1) // 2), min(80, (Do + 1) // 2)
    ngi_ref, ngo_ref = L.oracle_ng_create(rin, 4, 2000.0, 4.0), L.oracle_ng_create(rout, 4, 2000.0, 4.0)
    ngi, ngo = C.c_void_p(), C.c_void_p()
    hip.ng_create(rin, 4, 2000.0, 4.0, C.byref(ngi))
    hip.ng_create(rout, 4, 2000.0, 4.0, C.byref(ngo))
    nb = hip.lib.tdnnf_tdnn_update_natural_gradient_workspace_bytes(Do, Di, K, N, int(bias))
    ws = hip.ws(nb)
    Wd = dev(W)
    lr = 0.01
    basis_x, basis_y = _rand(rng, 6, Di), _rand(rng, 5, Do)
    for it in range(6):
        x = (_rand(rng, rows_in, 6) @ basis_x + 0.5 * _rand(rng, rows_in, Di)).astype(F)
        dy = (_rand(rng, N, 5) @ basis_y + 0.5 * _rand(rng, N, Do)).astype(F)
        coef = eff = None
        if darts:
            la, u = _rand(rng, K) * 0.5, rng.uniform(0.05, 0.95, K).astype(F)
            su = float(rng.uniform(0.05, 0.95))
            coef, eff = np.zeros(K, F), np.zeros(K, F)
            L.oracle_tdnn_darts_coef(ora.fptr(la), K, fl, 0.7, ora.fptr(u), su, ora.fptr(coef))
            L.oracle_tdnn_darts_effective_coef(ora.fptr(coef), K, fl, share, ora.fptr(eff))
        W0, b0, a0 = _rand(rng, Do, K * Di) * 0.1, (_rand(rng, Do) * 0.1 if bias else None), (_rand(rng, K) * 0.1 if darts else None)
        W_ref, b_ref, a_ref = W0.copy(), (b0.copy() if bias else None), (a0.copy() if darts else None)
        _oracle_update_ng(L, ora, x, dy, W, rho, ro, K, Di, Do, coef, eff, fl, share, 0.7, ngi_ref, ngo_ref, lr, W_ref, b_ref, a_ref)
        xd, _ = padded(x)
        dyd, _ = padded(dy)
        Wacc = dev(W0)
        bacc = dev(b0) if bias else None
        aacc = dev(a0) if darts else None
        cd, ed = (dev(coef), dev(eff)) if darts else (None, None)
        hip.tdnn_update_natural_gradient(C.byref(ix), xd, dyd, Do, Di, hip.vec(Wd), K * Di, hip.vec(cd) if darts else None,
                                         hip.vec(ed) if darts else None, fl, share, 0.7, ngi, ngo, lr, hip.vec(Wacc), K * Di,
                                         hip.vec(bacc) if bias else None, hip.vec(aacc) if darts else None, hip.vec(ws), nb, hip.stream())
        assert np.linalg.norm(W_ref - W0) > 0
        assert rel_l2(host(Wacc) - W0, W_ref - W0) < 5e-3, (it, rel_l2(host(Wacc) - W0, W_ref - W0))
        if bias:
            assert rel_l2(host(bacc) - b0, b_ref - b0) < 5e-3, (it, rel_l2(host(bacc) - b0, b_ref - b0))
        if darts:
            assert rel_l2(host(aacc), a_ref) < 1e-4, (it, host(aacc), a_ref)
    hip.lib.tdnnf_ng_destroy(ngi)
    hip.lib.tdnnf_ng_destroy(ngo)
    L.oracle_ng_destroy(ngi_ref)
    L.oracle_ng_destroy(ngo_ref)


@pytest.mark.parametrize("bias", [True, False], ids=["NaturalGradientAffine", "Linear"])
def test_affine_update_natural_gradient(hip, ora, pkg, bias):
    L = ora.lib()
    rng = np.random.default_rng(31 + bias)
    N, Di, Do = 640, 96, 160
    Dx = Di + (1 if bias else 0)
    rin, rout = min(20, (Dx + 1) // 2), min(80, (Do + 1) // 2)
    ngi_ref, ngo_ref = L.oracle_ng_create(rin, 4, 2000.0, 4.0), L.oracle_ng_create(rout, 4, 2000.0, 4.0)
    ngi, ngo = C.c_void_p(), C.c_void_p()
    hip.ng_create(rin, 4, 2000.0, 4.0, C.byref(ngi))
    hip.ng_create(rout, 4, 2000.0, 4.0, C.byref(ngo))
    nb = hip.lib.tdnnf_affine_update_natural_gradient_workspace_bytes(Do, Di, N, int(bias))
    ws = hip.ws(nb)
    ro = np.zeros(1, np.int32)
    bx, by = _rand(rng, 4, Di), _rand(rng, 7, Do)
    for it in range(6):
        x = (_rand(rng, N, 4) @ bx + 0.3 * _rand(rng, N, Di)).astype(F)
        dy = (_rand(rng, N, 7) @ by + 0.3 * _rand(rng, N, Do)).astype(F)
        W0, b0 = _rand(rng, Do, Di) * 0.1, (_rand(rng, Do) * 0.1 if bias else None)
        W_ref, b_ref = W0.copy(), (b0.copy() if bias else None)
        _oracle_update_ng(L, ora, x, dy, None, 1, ro, 1, Di, Do, None, None, 0, 0, 1.0, ngi_ref, ngo_ref, 0.02, W_ref, b_ref, None)
        Wacc, bacc = dev(W0), (dev(b0) if bias else None)
        xd, _ = padded(x)
        dyd, _ = padded(dy)
        hip.affine_update_natural_gradient(xd, dyd, ngi, ngo, 0.02, hip.vec(Wacc), Di, hip.vec(bacc) if bias else None, hip.vec(ws), nb, hip.stream())
        assert rel_l2(host(Wacc) - W0, W_ref - W0) < 5e-3, it
        if bias:
            assert rel_l2(host(bacc) - b0, b_ref - b0) < 5e-3, it
    # learning rate 0: nothing happens, the preconditioners are not advanced (:423-424 returns before the update)
    Wacc = dev(W0)
    hip.affine_update_natural_gradient(xd, dyd, ngi, ngo, 0.0, hip.vec(Wacc), Di, None, hip.vec(ws), nb, hip.stream())
    assert np.array_equal(host(Wacc), W0)
    # too small a workspace is an argument error, nothing is launched
    rc = hip.lib.tdnnf_affine_update_natural_gradient(pkg.hipabi.pmat(xd), pkg.hipabi.pmat(dyd), ngi, ngo, 0.02, pkg.hipabi.ptr(Wacc), Di, None,
                                                      pkg.hipabi.ptr(ws), 1024, None)
    assert rc == 1 and b"workspace too small" in hip.lib.tdnnf_last_error()
    hip.lib.tdnnf_ng_destroy(ngi)
    hip.lib.tdnnf_ng_destroy(ngo)
    L.oracle_ng_destroy(ngi_ref)
    L.oracle_ng_destroy(ngo_ref)


def test_onehot_backprop(hip, ora, pkg):
    """OnehotFunctionComponent::Backprop nnet-simple-component.cc:9539-9548: output_ += lr * colsum(out_deriv)."""
    rng = np.random.default_rng(3)
    N, Cn = 4000, 8
    d = _rand(rng, N, Cn)
    acc0 = _rand(rng, Cn)
    want = acc0 + 0.25 * d.astype(np.float64).sum(0)
    dd, _ = padded(d)
    acc = dev(acc0)
    nb = hip.lib.tdnnf_colreduce_workspace_bytes(N, Cn)
    ws = hip.ws(nb)
    hip.onehot_backprop(dd, 0.25, hip.vec(acc), hip.vec(ws), nb, hip.stream())
    assert rel_l2(host(acc), want) < 2e-6
