"""Pins the CPU oracle (oracle/*.c) for the component rows A1-A6 of SURVEY.md 8(a)
by independent checks: PyTorch-CPU float64 autograd of the same maths and
finite differences.  The reference has no tests or golden vectors,
so this is the strongest pin available: "parity unpinned" against real Kaldi.
"""
import ctypes as C

import numpy as np
import pytest
import torch

F = np.float32


def _rand(rng, *shape):
    return rng.standard_normal(shape).astype(F)


def _views(x, rho, offs, N):
    """torch version of GetInputPart (nnet-tdnn-component.cc:806-820)."""
    return [x[o:o + rho * (N - 1) + 1:rho] for o in offs]


CASES = [
    # (time_offsets, num_t_out, B, Di, Do, t_step_out)
    ([-1, 0, 1], 7, 3, 8, 5, 1),
    ([-3, 0], 6, 4, 12, 7, 3),
    ([0, 3], 5, 2, 6, 9, 3),
    ([0], 4, 3, 5, 4, 1),
    ([-6, -5, -4, -3, -2, -1, 0], 9, 2, 4, 6, 1),
]


@pytest.mark.parametrize("offs,nt,B,Di,Do,step", CASES)
def test_tdnn_fwd_bwd_vs_torch(ora, pkg, offs, nt, B, Di, Do, step):
    L = ora.lib()
    rng = np.random.default_rng(1)
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B, t_step_out=step)
    K = len(offs)
    x = _rand(rng, rows_in, Di)
    W = _rand(rng, Do, K * Di)
    b = _rand(rng, Do)
    c = (rng.random(K) + 0.1).astype(F)
    dy = _rand(rng, N, Do)
    y = np.zeros((N, Do), F)
    L.oracle_tdnn_propagate(ora.omat(x), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro),
                            ora.fptr(b), ora.fptr(c), 1, ora.omat(y))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    yt = bt + sum(float(c[i]) * v @ Wt[:, i * Di:(i + 1) * Di].T
                  for i, v in enumerate(_views(xt, rho, ro, N)))
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=2e-5, atol=2e-5)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    dx = _rand(rng, rows_in, Di)  # kBackpropAdds: pre-existing content must be kept
    dx0 = dx.copy()
    L.oracle_tdnn_backprop_data(ora.omat(dy), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro),
                                ora.fptr(c), ora.omat(dx))
    np.testing.assert_allclose(dx - dx0, xt.grad.numpy(), rtol=2e-5, atol=2e-5)
    Wacc = np.zeros_like(W)
    bacc = np.zeros_like(b)
    L.oracle_tdnn_update_simple(ora.omat(x), ora.omat(dy), Do, Di, K, rho, ora.iptr(ro), ora.fptr(c),
                                0.5, ora.fptr(Wacc), K * Di, ora.fptr(bacc))
    np.testing.assert_allclose(Wacc, 0.5 * Wt.grad.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(bacc, 0.5 * bt.grad.numpy(), rtol=2e-5, atol=2e-5)


def test_tdnn_equals_dilated_conv1d(ora, pkg):
    """Tdnn with offsets {-s,0} == conv1d(dilation=s) per sequence (SURVEY 8(c) KAT 2)."""
    L = ora.lib()
    rng = np.random.default_rng(2)
    B, T, Di, Do, s = 3, 11, 6, 4, 2
    offs = [-s, 0]
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, T, B)
    x = _rand(rng, rows_in, Di)
    W = _rand(rng, Do, 2 * Di)
    y = np.zeros((N, Do), F)
    L.oracle_tdnn_propagate(ora.omat(x), ora.fptr(W), 2 * Di, Do, Di, 2, rho, ora.iptr(ro), None, None,
                            2, ora.omat(y))
    xs = torch.tensor(x).reshape(T + s, B, Di).permute(1, 2, 0)  # B x Di x time
    w = torch.tensor(W).reshape(Do, 2, Di).permute(0, 2, 1)  # Do x Di x taps
    ref = torch.nn.functional.conv1d(xs.double(), w.double(), dilation=s)  # B x Do x T
    np.testing.assert_allclose(y.reshape(T, B, Do), ref.permute(2, 0, 1).numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("flags,name", [(1, "gumbel"), (2, "free_select"), (0, "softmax"),
                                         (4, "uniform"), (1 | 4, "gumbel+uniform")])
def test_darts_coef_modes(ora, flags, name):
    L = ora.lib()
    rng = np.random.default_rng(3)
    K = 7
    la = _rand(rng, K)
    u = rng.random(K).astype(F)
    coef = np.zeros(K, F)
    L.oracle_tdnn_darts_coef(ora.fptr(la), K, flags, 0.7, ora.fptr(u), 0.45, ora.fptr(coef))
    if flags & 4:
        assert coef.sum() == 1.0 and coef[int(0.45 * K)] == 1.0
    elif flags & 1:
        g = -np.log(-np.log(u.astype(np.float64)))
        z = (la + g) / 0.7
        ref = np.exp(z - z.max()) / np.exp(z - z.max()).sum()
        np.testing.assert_allclose(coef, ref, rtol=1e-5)
    elif flags & 2:
        np.testing.assert_allclose(coef, 1 / (1 + np.exp(-la.astype(np.float64))), rtol=1e-6)
    else:
        ref = np.exp(la - la.max()) / np.exp(la - la.max()).sum()
        np.testing.assert_allclose(coef, ref, rtol=1e-5)
    eff = np.zeros(K, F)
    L.oracle_tdnn_darts_effective_coef(ora.fptr(coef), K, flags, K - 1, ora.fptr(eff))
    if flags & 4:
        assert eff[K - 1] == 1.0 and eff[int(0.45 * K)] == 1.0 and eff.sum() <= 2.0
    elif flags & 2:
        np.testing.assert_array_equal(eff, coef)
    else:
        assert eff[K - 1] == 1.0
        np.testing.assert_array_equal(eff[:-1], coef[:-1])


@pytest.mark.parametrize("flags", [0, 1, 2])
def test_darts_alpha_grad_is_true_gradient(ora, pkg, flags):
    """The alpha update of nnet-tdnn-component.cc:516-561 is (before the x5/lr
    scalings) the gradient of <Y, dY> w.r.t. log-alpha with the share tap held at
    weight 1: check against autograd."""
    L = ora.lib()
    rng = np.random.default_rng(4)
    offs = [-2, -1, 0]
    K, B, Di, Do, nt = 3, 2, 5, 4, 6
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B)
    share = L.oracle_tdnn_share_index(ora.iptr(np.asarray(offs, np.int32)), K)
    assert share == K - 1
    x, W, dy = _rand(rng, rows_in, Di), _rand(rng, Do, K * Di), _rand(rng, N, Do)
    la = _rand(rng, K)
    u = rng.random(K).astype(F)
    tau = 0.6
    coef = np.zeros(K, F)
    L.oracle_tdnn_darts_coef(ora.fptr(la), K, flags, tau, ora.fptr(u), 0.0, ora.fptr(coef))
    s = np.zeros(K)
    L.oracle_tdnn_darts_tap_dots(ora.omat(x), ora.omat(dy), ora.fptr(W), K * Di, Do, Di, K, rho,
                                 ora.iptr(ro), ora.dptr(s))
    acc = np.zeros(K, F)
    lr = 0.25
    L.oracle_tdnn_darts_alpha_update(ora.dptr(s), ora.fptr(coef), K, flags, share, tau, lr, ora.fptr(acc))
    lat = torch.tensor(la, dtype=torch.float64, requires_grad=True)
    if flags & 1:
        g = -torch.log(-torch.log(torch.tensor(u, dtype=torch.float64)))
        c = torch.softmax((lat + g) / tau, 0)
    elif flags & 2:
        c = torch.sigmoid(lat)
    else:
        c = torch.softmax(lat, 0)
    xt, Wt = torch.tensor(x, dtype=torch.float64), torch.tensor(W, dtype=torch.float64)
    y = 0
    for i, v in enumerate(_views(xt, rho, ro, N)):
        ci = c[i] if (flags & 2 or i != share) else 1.0
        y = y + ci * v @ Wt[:, i * Di:(i + 1) * Di].T
    (y * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    mul = (5 * lr) if not (flags & 1) else lr  # :574-586 (use_entropy/update_alpha off here)
    np.testing.assert_allclose(acc, mul * lat.grad.numpy(), rtol=2e-4, atol=2e-5)


def test_batchnorm_vs_torch(ora):
    L = ora.lib()
    rng = np.random.default_rng(5)
    N, D, eps, rms = 37, 11, 1e-3, 1.0
    x = _rand(rng, N, D) * 2 + 0.3
    dz = _rand(rng, N, D)
    z = np.zeros_like(x)
    memo = np.zeros((5, D), F)
    L.oracle_batchnorm_propagate(ora.omat(x), eps, rms, ora.omat(z), ora.fptr(memo))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    zt = torch.nn.functional.batch_norm(xt, None, None, training=True, eps=eps)
    np.testing.assert_allclose(z, zt.detach().numpy(), rtol=1e-4, atol=1e-5)
    zt.backward(torch.tensor(dz, dtype=torch.float64))
    dx = np.zeros_like(x)
    L.oracle_batchnorm_backprop(ora.omat(z), ora.omat(dz), rms, ora.fptr(memo), ora.omat(dx))
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=2e-4, atol=2e-5)
    # stats + derived + test-mode forward == train-mode forward on the same batch
    cnt = C.c_double(0.0)
    ssum, ssq = np.zeros(D), np.zeros(D)
    L.oracle_batchnorm_store_stats(ora.fptr(memo), D, N, C.byref(cnt), ora.dptr(ssum), ora.dptr(ssq))
    assert cnt.value == N
    scale, offset = np.zeros(D, F), np.zeros(D, F)
    L.oracle_batchnorm_compute_derived(cnt.value, ora.dptr(ssum), ora.dptr(ssq), D, eps, rms,
                                       ora.fptr(scale), ora.fptr(offset))
    z2 = np.zeros_like(x)
    L.oracle_batchnorm_test_propagate(ora.omat(x), ora.fptr(scale), ora.fptr(offset), ora.omat(z2))
    np.testing.assert_allclose(z2, z, rtol=1e-4, atol=1e-5)
    dx2 = np.zeros_like(x)
    L.oracle_batchnorm_test_backprop(ora.omat(dz), ora.fptr(scale), ora.omat(dx2))
    np.testing.assert_allclose(dx2, dz * scale, rtol=1e-6)


@pytest.mark.parametrize("gumbel", [False, True])
def test_softmax_flops_vs_torch(ora, gumbel):
    L = ora.lib()
    rng = np.random.default_rng(6)
    N, Cc, eta, tau = 9, 8, 0.3, (0.5 if gumbel else 1.0)
    row = _rand(rng, 1, Cc)
    x = np.repeat(row, N, 0)  # rows identical, as in the recipes
    u = rng.random(Cc).astype(F) if gumbel else None
    p = np.zeros_like(x)
    L.oracle_softmax_flops_propagate(ora.omat(x), ora.fptr(u), tau, ora.omat(p))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    g = -torch.log(-torch.log(torch.tensor(u, dtype=torch.float64))) if gumbel else 0.0
    pt = torch.softmax((xt + g) / tau, 1)
    np.testing.assert_allclose(p, pt.detach().numpy(), rtol=1e-5)
    flops = -np.asarray([25, 50, 80, 100, 120, 160, 200, 240], F)
    dp = _rand(rng, N, Cc)
    dp_in = dp.copy()
    dx = np.zeros_like(x)
    L.oracle_softmax_flops_backprop(ora.omat(p), ora.omat(dp_in), eta, ora.fptr(flops), Cc, tau, ora.omat(dx))
    # reference mutates out_deriv in place (nnet-simple-component.cc:10016)
    np.testing.assert_allclose(dp_in, dp + eta / N / Cc * flops, rtol=1e-6)
    pt.backward(torch.tensor(dp_in, dtype=torch.float64))
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-4, atol=1e-6)


def test_onehot_copyn_product_constant(ora):
    L = ora.lib()
    rng = np.random.default_rng(7)
    for u in [0.0, 0.1249, 0.125, 0.5, 0.99999]:
        idx = L.oracle_onehot_index(u, 8)
        assert idx == min(int(np.float32(u) * 8), 7) or abs(u * 8 - round(u * 8)) < 1e-3
    out = np.zeros((4, 8), F)
    L.oracle_onehot_propagate(0.3, ora.omat(out))
    assert (out.sum(1) == 1).all() and (out[:, 2] == 1).all()
    a = _rand(rng, 5, 1)
    o = np.ones((5, 6), F)
    L.oracle_copyn_propagate(ora.omat(a), 2.0, ora.omat(o))
    np.testing.assert_allclose(o, 1 + 2 * np.repeat(a, 6, 1), rtol=1e-6)
    do = _rand(rng, 5, 6)
    da = np.zeros((5, 1), F)
    L.oracle_copyn_backprop(ora.omat(do), 2.0, ora.omat(da))
    np.testing.assert_allclose(da[:, 0], 2 * do.sum(1), rtol=1e-5)
    x = _rand(rng, 5, 6)
    y = np.zeros((5, 3), F)
    L.oracle_elementwise_product_propagate(ora.omat(x), 3, ora.omat(y))
    np.testing.assert_allclose(y, x[:, :3] * x[:, 3:], rtol=1e-6)
    dyy = _rand(rng, 5, 3)
    dxx = np.zeros_like(x)
    L.oracle_elementwise_product_backprop(ora.omat(x), ora.omat(dyy), 3, ora.omat(dxx))
    np.testing.assert_allclose(dxx, np.concatenate([dyy * x[:, 3:], dyy * x[:, :3]], 1), rtol=1e-6)
    alpha = _rand(rng, 8)
    out = np.zeros((4, 8), F)
    L.oracle_constant_function_propagate(ora.fptr(alpha), ora.omat(out))
    assert (out == alpha).all()
    acc = np.zeros(8, F)
    d = _rand(rng, 4, 8)
    L.oracle_constant_function_backprop(ora.omat(d), 0.1, ora.fptr(acc))
    np.testing.assert_allclose(acc, 0.5 * d.sum(0), rtol=1e-5)


def test_relu_affine_logsoftmax_vs_torch(ora):
    L = ora.lib()
    rng = np.random.default_rng(8)
    N, Di, Do = 13, 7, 5
    x, W, b, dy = _rand(rng, N, Di), _rand(rng, Do, Di), _rand(rng, Do), _rand(rng, N, Do)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wt = torch.tensor(W, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    yt = torch.log_softmax(torch.relu(xt @ Wt.T + bt), 1)
    yt.backward(torch.tensor(dy, dtype=torch.float64))
    a = np.zeros((N, Do), F)
    L.oracle_affine_propagate(ora.omat(x), ora.fptr(W), Di, ora.fptr(b), Do, ora.omat(a))
    r = np.zeros_like(a)
    L.oracle_relu_propagate(ora.omat(a), ora.omat(r))
    y = np.zeros_like(a)
    L.oracle_log_softmax_propagate(ora.omat(r), ora.omat(y))
    np.testing.assert_allclose(y, yt.detach().numpy(), rtol=1e-5, atol=1e-5)
    dr, da, dx = np.zeros_like(a), np.zeros_like(a), np.zeros_like(x)
    L.oracle_log_softmax_backprop(ora.omat(y), ora.omat(dy), ora.omat(dr))
    L.oracle_relu_backprop(ora.omat(r), ora.omat(dr), ora.omat(da))
    L.oracle_affine_backprop(ora.omat(da), ora.fptr(W), Di, Di, ora.omat(dx))
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=1e-4, atol=1e-5)
    Wacc, bacc = np.zeros_like(W), np.zeros_like(b)
    L.oracle_affine_update_simple(ora.omat(x), ora.omat(da), 1.0, ora.fptr(Wacc), Di, ora.fptr(bacc))
    np.testing.assert_allclose(Wacc, Wt.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bacc, bt.grad.numpy(), rtol=1e-4, atol=1e-5)


def test_relu_repair_and_stats(ora):
    L = ora.lib()
    out = np.asarray([[0, 1, 2], [0, 3, 0], [0, 1, 0], [0, 2, 0]], F)
    vs, ds = np.zeros(3), np.zeros(3)
    cnt = C.c_double(0)
    L.oracle_relu_store_stats(ora.omat(out), ora.dptr(vs), ora.dptr(ds), C.byref(cnt))
    assert cnt.value == 4 and list(ds) == [0, 4, 1] and list(vs) == [0, 7, 2]
    d = np.zeros((2, 3), F)
    L.oracle_relu_repair(ora.dptr(ds), cnt.value, 3, 1e-5, 0.05, 0.95, ora.omat(d))
    # col0 never active -> +scale/0.5; col1 always active -> -scale/0.5; col2 untouched
    np.testing.assert_allclose(d[0], [2e-5, -2e-5, 0], rtol=1e-6)
