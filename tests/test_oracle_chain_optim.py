"""Pins the oracle's UPSTREAM restatements: chain
denominator / numerator vs brute-force path enumeration and torch float64
autograd, the semi-orthogonal step, max-change / L2, and natural gradient
identities.  "Parity unpinned": there is no reference text for A7/A8."""
import ctypes as C
import itertools

import numpy as np
import pytest
import torch

F = np.float32


def _den_logprob_torch(g, y, B, leaky):
    """log p_den per sequence, dense float64 autograd (no renormalisation, no clamp)."""
    H = g["H"]
    T = y.shape[0] // B
    init = torch.tensor(g["init"], dtype=torch.float64)
    src, dst = torch.tensor(g["src"], dtype=torch.long), torch.tensor(g["dst"], dtype=torch.long)
    pdf = torch.tensor(g["pdf"], dtype=torch.long)
    prob = torch.tensor(g["prob"], dtype=torch.float64)
    tot = []
    for s in range(B):
        a = init + leaky * init.sum() * init
        for t in range(T):
            x = torch.exp(y[t * B + s])
            contrib = a[src] * prob * x[pdf]
            a = torch.zeros(H, dtype=torch.float64).index_add(0, dst, contrib)
            a = a + leaky * a.sum() * init
        tot.append(torch.log(a.sum()))
    return torch.stack(tot)


def test_denominator_vs_brute_force_paths(ora, pkg):
    """H=3, T=3, no leaky transitions: enumerate every state path."""
    L = ora.lib()
    g = pkg.synth.make_den_graph(3, 4, mean_out_degree=2.0, seed=11)
    rng = np.random.default_rng(0)
    B, T, P = 2, 3, 4
    y = rng.standard_normal((T * B, P)).astype(F)
    tot = C.c_double()
    gs = ora.den_graph_struct(g)
    L.oracle_chain_denominator(C.byref(gs), ora.omat(y), B, 0.0, 0.0, C.byref(tot), None)
    ref = 0.0
    arcs = list(zip(g["src"], g["dst"], g["pdf"], g["prob"]))
    for s in range(B):
        p = 0.0
        for h0 in range(3):
            for path in itertools.product(range(len(arcs)), repeat=T):
                w, cur, ok = float(g["init"][h0]), h0, True
                for t, ai in enumerate(path):
                    a = arcs[ai]
                    if a[0] != cur:
                        ok = False
                        break
                    w *= float(a[3]) * np.exp(float(y[t * B + s, a[2]]))
                    cur = a[1]
                if ok:
                    p += w
        ref += np.log(p)
    assert abs(tot.value - ref) < 1e-5 * max(1, abs(ref))


@pytest.mark.parametrize("leaky", [0.0, 0.1])
def test_denominator_fwd_bwd_vs_autograd(ora, pkg, leaky):
    L = ora.lib()
    H, P, B, T = 17, 9, 3, 12
    g = pkg.synth.make_den_graph(H, P, mean_out_degree=3.0, seed=5)
    rng = np.random.default_rng(1)
    y = rng.standard_normal((T * B, P)).astype(F)
    yt = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    lp = _den_logprob_torch(g, yt, B, leaky)
    lp.sum().backward()
    tot = C.c_double()
    deriv = np.zeros_like(y)
    gs = ora.den_graph_struct(g)
    ok = L.oracle_chain_denominator(C.byref(gs), ora.omat(y), B, leaky, -1.0, C.byref(tot), ora.omat(deriv))
    assert ok == 1
    assert abs(tot.value - float(lp.sum().detach())) < 1e-5 * abs(float(lp.sum().detach()))
    np.testing.assert_allclose(deriv, -yt.grad.numpy(), rtol=2e-4, atol=2e-6)
    # occupation probabilities sum to one per (t, sequence)  (SURVEY 8(c) KAT 3)
    np.testing.assert_allclose(-deriv.sum(1), 1.0, rtol=1e-4)


def test_numerator_vs_brute_force_and_autograd(ora, pkg):
    L = ora.lib()
    B, T, P = 2, 4, 6
    sup = pkg.synth.make_supervision(B, T, P, max_alt=2, seed=3, weight=1.0)
    rng = np.random.default_rng(2)
    y = rng.standard_normal((T * B, P)).astype(F)
    post = np.zeros_like(y)
    ss = ora.supervision_struct(sup)
    tot = L.oracle_chain_numerator(C.byref(ss), ora.omat(y), ora.omat(post))
    # brute force: enumerate arc paths per sequence
    yt = torch.tensor(y, dtype=torch.float64, requires_grad=True)
    total = 0
    for s in range(B):
        a0, a1 = sup["seq_arc_begin"][s], sup["seq_arc_begin"][s + 1]
        by_time = [[a for a in range(a0, a1) if sup["state_time"][sup["arc_src"][a]] == t] for t in range(T)]
        terms = []
        for path in itertools.product(*by_time):
            okp = sup["arc_src"][path[0]] == sup["seq_state_begin"][s]
            for u, v in zip(path[:-1], path[1:]):
                okp = okp and sup["arc_dst"][u] == sup["arc_src"][v]
            if okp:
                terms.append(sum(float(sup["arc_logprob"][a]) + yt[sup["state_time"][sup["arc_src"][a]] * B + s,
                                                                  int(sup["arc_pdf"][a])] for a in path))
        total = total + torch.logsumexp(torch.stack(terms), 0)
    assert abs(tot - float(total.detach())) < 1e-5 * abs(float(total.detach()))
    total.backward()
    np.testing.assert_allclose(post, yt.grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(post.sum(1), 1.0, rtol=1e-4)


def test_chain_objf_and_deriv_composition(ora, pkg):
    L = ora.lib()
    H, P, B, T = 13, 7, 3, 6
    g = pkg.synth.make_den_graph(H, P, mean_out_degree=3.0, seed=9)
    sup = pkg.synth.make_supervision(B, T, P, seed=4, weight=1.0)
    rng = np.random.default_rng(3)
    y = rng.standard_normal((T * B, P)).astype(F)
    gs, ss = ora.den_graph_struct(g), ora.supervision_struct(sup)
    objf, l2, w = C.c_double(), C.c_double(), C.c_double()
    d, xd = np.zeros_like(y), np.zeros_like(y)
    ok = L.oracle_chain_objf_and_deriv(C.byref(gs), C.byref(ss), ora.omat(y), 0.1, 5e-5, 0.1, C.byref(objf),
                                       C.byref(l2), C.byref(w), ora.omat(d), ora.omat(xd))
    assert ok == 1 and w.value == B * T
    den = C.c_double()
    L.oracle_chain_denominator(C.byref(gs), ora.omat(y), B, 0.1, 0.0, C.byref(den), None)
    num = L.oracle_chain_numerator(C.byref(ss), ora.omat(y), None)
    assert abs(objf.value - (num - den.value)) < 1e-9
    assert abs(l2.value + 0.5 * 5e-5 * float((y.astype(np.float64) ** 2).sum())) < 1e-9
    # finite-difference check of d(objf + l2)/dy on a few entries
    for (r, c) in [(0, 0), (5, 3), (17, 6)]:
        eps = 1e-2
        vals = []
        for sgn in (+1, -1):
            y2 = y.copy()
            y2[r, c] += sgn * eps
            o2, l22, w2 = C.c_double(), C.c_double(), C.c_double()
            L.oracle_chain_objf_and_deriv(C.byref(gs), C.byref(ss), ora.omat(y2), 0.1, 5e-5, 0.1, C.byref(o2),
                                          C.byref(l22), C.byref(w2), None, None)
            vals.append(o2.value + l22.value)
        fd = (vals[0] - vals[1]) / (2 * eps)
        assert abs(fd - d[r, c]) < 2e-3, (fd, d[r, c])
    assert np.isfinite(d).all() and (xd >= 0).all()
    # NaN input -> failure path: objf = -10*weight, zero derivs
    y[0, 0] = np.nan
    ok = L.oracle_chain_objf_and_deriv(C.byref(gs), C.byref(ss), ora.omat(y), 0.1, 0.0, 0.1, C.byref(objf),
                                       C.byref(l2), C.byref(w), ora.omat(d), ora.omat(xd))
    assert ok == 0 and objf.value == -10.0 * B * T and not d.any() and not xd.any()


def test_den_initial_probs_matches_synth(ora, pkg):
    L = ora.lib()
    g = pkg.synth.make_den_graph(40, 10, seed=2)
    init = np.zeros(40, F)
    L.oracle_den_initial_probs(40, len(g["src"]), ora.iptr(g["src"]), ora.iptr(g["dst"]), ora.fptr(g["prob"]),
                               0, 100, ora.fptr(init))
    np.testing.assert_allclose(init, g["init"], rtol=1e-5, atol=1e-8)
    assert abs(init.sum() - 1.0) < 1e-5


def _orth_err(M):
    P = M.astype(np.float64) @ M.T
    s2 = np.trace(P @ P) / np.trace(P)
    return np.linalg.norm(P - s2 * np.eye(len(P)))


def test_constrain_orthonormal(ora):
    """Monotone decrease of ||MM^T - s^2 I|| and fixed point at a semi-orthogonal M
    (SURVEY 8(c) KAT 4; nnet-utils.cc:914-1032)."""
    L = ora.lib()
    rng = np.random.default_rng(0)
    M = (rng.standard_normal((16, 48)) / np.sqrt(48)).astype(F)
    errs = [_orth_err(M)]
    for _ in range(12):
        L.oracle_constrain_orthonormal(-1.0, ora.fptr(M), 16, 48, 48)
        errs.append(_orth_err(M))
    assert all(b < a for a, b in zip(errs, errs[1:])) and errs[-1] < 1e-3 * errs[0]
    Q = np.ascontiguousarray(np.linalg.qr(rng.standard_normal((48, 16)))[0].T.astype(F) * 1.7)
    Q0 = Q.copy()
    L.oracle_constrain_orthonormal(-1.0, ora.fptr(Q), 16, 48, 48)
    np.testing.assert_allclose(Q, Q0, atol=1e-5)
    # fixed scale: converges to scale^2 I
    for _ in range(30):
        L.oracle_constrain_orthonormal(1.0, ora.fptr(M), 16, 48, 48)
    np.testing.assert_allclose(M.astype(np.float64) @ M.T, np.eye(16), atol=1e-3)


def test_max_change_and_l2(ora):
    L = ora.lib()
    dots = np.asarray([4.0, 0.01, 9.0])
    mc = np.asarray([0.75, 0.75, 0.0], F)
    sf = np.zeros(3, F)
    ok = C.c_int()
    L.oracle_max_change_scales(ora.dptr(dots), ora.fptr(mc), 3, 2.0, 1.0, 1.0, ora.fptr(sf), C.byref(ok))
    # comp0: norm 2 > .75 -> .375 ; comp1 untouched ; comp2 max-change 0 = unlimited
    per = np.asarray([0.375, 1.0, 1.0])
    tot = np.sqrt((per ** 2 * dots).sum())
    glob = 2.0 / tot if tot > 2.0 else 1.0
    assert ok.value == 1
    np.testing.assert_allclose(sf, per * glob, rtol=1e-6)
    dots[0] = np.inf
    L.oracle_max_change_scales(ora.dptr(dots), ora.fptr(mc * 0), 3, 2.0, 1.0, 1.0, ora.fptr(sf), C.byref(ok))
    assert ok.value == 0
    p, d = np.ones(5, F), np.zeros(5, F)
    L.oracle_apply_l2(ora.fptr(p), ora.fptr(d), 5, -2.0 * 128 * 0.001 * 0.01)
    np.testing.assert_allclose(d, -2.0 * 128 * 0.001 * 0.01, rtol=1e-6)


def test_natural_gradient_identities(ora):
    """SURVEY 8(c) KAT 5: X_hat = X - (X W^T) W with the state BEFORE the update,
    scale^2 = tr(XX^T)/tr(X_hat X_hat^T); preconditioning shrinks the dominant
    directions; D == 1 is a no-op."""
    L = ora.lib()
    rng = np.random.default_rng(0)
    N, D, R = 64, 24, 6
    basis = rng.standard_normal((3, D))
    ng = L.oracle_ng_create(R, 4, 2000.0, 4.0)
    W, d = np.zeros((R, D), F), np.zeros(R, F)
    rho, t = C.c_float(), C.c_int()
    for it in range(14):
        X = (rng.standard_normal((N, 3)) @ basis * 3 + rng.standard_normal((N, D)) * 0.3).astype(F)
        X0 = X.copy()
        have = L.oracle_ng_state(ng, ora.fptr(W), ora.fptr(d), C.byref(rho), C.byref(t))
        scale = C.c_float()
        L.oracle_ng_precondition(ng, ora.omat(X), C.byref(scale))
        if have:
            ref = X0.astype(np.float64) - (X0.astype(np.float64) @ W.T.astype(np.float64)) @ W
            np.testing.assert_allclose(X, ref, rtol=1e-3, atol=1e-4)
            assert t.value == it
        s2 = (X0.astype(np.float64) ** 2).sum() / (X.astype(np.float64) ** 2).sum()
        assert abs(scale.value - np.sqrt(s2)) < 1e-3 * np.sqrt(s2)
        assert np.isfinite(X).all()
    # after training, the dominant 3-dim subspace is damped relative to the noise floor
    proj = np.linalg.qr(basis.T)[0]
    X = (rng.standard_normal((N, 3)) @ basis * 3 + rng.standard_normal((N, D)) * 0.3).astype(F)
    e_in = ((X @ proj) ** 2).sum() / (X ** 2).sum()
    sc = C.c_float()
    L.oracle_ng_precondition(ng, ora.omat(X), C.byref(sc))
    e_out = ((X @ proj) ** 2).sum() / (X ** 2).sum()
    # alpha=4 smoothing: F^-1 ~ (F + alpha/D tr(F) I)^-1 damps each dominant direction's energy by
    # ((tr/3 + tr/6)/(tr/6))^2 = 9 relative to the noise floor
    odds = (e_out / (1 - e_out)) / (e_in / (1 - e_in))
    assert 0.05 < odds < 0.2, odds
    L.oracle_ng_destroy(ng)
    ng = L.oracle_ng_create(R, 4, 2000.0, 4.0)
    col = rng.standard_normal((N, 1)).astype(F)
    c0 = col.copy()
    L.oracle_ng_precondition(ng, ora.omat(col), C.byref(sc))
    assert sc.value == 1.0 and (col == c0).all()
    L.oracle_ng_destroy(ng)
