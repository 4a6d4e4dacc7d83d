"""Pins the CPU reference of the whole training step (tests/oracle_net.py): finite differences of the
total objective against the back-propagated raw gradients, on a tiny TDNN-F net that exercises every
grid case (stride 1 at full rate, the stride-0 layer, the stride-1 -> step-3 transition with the rho
row order, stride-3 layers)."""
import numpy as np
import pytest

from tests.oracle_net import OracleNet, decision


def tiny_setup(pkg, strides=(1, 1, 0, 3, 3), T=12, B=2, seed=0, **kw):
    cfg = pkg.trainer.make_config(frames_per_chunk=T, num_sequences=B, strides=list(strides), bottleneck=8, feat_dim=8,
                                  ivector_dim=4, num_pdfs=24, hidden_dim=32, small_dim=16, **kw)
    comps, begin = [], 0
    lda_dim = 3 * 8 + 4
    Kd = cfg.darts_num_offsets
    bnC, bn = cfg.bn_num_choices, cfg.bottleneck_dim[0]

    def add(name, rows, cols, hb, lrf=1.0, l2=0.01, mc=0.75, orth=0.0, na=0):
        nonlocal begin
        comps.append(dict(name=name, begin=begin, rows=rows, cols=cols, has_bias=hb, lr_factor=lrf, l2=l2, max_change=mc,
                          orthonormal=orth, num_alpha=na))
        begin = (begin + rows * cols + na + (rows if hb else 0) + 3) // 4 * 4

    add("lda", lda_dim, lda_dim, 1, lrf=0.0, l2=0.0, mc=0.0)
    add("tdnn1.affine", 32, lda_dim, 1)
    lo = kw.get("layer_offsets")
    for i, s in enumerate(strides):
        K = Kd if Kd else (2 if s > 0 else 1)
        Kl, Ka = (K, K) if lo is None else (2 if lo[i][0] > 0 else 1, 2 if lo[i][1] > 0 else 1)
        if bnC:  # X.softmax (Onehot) / X.alpha (ConstantFunction): C-vector, no l2, no max-change
            add(f"tdnnf{i + 2}" + (".softmax" if cfg.bn_mode == 0 else ".alpha"), bnC, 1, 0, l2=0.0, mc=0.0)
        add(f"tdnnf{i + 2}.linear", bn, Kl * 32, 1 if Kd else 0, orth=0.0 if Kd else -1.0, na=Kd)
        add(f"tdnnf{i + 2}.affine", 32, Ka * bn, 1, na=Kd)
    add("prefinal-l", 16, 32, 0, orth=-1.0)
    for hn in ("chain", "xent"):
        add(f"prefinal-{hn}.affine", 32, 16, 1)
        add(f"prefinal-{hn}.linear", 16, 32, 0, orth=-1.0)
        add("output.affine" if hn == "chain" else "output-xent.affine", 24, 16, 1, lrf=1.0 if hn == "chain" else 5.0, l2=0.002, mc=1.5)
    rng = np.random.default_rng(seed)
    params = np.zeros(begin, np.float32)
    for c in comps:
        n = c["rows"] * c["cols"]
        params[c["begin"]:c["begin"] + n] = (rng.standard_normal(n) / np.sqrt(c["cols"])).astype(np.float32)
        na = c["num_alpha"]
        params[c["begin"] + n:c["begin"] + n + na] = rng.standard_normal(na).astype(np.float32) * 0.5
        if c["has_bias"]:
            params[c["begin"] + n + na:c["begin"] + n + na + c["rows"]] = rng.standard_normal(c["rows"]).astype(np.float32) * 0.3
        if c["name"].endswith((".softmax", ".alpha")):
            params[c["begin"]:c["begin"] + n] = rng.standard_normal(n).astype(np.float32) * 0.5
    net = OracleNet(pkg, cfg, comps)
    feats = rng.standard_normal((net.num_t_in * B, 8)).astype(np.float32)
    iv = rng.standard_normal((B, 4)).astype(np.float32)
    den = pkg.synth.make_den_graph(12, 24, mean_out_degree=3.0, seed=seed + 1)
    sup = pkg.synth.make_supervision(B, T // 3, 24, seed=seed + 2)
    return cfg, comps, params, net, feats, iv, den, sup


def test_decision_hash_is_stable():
    assert [decision(0, k) % 4 for k in range(6)] == [decision(0, k) % 4 for k in range(6)]
    assert len({decision(s, 1) for s in range(50)}) == 50


# the last three: derived children (derive.child_config_kwargs), X.linear {-a, 0} / X.affine {0, b} per layer
@pytest.mark.parametrize("strides", [(1, 1, 0, 3, 3), (1, 0, 3), (1, 1, 1, 0, 6), ((1, 2), (0, 1), (2, 0), (3, 0), (2, 3)),
                                     ((2, 1), (0, 0), (1, 3), (2, 1)), ((1, 1), (3, 6), (6, 3))])
def test_oracle_net_gradients_by_finite_differences(pkg, strides):
    kw = {}
    if isinstance(strides[0], tuple):
        kw["layer_offsets"] = list(strides)
        strides = tuple(max(a, b) for a, b in strides)
    T = 12 if max(strides) <= 3 and not kw else 18
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=strides, T=T, relu_self_repair_scale=0.0, chain_l2=1e-3, **kw)
    res, grads, acts = net.forward_backward(params, feats, iv, den, sup)
    assert res["ok"] == 1
    post = acts["xent.post"]
    rng = np.random.default_rng(5)

    def total(p):
        r, _, _ = net.forward_backward(p, feats, iv, den, sup, fixed_xent_post=post, forward_only=True)
        return r["objf"] + r["l2_term"] + cfg.xent_regularize * r["xent_objf"]

    bad, total_checked = 0, 0
    for c in comps[1:]:
        n = c["rows"] * c["cols"] + (c["rows"] if c["has_bias"] else 0)
        for idx in rng.choice(n, size=3, replace=False):
            i = c["begin"] + int(idx)
            ok_any = False
            for eps in (4e-3, 1e-3):  # ReLU kinks make single-eps differences noisy on a net this small
                pp, pm = params.copy(), params.copy()
                pp[i] += eps
                pm[i] -= eps
                fd = (total(pp) - total(pm)) / (2 * eps)
                if abs(fd - grads[i]) <= 2e-2 * max(abs(fd), abs(grads[i])) + 3e-3:
                    ok_any = True
            total_checked += 1
            if not ok_any:
                bad += 1
                print(c["name"], idx, fd, grads[i])
    # (children run every layer at the input frame rate: several times the ReLU units, so more kinks inside the stencils;
    #  their bookkeeping is pinned exactly by test_child_forward_by_time_lookup)
    assert bad <= max(1, total_checked // (4 if kw else 10)), (bad, total_checked)
    assert not grads[comps[0]["begin"]:comps[1]["begin"]].any()  # the fixed lda layer gets no gradient


def test_oracle_net_update_moves_params_and_keeps_lda(pkg):
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg)
    res, grads, _ = net.forward_backward(params, feats, iv, den, sup)
    p2 = net.update(params, grads, 1e-3, float(cfg.num_sequences), step=3)
    assert np.isfinite(p2).all() and (p2 != params).any()
    lda = slice(comps[0]["begin"], comps[1]["begin"])
    assert (p2[lda] == params[lda]).all()
    # the global max-change bounds the step
    assert np.linalg.norm(p2 - params) < 2.0 + 1.0  # + slack for the orthonormal steps


@pytest.mark.parametrize("flags", [0, 2])
def test_oracle_darts_net_alpha_gradient_by_finite_differences(pkg, flags):
    """Offset supernet (TdnnDARTSV3 in every tdnnf layer): the accumulated architecture-logit update equals
    5 x d(objective)/d(log-alpha) in the softmax and free-select modes (nnet-tdnn-component.cc:574-586 with lr = 1)."""
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=(1, 1, 1), T=12, relu_self_repair_scale=0.0,
                                                              darts_num_offsets=3, darts_flags=flags)
    draws = np.random.default_rng(9).random(3 * 2 * 4).astype(np.float32)
    res, grads, acts = net.forward_backward(params, feats, iv, den, sup, draws=draws)
    post = acts["xent.post"]

    def total(p):
        r, _, _ = net.forward_backward(p, feats, iv, den, sup, fixed_xent_post=post, forward_only=True, draws=draws)
        return r["objf"] + r["l2_term"] + cfg.xent_regularize * r["xent_objf"]

    bad = checked = 0
    for c in comps:
        for k in range(c["num_alpha"]):
            i = c["begin"] + c["rows"] * c["cols"] + k
            ok_any = False
            for eps in (4e-3, 1e-3):
                pp, pm = params.copy(), params.copy()
                pp[i] += eps
                pm[i] -= eps
                fd = 5.0 * (total(pp) - total(pm)) / (2 * eps)
                if abs(fd - grads[i]) <= 3e-2 * max(abs(fd), abs(grads[i])) + 2e-2:
                    ok_any = True
            checked += 1
            bad += 0 if ok_any else 1
            if not ok_any:
                print(c["name"], k, fd, grads[i])
    assert checked == 18 and bad <= 2, (bad, checked)


@pytest.mark.parametrize("mode", [1, 2])
def test_oracle_bottleneck_supernet_alpha_gradient_by_finite_differences(pkg, mode):
    """Bottleneck-dimension supernet, cv-update wiring (ConstantFunction -> (Gumbel)SoftmaxFlops -> Sum -> CopyN ->
    ElementwiseProduct with the linear blocks): the accumulated update of X.alpha equals
    5 x d(objective + flops penalty)/d(alpha), the penalty being scale / C * <p, flops> summed over rows / rows."""
    dims = [2, 1, 3, 2]
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=(1, 0, 3), T=12, relu_self_repair_scale=0.0,
                                                              bn_choice_dims=dims, bn_mode=mode, bn_flops_scale=0.3, bn_temp_proportion=0.7)
    draws = np.random.default_rng(4).uniform(0.05, 0.95, 3 * 4).astype(np.float32)
    res, grads, acts = net.forward_backward(params, feats, iv, den, sup, draws=draws)
    post = acts["xent.post"]
    flops = -np.cumsum(dims).astype(np.float64)

    def penalty(p):
        tot = 0.0
        for i in range(3):
            c = net.comp[f"tdnnf{i + 2}.alpha"]
            a = p[c["begin"]:c["begin"] + 4].astype(np.float64)
            if mode == 2:
                a = (a - np.log(-np.log(draws[4 * i:4 * i + 4].astype(np.float64)))) / 0.7
            q = np.exp(a - a.max())
            q /= q.sum()
            tot += 0.3 / 4 * float(q @ flops)
        return tot

    def total(p):
        r, _, _ = net.forward_backward(p, feats, iv, den, sup, fixed_xent_post=post, forward_only=True, draws=draws)
        return r["objf"] + r["l2_term"] + cfg.xent_regularize * r["xent_objf"] + penalty(p)

    bad = checked = 0
    for i in range(3):
        c = net.comp[f"tdnnf{i + 2}.alpha"]
        for k in range(4):
            j = c["begin"] + k
            ok_any = False
            for eps in (4e-3, 1e-3):
                pp, pm = params.copy(), params.copy()
                pp[j] += eps
                pm[j] -= eps
                fd = 5.0 * (total(pp) - total(pm)) / (2 * eps)
                if abs(fd - grads[j]) <= 3e-2 * max(abs(fd), abs(grads[j])) + 2e-2:
                    ok_any = True
            checked += 1
            bad += 0 if ok_any else 1
            if not ok_any:
                print(c["name"], k, fd, grads[j])
    assert checked == 12 and bad <= 1, (bad, checked)


def test_oracle_bottleneck_supernet_onehot_is_a_truncated_bottleneck(pkg):
    """Pretrain wiring (OnehotFunction): with choice j sampled the layer equals a TDNN-F layer whose bottleneck is the
    first cum[j] columns; the Onehot component's own vector still collects colsum of its output derivative."""
    dims = [2, 1, 3, 2]
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=(1, 0, 3), T=12, relu_self_repair_scale=0.0,
                                                              bn_choice_dims=dims, bn_mode=0)
    draws = np.asarray([0.30, 0.80, 0.55], np.float32)  # choices 1, 3, 2 -> bottlenecks 3, 8, 6
    res, grads, acts = net.forward_backward(params, feats, iv, den, sup, draws=draws)
    assert res["ok"] == 1
    cum = np.cumsum(dims)
    for i, j in enumerate([1, 3, 2]):
        cl, ca = net.comp[f"tdnnf{i + 2}.linear"], net.comp[f"tdnnf{i + 2}.affine"]
        gl = grads[cl["begin"]:cl["begin"] + cl["rows"] * cl["cols"]].reshape(cl["rows"], cl["cols"])
        assert not gl[cum[j]:].any() and gl[:cum[j]].any()       # masked-out rows of the linear weights get no gradient
        ga = grads[ca["begin"]:ca["begin"] + ca["rows"] * ca["cols"]].reshape(ca["rows"], -1, cl["rows"])
        assert not ga[:, :, cum[j]:].any() and ga[:, :, :cum[j]].any()
        cs = net.comp[f"tdnnf{i + 2}.softmax"]
        g = grads[cs["begin"]:cs["begin"] + 4]
        assert np.all(np.isfinite(g)) and g.any()


CHILDREN = [((1, 2), (0, 1), (2, 0), (3, 0), (2, 3)), ((2, 1), (0, 0), (1, 3), (2, 1)), ((1, 1), (3, 6), (6, 3)), ((0, 2), (5, 0), (4, 6))]


@pytest.mark.parametrize("lo", CHILDREN + [((1, 1), (1, 1), (0, 0), (3, 3), (3, 3)), ((1, 1), (0, 0), (6, 6))], ids=lambda v: "-".join("%d.%d" % ab for ab in v))
def test_child_forward_by_time_lookup(pkg, lo):
    """Derived children (X.linear {-a, 0}, X.affine {0, b}, any a, b): every layer recomputed frame by frame from
    dictionaries keyed by TIME -- no grids arithmetic, no row_offsets / row_stride, no rho row order -- must reproduce
    the oracle's activations on the oracle's grids; a frame the grids do not provide is a KeyError here."""
    strides = tuple(max(a, b) for a, b in lo)
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=strides, T=18, relu_self_repair_scale=0.0, layer_offsets=list(lo))
    _, _, acts = net.forward_backward(params, feats, iv, den, sup, forward_only=True)
    B, Hd, bn = cfg.num_sequences, cfg.hidden_dim, cfg.bottleneck_dim[0]

    def frames(x, g):  # matrix on grid g (t0, step, n), rows t-major -> {t: (B x D)}
        assert x.shape[0] == g[2] * B
        return {g[0] + i * g[1]: x[i * B:(i + 1) * B].astype(np.float64) for i in range(g[2])}

    def matrix(fr, g):
        return np.concatenate([fr[g[0] + i * g[1]] for i in range(g[2])], axis=0)

    prev = frames(acts["tdnn1.batchnorm"], net.g_lda)
    for i, ((a, b), Ly) in enumerate(zip(lo, net.layers)):
        nm = f"tdnnf{i + 2}"
        Wl, Wa, ba = net.W(params, nm + ".linear").astype(np.float64), net.W(params, nm + ".affine").astype(np.float64), net.b(params, nm + ".affine")
        lin_taps = [-a, 0] if a > 0 else [0]
        aff_taps = [0, b] if b > 0 else [0]
        assert Wl.shape == (bn, len(lin_taps) * Hd) and Wa.shape == (Hd, len(aff_taps) * bn)
        gl, go = Ly["lin"], Ly["out"]
        lin = {t: sum(prev[t + o] @ Wl[:, k * Hd:(k + 1) * Hd].T for k, o in enumerate(lin_taps)) for t in (gl[0] + j * gl[1] for j in range(gl[2]))}
        np.testing.assert_allclose(matrix(lin, gl), acts[nm + ".linear"], rtol=2e-4, atol=2e-5)
        aff = {t: sum(lin[t + o] @ Wa[:, k * bn:(k + 1) * bn].T for k, o in enumerate(aff_taps)) + ba for t in (go[0] + j * go[1] for j in range(go[2]))}
        relu = np.maximum(matrix(aff, go), 0)
        mean, var = relu.mean(0), relu.var(0)
        z = (relu - mean) / np.sqrt(var + 1e-3)  # BatchNorm over every row of the layer's output grid (A3)
        out = {t: cfg.bypass_scale * prev[t] + z[j * B:(j + 1) * B] for j, t in enumerate(go[0] + j * go[1] for j in range(go[2]))}
        np.testing.assert_allclose(matrix(out, go), acts[nm + ".noop"], rtol=5e-4, atol=5e-5)
        prev = out
    # the last layer's grid is the chain output grid
    assert net.layers[-1]["out"] == (0, cfg.frame_subsampling, cfg.frames_per_chunk // cfg.frame_subsampling)
    # and nothing is computed that no one reads: each grid is exactly as long as its consumers need (+ rho padding)
    for (a, b), Ly in zip(lo, net.layers):
        gl, go, gi = Ly["lin"], Ly["out"], Ly["inn"]
        last_needed = go[0] + (go[2] - 1) * go[1] + b
        rho = go[1] // gl[1]
        assert gl[0] == go[0] and 0 <= (gl[0] + (gl[2] - 1) * gl[1]) - last_needed < max(rho, 1) * gl[1]
        assert gi[0] == gl[0] - a and gi[0] + (gi[2] - 1) * gi[1] == gl[0] + (gl[2] - 1) * gl[1]


def test_oracle_dropout_gradients_by_finite_differences(pkg):
    """GeneralDropoutComponent masks (continuous, one row per sequence) between BatchNorm and the bypass sum: with the masks
    fixed the net stays differentiable; also: proportion 0 is the identity, the masks have mean ~1 and live in [1 - 2p, 1 + 2p]."""
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=(1, 0, 3), T=18, relu_self_repair_scale=0.0, use_dropout=1)
    nd = (cfg.num_layers + 1) * cfg.num_sequences * cfg.hidden_dim
    draws = np.random.default_rng(9).uniform(0, 1, nd).astype(np.float32)
    r0, g0, a0 = net.forward_backward(params, feats, iv, den, sup, draws=draws)
    net.set_dropout_proportion(0.0)
    r1, g1, _ = net.forward_backward(params, feats, iv, den, sup)
    assert r0["objf"] == r1["objf"] and np.array_equal(g0, g1)  # proportion 0: nothing changes
    net.set_dropout_proportion(0.3)
    masks = net._dropout_masks(draws)
    assert masks.shape == (cfg.num_layers + 1, cfg.num_sequences, cfg.hidden_dim) and masks.min() >= 0.4 - 1e-6 and masks.max() <= 1.6 + 1e-6
    assert abs(masks.mean() - 1.0) < 0.05
    res, grads, acts = net.forward_backward(params, feats, iv, den, sup, draws=draws)
    assert res["objf"] != r0["objf"]
    post = acts["xent.post"]

    def total(p):
        r, _, _ = net.forward_backward(p, feats, iv, den, sup, fixed_xent_post=post, forward_only=True, draws=draws)
        return r["objf"] + r["l2_term"] + cfg.xent_regularize * r["xent_objf"]

    rng = np.random.default_rng(5)
    bad = total_checked = 0
    for c in comps[1:]:
        n = c["rows"] * c["cols"] + (c["rows"] if c["has_bias"] else 0)
        for idx in rng.choice(n, size=3, replace=False):
            i = c["begin"] + int(idx)
            ok = False
            for eps in (4e-3, 1e-3):
                pp, pm = params.copy(), params.copy()
                pp[i] += eps
                pm[i] -= eps
                fd = (total(pp) - total(pm)) / (2 * eps)
                ok = ok or abs(fd - grads[i]) <= 2e-2 * max(abs(fd), abs(grads[i])) + 3e-3
            total_checked += 1
            bad += not ok
    assert bad <= max(1, total_checked // 10), (bad, total_checked)
