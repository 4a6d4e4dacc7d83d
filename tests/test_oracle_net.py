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

    def add(name, rows, cols, hb, lrf=1.0, l2=0.01, mc=0.75, orth=0.0, na=0):
        nonlocal begin
        comps.append(dict(name=name, begin=begin, rows=rows, cols=cols, has_bias=hb, lr_factor=lrf, l2=l2, max_change=mc,
                          orthonormal=orth, num_alpha=na))
        begin = (begin + rows * cols + na + (rows if hb else 0) + 3) // 4 * 4

    add("lda", lda_dim, lda_dim, 1, lrf=0.0, l2=0.0, mc=0.0)
    add("tdnn1.affine", 32, lda_dim, 1)
    for i, s in enumerate(strides):
        K = Kd if Kd else (2 if s > 0 else 1)
        add(f"tdnnf{i + 2}.linear", 8, K * 32, 1 if Kd else 0, orth=0.0 if Kd else -1.0, na=Kd)
        add(f"tdnnf{i + 2}.affine", 32, K * 8, 1, na=Kd)
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
    net = OracleNet(pkg, cfg, comps)
    feats = rng.standard_normal((net.num_t_in * B, 8)).astype(np.float32)
    iv = rng.standard_normal((B, 4)).astype(np.float32)
    den = pkg.synth.make_den_graph(12, 24, mean_out_degree=3.0, seed=seed + 1)
    sup = pkg.synth.make_supervision(B, T // 3, 24, seed=seed + 2)
    return cfg, comps, params, net, feats, iv, den, sup


def test_decision_hash_is_stable():
    assert [decision(0, k) % 4 for k in range(6)] == [decision(0, k) % 4 for k in range(6)]
    assert len({decision(s, 1) for s in range(50)}) == 50


@pytest.mark.parametrize("strides", [(1, 1, 0, 3, 3), (1, 0, 3), (1, 1, 1, 0, 6)])
def test_oracle_net_gradients_by_finite_differences(pkg, strides):
    T = 12 if max(strides) == 3 else 18
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=strides, T=T, relu_self_repair_scale=0.0, chain_l2=1e-3)
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
    assert bad <= max(1, total_checked // 10), (bad, total_checked)
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
