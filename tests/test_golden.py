"""Committed known-answer vectors (tests/golden/r01_golden.npz, generator: tests/golden/make_golden.py).
The reference ships no golden vectors; groups tdnn_/bn_/den_ come from an independent float64 PyTorch formulation,
net_ is an oracle-generated regression pin.  CPU tests hold the oracle to them, `-m gpu` tests hold the HIP path to them."""
import ctypes as C
import os

import numpy as np
import pytest

F = np.float32
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r01_golden.npz"))


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def den_graph():
    H, P, B, T = (int(v) for v in G["den_dims"])
    return dict(H=H, P=P, src=G["den_src"], dst=G["den_dst"], pdf=G["den_pdf"], prob=G["den_prob"], init=G["den_init"]), B, T


# ------------------------------------------------------------------------------------------------- CPU: the oracle
@pytest.mark.parametrize("tag", ["k3", "stride3"])
def test_oracle_tdnn_matches_golden(ora, tag):
    L = ora.lib()
    g = {k[len(f"tdnn_{tag}_"):]: G[k] for k in G.files if k.startswith(f"tdnn_{tag}_")}
    nt, B, Di, Do, step, rho, rows_in, N = (int(v) for v in g["dims"])
    K, ro = len(g["offsets"]), np.ascontiguousarray(g["row_offsets"])
    x, W, b, c, dy = (np.ascontiguousarray(g[k]) for k in ("x", "W", "b", "c", "dy"))
    y = np.zeros((N, Do), F)
    L.oracle_tdnn_propagate(ora.omat(x), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), ora.fptr(b), ora.fptr(c), 1, ora.omat(y))
    assert rel(y, g["y"]) < 2e-6
    dx = np.zeros_like(x)
    L.oracle_tdnn_backprop_data(ora.omat(dy), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), ora.fptr(c), ora.omat(dx))
    assert rel(dx, g["dx"]) < 2e-6
    Wg, bg = np.zeros_like(W), np.zeros_like(b)
    L.oracle_tdnn_update_simple(ora.omat(x), ora.omat(dy), Do, Di, K, rho, ora.iptr(ro), ora.fptr(c), 1.0, ora.fptr(Wg), K * Di, ora.fptr(bg))
    assert rel(Wg, g["dW"]) < 2e-6 and rel(bg, g["db"]) < 2e-6


def test_oracle_batchnorm_matches_golden(ora):
    L = ora.lib()
    x, dz = np.ascontiguousarray(G["bn_x"]), np.ascontiguousarray(G["bn_dz"])
    z, memo, dx = np.zeros_like(x), np.zeros((5, x.shape[1]), F), np.zeros_like(x)
    L.oracle_batchnorm_propagate(ora.omat(x), 1e-3, 1.0, ora.omat(z), ora.fptr(memo))
    L.oracle_batchnorm_backprop(ora.omat(z), ora.omat(dz), 1.0, ora.fptr(memo), ora.omat(dx))
    assert rel(z, G["bn_z"]) < 2e-6 and rel(dx, G["bn_dx"]) < 2e-5


def test_oracle_denominator_matches_golden(ora):
    L = ora.lib()
    g, B, T = den_graph()
    y = np.ascontiguousarray(G["den_y"])
    tot, deriv = C.c_double(), np.zeros_like(y)
    gs = ora.den_graph_struct(g)
    assert L.oracle_chain_denominator(C.byref(gs), ora.omat(y), B, float(G["den_leaky"][0]), -1.0, C.byref(tot), ora.omat(deriv)) == 1
    assert abs(tot.value - G["den_logprob"][0]) < 1e-6 * abs(G["den_logprob"][0])
    assert rel(-deriv, G["den_occupancy"]) < 1e-4


def test_oracle_net_step_matches_golden(pkg):
    from tests.test_oracle_net import tiny_setup
    cfg, comps, params, net, feats, iv, den, sup = tiny_setup(pkg, strides=(1, 0, 3), T=12, B=2, seed=3)
    res, grads, _ = net.forward_backward(params, feats, iv, den, sup, step=0)
    assert np.allclose([res["objf"], res["xent_objf"], res["weight"]], G["net_objf"], rtol=1e-6)
    norms = [np.linalg.norm(grads[c["begin"]:c["begin"] + c["rows"] * c["cols"]].astype(np.float64)) for c in comps]
    assert np.allclose(norms, G["net_grad_norms"], rtol=1e-5, atol=1e-9)
    p2 = net.update(params, grads, 1e-3, float(cfg.num_sequences), 0)
    assert np.allclose(np.linalg.norm((p2 - params).astype(np.float64)), G["net_update_norm"][0], rtol=1e-5)


# ------------------------------------------------------------------------------------------------ GPU: the HIP path
@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["k3", "stride3"])
def test_hip_tdnn_matches_golden(pkg, tag):
    from tests.gpu_util import Hip, dev, host, padded
    hip = Hip(pkg)
    g = {k[len(f"tdnn_{tag}_"):]: G[k] for k in G.files if k.startswith(f"tdnn_{tag}_")}
    nt, B, Di, Do, step, rho, rows_in, N = (int(v) for v in g["dims"])
    K = len(g["offsets"])
    ix = pkg.hipabi.indexes(rho, g["row_offsets"])
    xd, _ = padded(g["x"])
    dyd, _ = padded(g["dy"])
    yd, _ = padded(np.zeros((N, Do), F))
    dxd, _ = padded(np.zeros((rows_in, Di), F))
    Wd, bd, cd = dev(g["W"]), dev(g["b"]), dev(g["c"])
    hip.tdnn_propagate(C.byref(ix), xd, hip.vec(Wd), K * Di, Do, Di, hip.vec(bd), hip.vec(cd), 1, yd, hip.stream())
    assert rel(host(yd), g["y"]) < 2e-5
    hip.tdnn_backprop_data(C.byref(ix), dyd, hip.vec(Wd), K * Di, Do, Di, hip.vec(cd), dxd, hip.stream())
    assert rel(host(dxd), g["dx"]) < 2e-5
    Wacc, bacc = dev(np.zeros_like(g["W"])), dev(np.zeros_like(g["b"]))
    nbytes = hip.tdnn_update_workspace_bytes(Do, Di, K, N)
    ws = hip.ws(nbytes)
    hip.tdnn_update_simple(C.byref(ix), xd, dyd, Do, Di, hip.vec(cd), 1.0, hip.vec(Wacc), K * Di, hip.vec(bacc), hip.vec(ws), nbytes, hip.stream())
    assert rel(host(Wacc), g["dW"]) < 2e-5 and rel(host(bacc), g["db"]) < 2e-5


@pytest.mark.gpu
def test_hip_net_step_matches_golden(pkg):
    """The C++ trainer on the golden tiny net: objective, per-component gradient norms, update norm."""
    from tests.gpu_util import dev, host
    from tests.test_oracle_net import tiny_setup
    cfg, comps, params, ref, feats, iv, den, sup = tiny_setup(pkg, strides=(1, 0, 3), T=12, B=2, seed=3)
    net = pkg.trainer.ChainNet(cfg)
    assert [c["begin"] for c in net.components] == [c["begin"] for c in comps]
    net.set_params(params)
    dg, ds = pkg.hipabi.DenGraph(den), pkg.hipabi.Supervision(sup)
    r = host(net.forward_backward(dev(feats), dev(iv), dg, ds, step=0))
    assert abs(r[0] - G["net_objf"][0]) < 1e-4 * abs(G["net_objf"][0]) and abs(r[6] - G["net_objf"][1]) < 1e-4 * abs(G["net_objf"][1])
    g = host(net.grads)
    norms = [np.linalg.norm(g[c["begin"]:c["begin"] + c["rows"] * c["cols"]].astype(np.float64)) for c in comps]
    assert np.allclose(norms, G["net_grad_norms"], rtol=1e-3, atol=1e-7)
    net.update(1e-3, step=0)
    assert np.allclose(np.linalg.norm((host(net.params) - params).astype(np.float64)), G["net_update_norm"][0], rtol=2e-3)
    net.close()
