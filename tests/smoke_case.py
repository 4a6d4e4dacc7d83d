"""One small invocation of the hot path on cuda:0 (a TDNN-F layer forward/backward through the
C-ABI plus the chain objective), checked against the CPU oracle.  Used by __graft_entry__.smoke()."""
import ctypes as C

import numpy as np


def run(pkg):
    import torch
    from oracle import pyoracle as ora
    from tests.gpu_util import Hip, dev, host, rel_l2
    assert torch.cuda.is_available(), "smoke() needs a GPU"
    hip = Hip(pkg)
    L = ora.lib()
    rng = np.random.default_rng(0)
    F = np.float32
    offs, nt, B, Di, Do = [-1, 0], 20, 8, 1536, 160
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B)
    K = 2
    x = rng.standard_normal((rows_in, Di)).astype(F)
    W = (rng.standard_normal((Do, K * Di)) / np.sqrt(K * Di)).astype(F)
    dy = rng.standard_normal((N, Do)).astype(F)
    ix = pkg.hipabi.indexes(rho, ro)
    y_ref = np.zeros((N, Do), F)
    L.oracle_tdnn_propagate(ora.omat(x), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), None, None, 2, ora.omat(y_ref))
    xd, Wd, yd = dev(x), dev(W), torch.zeros(N, Do, device="cuda")
    hip.tdnn_propagate(C.byref(ix), xd, hip.vec(Wd), K * Di, Do, Di, None, None, 2, yd, hip.stream())
    e_fwd = rel_l2(host(yd), y_ref)
    dx_ref = np.zeros((rows_in, Di), F)
    L.oracle_tdnn_backprop_data(ora.omat(dy), ora.fptr(W), K * Di, Do, Di, K, rho, ora.iptr(ro), None, ora.omat(dx_ref))
    dxd = torch.zeros(rows_in, Di, device="cuda")
    hip.tdnn_backprop_data(C.byref(ix), dev(dy), hip.vec(Wd), K * Di, Do, Di, None, dxd, hip.stream())
    e_bwd = rel_l2(host(dxd), dx_ref)
    G_ref = np.zeros_like(W)
    L.oracle_tdnn_update_simple(ora.omat(x), ora.omat(dy), Do, Di, K, rho, ora.iptr(ro), None, 1.0, ora.fptr(G_ref), K * Di, None)
    G = torch.zeros(Do, K * Di, device="cuda")
    nb = hip.tdnn_update_workspace_bytes(Do, Di, K, N)
    ws = hip.ws(nb)
    hip.tdnn_update_simple(C.byref(ix), xd, dev(dy), Do, Di, None, 1.0, hip.vec(G), K * Di, None, hip.vec(ws), nb, hip.stream())
    e_grad = rel_l2(host(G), G_ref)
    # chain objective on a small graph
    H, P, Bc, T = 100, 80, 4, 12
    g = pkg.synth.make_den_graph(H, P, mean_out_degree=5.0, seed=1)
    sup = pkg.synth.make_supervision(Bc, T, P, seed=2)
    yy = rng.standard_normal((T * Bc, P)).astype(F)
    gs, ss = ora.den_graph_struct(g), ora.supervision_struct(sup)
    objf, l2t, w = C.c_double(), C.c_double(), C.c_double()
    d_ref = np.zeros_like(yy)
    L.oracle_chain_objf_and_deriv(C.byref(gs), C.byref(ss), ora.omat(yy), 0.1, 0.0, 0.1, C.byref(objf), C.byref(l2t),
                                  C.byref(w), ora.omat(d_ref), None)
    dg, ds = pkg.hipabi.DenGraph(g), pkg.hipabi.Supervision(sup)
    nb = hip.chain_workspace_bytes(dg.h, Bc, T)
    ws2 = hip.ws(nb)
    res = torch.zeros(8, dtype=torch.float64, device="cuda")
    dd = torch.zeros(T * Bc, P, device="cuda")
    hip.chain_objf_and_deriv(dg.h, ds.h, dev(yy), None, 0.1, 0.0, 0.1, hip.vec(res), dd, None, hip.vec(ws2), nb, hip.stream())
    e_objf = abs(host(res)[0] - objf.value) / abs(objf.value)
    e_deriv = rel_l2(host(dd), d_ref)
    print(f"smoke: tdnn fwd {e_fwd:.2e} bwd {e_bwd:.2e} grad {e_grad:.2e}; chain objf {e_objf:.2e} deriv {e_deriv:.2e}")
    assert e_fwd < 1e-4 and e_bwd < 1e-4 and e_grad < 1e-3 and e_objf < 1e-4 and e_deriv < 1e-3
