"""Denominator / numerator derivative error against the oracle as a function of the number of frames (diagnostic)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as ge
pkg = ge.load_package()
from oracle import pyoracle as ora
from tests.gpu_util import Hip, dev, host, rel_l2
hip = Hip(pkg); L = ora.lib(); F = np.float32
H, P, B = 4000, 6034, 8
g = pkg.synth.make_den_graph(H, P, mean_out_degree=12.0, seed=1)
for T in (50, 200, 500):
  for scale in (0.1, 1.0):
    sup = pkg.synth.make_supervision_from_den(g, B, T, num_paths=2, seed=T)
    rng = np.random.default_rng(T)
    y = (rng.standard_normal((T * B, P)) * scale).astype(F)
    gs, ss = ora.den_graph_struct(g), ora.supervision_struct(sup)
    objf, l2t, w = C.c_double(), C.c_double(), C.c_double()
    d_ref, xd_ref = np.zeros_like(y), np.zeros_like(y)
    L.oracle_chain_objf_and_deriv(C.byref(gs), C.byref(ss), ora.omat(y), 0.1, 0.0, 0.1, C.byref(objf), C.byref(l2t), C.byref(w), ora.omat(d_ref), ora.omat(xd_ref))
    den_ref = xd_ref.astype(np.float64) - d_ref  # gamma_den
    for mode in (1, 2):
        pkg.hipabi.check(pkg.hipabi.load().tdnnf_chain_set_denominator_mode(mode))
        dg, ds = pkg.hipabi.DenGraph(g), pkg.hipabi.Supervision(sup)
        nb = hip.chain_workspace_bytes(dg.h, B, T)
        ws = hip.ws(nb)
        res = torch.zeros(8, dtype=torch.float64, device="cuda")
        dd = torch.zeros(T * B, P, device="cuda"); xdd = torch.zeros(T * B, P, device="cuda")
        hip.chain_objf_and_deriv(dg.h, ds.h, dev(y), None, 0.1, 0.0, 0.1, hip.vec(res), dd, xdd, hip.vec(ws), nb, hip.stream())
        r = host(res); d = host(dd).astype(np.float64); xd = host(xdd).astype(np.float64) / 0.1
        den = xd - d
        print("T %4d scale %.1f mode %d: objf rel %.2e deriv rel %.2e  den rel %.2e  max|rowsum(den)-1| %.2e (oracle %.2e)  max|rowsum(num)-1| %.2e" % (
            T, scale, mode, abs(r[0] - objf.value) / abs(objf.value), rel_l2(d, d_ref), rel_l2(den, den_ref), np.abs(den.sum(1) - 1).max(),
            np.abs(den_ref.sum(1) - 1).max(), np.abs(xd.sum(1) - 1).max()), flush=True)
pkg.hipabi.load().tdnnf_chain_set_denominator_mode(0)
