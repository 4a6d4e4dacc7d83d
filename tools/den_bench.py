#!/usr/bin/env python3
"""Stand-alone timing of chain::ComputeChainObjfAndDeriv (denominator + numerator) on synthetic graphs.
usage (GPU box): [DEN_MODE=0|1|2] python tools/den_bench.py [states] [degree] [B] [T_out]   (1 persistent, 2 wide, 0 automatic)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg.hipabi
lib = abi.load()
abi.check(lib.tdnnf_chain_set_denominator_mode(int(os.environ.get("DEN_MODE", "0"))))
H = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
deg = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
T = int(sys.argv[4]) if len(sys.argv) > 4 else 500
P = 6034
den = pkg.synth.make_den_graph(H, P, mean_out_degree=deg, seed=1)
sup = pkg.synth.make_supervision_from_den(den, B, T, num_paths=2, seed=2)
dg, ds = abi.DenGraph(den), abi.Supervision(sup)
y = torch.randn(B * T, P, device="cuda")
d = torch.zeros_like(y)
res = torch.zeros(8, dtype=torch.float64, device="cuda")
nb = lib.tdnnf_chain_workspace_bytes(dg.h, B, T)
ws = abi.workspace(nb)
s = abi.stream()


def run():
    abi.check(lib.tdnnf_chain_objf_and_deriv(dg.h, ds.h, abi.pmat(y), None, 0.1, 0.0, 0.0, abi.ptr(res), abi.pmat(d), None, abi.ptr(ws), nb, s))


run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
r = res.cpu().numpy()
print("states %d arcs %d  B %d  T %d: %.1f ms  (%.1f us per frame and pass)  objf/frame %.4f ok %d" % (H, len(den["src"]), B, T, ms, 1e3 * ms / (2 * T), r[0] / r[2], r[5]))
