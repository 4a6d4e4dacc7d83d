#!/usr/bin/env python3
"""Is the in-step wgrad slower than the micro-benchmark because its operands are cold?  Runs the tdnnf.linear 1/3-rate
weight gradient (Do 160, 2 x 1536, 64000 rows) over `nbuf` different operand sets in rotation.
usage (GPU box): python tools/wgrad_cold.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg.hipabi
lib = abi.load()
offs, nt, B, Di, Do = [-3, 0], 500, 128, 1536, 160
rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B, t_step_out=3)
K = len(offs)
ix = abi.indexes(rho, ro)
nb = lib.tdnnf_tdnn_update_workspace_bytes(Do, Di, K, N)
ws = abi.workspace(nb)
s = abi.stream()
for nbuf in (1, 4, 16):
    xs = [torch.randn(rows_in, Di, device="cuda") for _ in range(nbuf)]
    dys = [torch.randn(N, Do, device="cuda") for _ in range(nbuf)]
    G = torch.zeros(Do, K * Di, device="cuda")
    gb = torch.zeros(Do, device="cuda")

    def call(i):
        abi.check(lib.tdnnf_tdnn_update_simple(C.byref(ix), abi.pmat(xs[i % nbuf]), abi.pmat(dys[i % nbuf]), Do, Di, None, 1.0, abi.ptr(G), K * Di,
                                               abi.ptr(gb), abi.ptr(ws), nb, s))
    for i in range(nbuf):
        call(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 32
    e0.record()
    for i in range(reps):
        call(i)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print(f"nbuf {nbuf:2d}: {t * 1e3:7.1f} us  {2.0 * N * K * Di * Do / t / 1e9:6.1f} TF")
    del xs, dys

# sustained: does the rate drop when the same kernel runs back to back for seconds (clocks under power management)?
xs = [torch.randn(rows_in, Di, device="cuda") for _ in range(4)]
dys = [torch.randn(N, Do, device="cuda") for _ in range(4)]
G = torch.zeros(Do, K * Di, device="cuda")
gb = torch.zeros(Do, device="cuda")
for chunk in range(8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(500):
        abi.check(lib.tdnnf_tdnn_update_simple(C.byref(ix), abi.pmat(xs[i % 4]), abi.pmat(dys[i % 4]), Do, Di, None, 1.0, abi.ptr(G), K * Di,
                                               abi.ptr(gb), abi.ptr(ws), nb, s))
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 500
    print(f"sustained chunk {chunk}: {t * 1e3:7.1f} us  {2.0 * N * K * Di * Do / t / 1e9:6.1f} TF")
