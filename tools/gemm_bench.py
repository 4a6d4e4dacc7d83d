#!/usr/bin/env python3
"""Micro-benchmark of the f32 MFMA GEMM kernels on the shapes of the 7q step (chunk 1500, 128 sequences).
Usage (GPU box): python tools/gemm_bench.py [reps [shape-filter [OPTION=VALUE ...]]]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
abi = pkg.hipabi
lib = abi.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = sys.argv[2] if len(sys.argv) > 2 else ""
for spec in sys.argv[3:]:  # library options, NAME=VALUE (e.g. gemm_ring=0)
    abi.check(lib.tdnnf_set_option(spec.split("=")[0].encode(), int(spec.split("=")[1])))


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def run(name, offs, nt, B, Di, Do, step=1):
    if only and only not in name:
        return
    rho, ro, rows_in, N = pkg.synth.tdnn_indexes(offs, nt, B, t_step_out=step)
    K = len(offs)
    x = torch.randn(rows_in, Di, device="cuda")
    W = torch.randn(Do, K * Di, device="cuda") / (K * Di) ** 0.5
    b = torch.randn(Do, device="cuda")
    y = torch.zeros(N, Do, device="cuda")
    dy = torch.randn(N, Do, device="cuda")
    dx = torch.zeros(rows_in, Di, device="cuda")
    G = torch.zeros(Do, K * Di, device="cuda")
    gb = torch.zeros(Do, device="cuda")
    ix = abi.indexes(rho, ro)
    nb = lib.tdnnf_tdnn_update_workspace_bytes(Do, Di, K, N)
    ws = abi.workspace(nb)
    s = abi.stream()
    flops = 2.0 * N * K * Di * Do
    t = timed(lambda: abi.check(lib.tdnnf_tdnn_propagate(C.byref(ix), abi.pmat(x), abi.ptr(W), K * Di, Do, Di, abi.ptr(b), None, 1, abi.pmat(y), s)))
    print(f"{name:28s} fwd   M={N:7d} N={Do:5d} K={K}x{Di:5d}  {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TF")
    t = timed(lambda: abi.check(lib.tdnnf_tdnn_backprop_data(C.byref(ix), abi.pmat(dy), abi.ptr(W), K * Di, Do, Di, None, abi.pmat(dx), s)))
    print(f"{name:28s} bwd   M={rows_in:7d} N={Di:5d} K={K}x{Do:5d}  {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TF")
    t = timed(lambda: abi.check(lib.tdnnf_tdnn_update_simple(C.byref(ix), abi.pmat(x), abi.pmat(dy), Do, Di, None, 1.0, abi.ptr(G), K * Di,
                                                             abi.ptr(gb), abi.ptr(ws), nb, s)))
    print(f"{name:28s} wgrad M={Do:7d} N={K * Di:5d} K={N:7d}  {t * 1e3:8.1f} us  {flops / t / 1e9:7.1f} TF (incl. slab reduce + bias colsum)")


B = 128
run("tdnnf.linear full-rate", [-1, 0], 1564, B, 1536, 160)
run("tdnnf.affine full-rate", [0, 1], 1563, B, 160, 1536)
run("tdnnf.linear 1/3-rate", [-3, 0], 520, B, 1536, 160, step=3)
run("tdnnf.affine 1/3-rate", [0, 3], 519, B, 160, 1536, step=3)
run("tdnnf.linear 1/3 511 tiles", [-3, 0], 511, B, 1536, 160, step=3)
run("tdnn1.affine", [0], 1567, B, 220, 1536)
run("prefinal.affine", [0], 500, B, 256, 1536)
run("prefinal.linear", [0], 500, B, 1536, 256)
run("output.affine", [0], 500, B, 256, 6034)
run("config1.linear", [-1, 0, 1], 150, B, 40, 160)
run("outprobe K6016", [0], 500, B, 256, 6016)
run("outprobe K6144", [0], 500, B, 256, 6144)
run("outprobe K6034 N4096->", [0], 500, B, 4096, 6034)
run("probe160 longK 1round", [-1, 0], 512, B, 1536, 160)
run("probe128 longK", [-1, 0], 512, B, 1536, 1536)
# shapes of the natural-gradient statistics H = X W^T (rank 80 on the output side, 20 on the input side); "fwd" lines only
run("ngshape out affine R80", [0], 1563, B, 1536, 80)
run("ngshape in linear R20", [-1, 0], 1564, B, 1536, 20)
run("ngshape in affine R20", [0, 1], 1563, B, 160, 20)
run("ngshape out linear R80", [0], 1564, B, 160, 80)
