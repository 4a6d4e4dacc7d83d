"""Stand-alone timing of the natural-gradient statistics pass H = X~ W^T (tdnnf_ng_stats_pass): the vector-ALU kernel (ng_valu.hip) against the
MFMA rows GEMM, at the shapes of the 7q step.  usage: python3 tools/ng_pass_bench.py [reps]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
abi = importlib.import_module("tdnn-f_nas_amd.hipabi")
lib = abi.load()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for kv in sys.argv[2:]:  # further arguments: library options NAME=VALUE
    k, v = kv.split("=")
    abi.check(lib.tdnnf_set_option(k.encode(), int(v)))
# (name, rank, rows N, row offsets, Di)
SHAPES = [
    ("linear in, full rate 1500x128", 20, 200192, (0, 128), 1536),
    ("linear in, 1/3 rate 1500x128", 20, 66688, (0, 128), 1536),
    ("prefinal in 1500x128", 20, 64000, (0,), 1536),
    ("affine in, full rate", 20, 200192, (0, 128), 160),
    ("linear out, full rate", 80, 200192, (0,), 160),
    ("output out 1500x128", 80, 64000, (0,), 6034),
    ("linear in, full rate 1500x16", 20, 24096, (0, 32), 1536),
    ("linear in, 1/3 rate 1500x16", 20, 8032, (0, 32), 1536),
    ("linear in, 1/3 rate 150x64", 20, 3328, (0, 128), 1536),
]
for name, R, N, offs, Di in SHAPES:
    K = len(offs)
    ld = (Di + 31) // 32 * 32
    X = torch.randn((N + max(offs), ld), device="cuda")
    D = K * Di
    W = torch.randn((R, D), device="cuda") / D ** 0.5
    wt = torch.zeros((D + 64, R), device="cuda")
    wt[:D] = W.t()
    ldw = D
    H = torch.zeros((N, R), device="cuda")
    cap = max(1024, (N + 127) // 128)
    part = torch.zeros((cap,), dtype=torch.float64, device="cuda")
    ix = abi.indexes(1, offs)
    ref = None
    line = "%-34s N %6d D %5d R %2d :" % (name, N, D, R)
    nb = lib.tdnnf_ng_stats_pass_workspace_bytes(R, Di, K, N)
    ws = torch.zeros((nb // 4 + 16,), device="cuda")
    forms = (0, 1, 2) if (K > 1 and Di >= 1024 and N % 128 == 0 and N >= 32768 and all(o % 128 == 0 for o in offs)) else (0, 1)
    for valu in forms:
        def run():
            abi.check(lib.tdnnf_ng_stats_pass(C.byref(ix), abi.pmat(X[:, :Di]), Di, None, abi.ptr(wt), abi.ptr(W), ldw, None, abi.pmat(H), abi.ptr(part), cap,
                                              valu, abi.ptr(ws), nb, abi.stream()))
        run()
        torch.cuda.synchronize()
        if ref is None:
            ref = H.clone()
        else:
            err = ((H - ref).norm() / ref.norm()).item()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        gb = 4.0 * ((N + max(offs)) * Di + N * R) / 1e9
        line += "  %s %7.1f us %5.1f TFLOP/s %5.2f TB/s" % (("mfma", "valu", "1pass")[valu], us, 2.0 * N * D * R / us / 1e6, gb / us * 1e6 / 1e3)
    print(line + "  rel diff %.1e" % err)
